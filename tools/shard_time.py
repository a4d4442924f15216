"""GPU: what one rank of an N-way sharded Level 1 does (distributed.level1_slabs_sharded), timed on ONE GPU with the ranks played one
after the other -- a projection: no second GPU, the neighbour exchange is a device-to-device copy here.
usage: python tools/shard_time.py [size] [world]"""
import os
import sys
import time

import gc
import numpy as np
gc.disable()   # (a cyclic collection in the middle of a timed call would look like a 70 ms kernel)
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from contourist_amd import _ffi, distributed, synthetic   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
A = synthetic.smooth_noise_torch((n, n, n), 1235, 1400, dev)
whole = _ffi.Context(0)
whole.adopt_device_grid(A.data_ptr(), (n, n, n), keepalive=A)
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    whole.extract3d(0.0, 1)
    t1 = time.perf_counter()
    post = whole.postprocess3d(0)
    t2 = time.perf_counter()
print("undivided %d^3: extract %.2f ms, Level 1 %.2f ms, %d triangles" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, post["n_triangles"]))
del whole
ctxs = [_ffi.Context(0) for _ in range(world)]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for rnd in range(rounds):      # later rounds: buffers exist
    lists, t_local = [], []
    for r in range(world):
        lay = distributed.shard_layout(n, world, r)
        local = A[lay["e0"]:lay["e1"]]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lists.append(distributed.shard_local(ctxs[r], local, lay, 0.0, (n, n, n), torch_device=dev))
        t_local.append((time.perf_counter() - t0) * 1e3)
        if rnd > 0 and t_local[-1] > 20.0:     # the stall: when (CLOCK_MONOTONIC ns, to line up with a rocprofv3 --hip-trace)
            print("SLOW local call: round %d rank %d %.2f ms, ended at monotonic %d ns" % (rnd, r, t_local[-1], time.clock_gettime_ns(time.CLOCK_MONOTONIC)), flush=True)
    small, t_pair = [], []
    for r in range(world):
        t0 = time.perf_counter()
        if r == 0:
            pairs, unmatched = np.zeros((0, 2), dtype=np.int64), 0
        else:
            pairs, unmatched = distributed.pair_labels(lists[r]["own1"][0], lists[r]["own1"][1], lists[r - 1]["copy4"][0], lists[r - 1]["copy4"][1])
        small.append(distributed.shard_small(lists[r], pairs, unmatched))
        t_pair.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    answers, stats = distributed.merge_shard_components(small)
    t_merge = (time.perf_counter() - t0) * 1e3
    t_fin = []
    for r in range(world):
        t0 = time.perf_counter()
        out = distributed.shard_finish(ctxs[r], lists[r], answers[r], download=False)
        t_fin.append((time.perf_counter() - t0) * 1e3)
    print("round", rnd, "local ms per rank:", " ".join("%.2f" % v for v in t_local), "| pairing:", " ".join("%.2f" % v for v in t_pair),
          "| finish:", " ".join("%.2f" % v for v in t_fin))
print("per rank of %d (second round), ms: march + local post-pass %.2f..%.2f, pairing with the lower neighbour %.2f..%.2f, finish %.2f..%.2f; "
      "rank 0 merge %.2f ms %s; boundary triangles per rank up to %d" % (
          world, min(t_local), max(t_local), min(t_pair[1:] or [0]), max(t_pair[1:] or [0]), min(t_fin), max(t_fin), t_merge, stats,
          max(L["n_own_lower"] + L["n_upper_copies"] for L in lists)))
