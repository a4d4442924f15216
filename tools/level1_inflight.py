#!/usr/bin/env python3
"""Level 0 + Level 1 of the 512^3 bench field on 1..3 contexts driven by as many host threads (ctypes releases the GIL inside a call):
what a caller streaming volumes through cx_extract3d + cx_postprocess3d gets per volume when it keeps several in flight."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
A = torch.from_numpy(synthetic.smooth_noise_host((n, n, n), 1235, 1400 if n == 512 else 300)).to(dev)
ctxs = []
for k in range(3):
    c = _ffi.Context(0)
    c.adopt_device_grid(A.data_ptr(), (n, n, n), keepalive=A)
    c.extract3d(0.0, 1); c.postprocess3d()
    ctxs.append(c)
REPS = 6
def work(c, out, i):
    t0 = time.perf_counter()
    for _ in range(REPS):
        c.extract3d(0.0, 1)
        out[i] = c.postprocess3d()
for nfl in (1, 2, 3):
    best = None
    for rep in range(2):
        out = [None] * nfl
        th = [threading.Thread(target=work, args=(ctxs[i], out, i)) for i in range(nfl)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (REPS * nfl)
        best = dt if best is None else min(best, dt)
    print("%d in flight: %.2f ms per volume (Level 0 + Level 1), %s" % (nfl, best * 1e3, {k: out[0][k] for k in ("n_vertices", "n_triangles", "n_components")}), flush=True)
