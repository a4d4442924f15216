#!/usr/bin/env python3
"""the seeded selection (cx_select_seeded3d_ex: the reference's find_initial_voxels + expand_voxels, tetrahedral.py:396-463) on the
512^3 bench field: one crossing segment as seed, then 64 of them; warm calls after one extraction."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = synthetic.smooth_noise_host((n, n, n), 1235, 1400 if n == 512 else 300)
A = torch.from_numpy(H).cuda()
ctx = _ffi.Context(0)
ctx.adopt_device_grid(A.data_ptr(), (n, n, n), keepalive=A)
counts = ctx.extract3d(0.0, 1)
# crossing segments along the last axis: (i, j, k) -> (i, j, k + 1) with a strict sign change
neg = H < 0
cr = np.argwhere(neg[:, :, :-1] != neg[:, :, 1:])
rng = np.random.RandomState(5)
for nseed in (1, 64):
    pick = cr[rng.choice(len(cr), size=nseed, replace=False)]
    eps = [((int(i), int(j), int(k)), (int(i), int(j), int(k) + 1)) for (i, j, k) in pick]
    r = ctx.select_seeded(eps)
    torch.cuda.synchronize()
    best = None
    for rep in range(5):
        t0 = time.perf_counter()
        r = ctx.select_seeded(eps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("%d seed segment(s): %.2f ms  %s of %d triangles (%d active cells)" % (nseed, best * 1e3, r, counts["n_triangles"], counts["n_cells"]), flush=True)
t0 = time.perf_counter(); post = ctx.postprocess3d(); torch.cuda.synchronize()
print("Level 1 of the selection: %.2f ms %s" % ((time.perf_counter() - t0) * 1e3, post))
