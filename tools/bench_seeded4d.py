#!/usr/bin/env python3
"""the 4-D seeded selection (cx_select_seeded4d_ex) on config 4 (128^3 x 64): one crossing segment as seed; warm calls."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
shape = (128, 128, 128, 64)
A = synthetic.moving_blobs_torch(shape, 1236, torch.device("cuda", 0))
ctx = _ffi.Context(0)
ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
counts = ctx.extract4d(0.5, 1)
H = A[:, :, :, 32].cpu().numpy()
neg = H < 0.5
cr = np.argwhere(neg[:, :, :-1] != neg[:, :, 1:])
i, j, k = [int(x) for x in cr[len(cr) // 2]]
eps = [((i, j, k, 32), (i, j, k + 1, 32))]
r = ctx.select_seeded4d(eps)
torch.cuda.synchronize()
best = None
for rep in range(5):
    t0 = time.perf_counter()
    r = ctx.select_seeded4d(eps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print("1 seed segment: %.2f ms  %s of %d tetrahedra (%d active hyper-voxels)" % (best * 1e3, r, counts["n_tetrahedra"], counts["n_cells"]), flush=True)
