#!/usr/bin/env python3
"""what one step costs the HOST: extract3d_async calls enqueued back to back on a tiny grid (the GPU work is negligible, the
queue never fills), wall time per call before any synchronisation"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
A = synthetic.smooth_noise_torch((64, 64, 64), 5, 40, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
ctx.extract3d(0.0, 1)
for name, fn in (("extract3d_async", lambda: ctx.extract3d_async(0.0, 1)),
                 ("adopt + extract3d_async", lambda: (ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A), ctx.extract3d_async(0.0, 1))),
                 ("slab_step (world 1)", lambda: ctx.slab_step(A.data_ptr(), 64, 64, 64, 0, 1, 0.0, 1, keepalive=A))):
    best = 1e9
    for rnd in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        dt = (time.perf_counter() - t0) / 200 * 1e6
        torch.cuda.synchronize()
        best = min(best, dt)
    print("%-28s %.1f us per call on the host" % (name, best))
