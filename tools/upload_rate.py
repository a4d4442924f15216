#!/usr/bin/env python3
"""cx_grid_upload of a 512^3 fp32 volume from a pageable numpy array (the boundary's host-buffer case) next to a pinned torch copy:
9.5 ms = 56 GB/s either way on the pool's boxes (third session of round 4) -- the runtime's own staging is as fast as pinned memory here."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi
H = np.random.RandomState(1).standard_normal((512, 512, 512)).astype(np.float32)
ctx = _ffi.Context(0)
ctx.upload_grid(H); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); ctx.upload_grid(H); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("cx_grid_upload 537 MB from a pageable numpy array: %.1f ms = %.1f GB/s" % (dt * 1e3, H.nbytes / dt / 1e9))
P = torch.from_numpy(H).pin_memory()
for rep in range(2):
    t0 = time.perf_counter(); D = P.cuda(non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("torch pinned -> device: %.1f ms = %.1f GB/s" % (dt * 1e3, H.nbytes / dt / 1e9))
