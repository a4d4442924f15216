#!/usr/bin/env python3
"""time of the 4-D Level 0 (config 4) per call, a few calls; for A/B of the bitmap kernel's launch geometry (CX4_SB_WGS)"""
import os, sys, time
os.environ.setdefault("CX_DEBUG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
shape = (128, 128, 128, 64)
A = synthetic.moving_blobs_torch(shape, 1236, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
c = ctx.extract4d(0.5, 1)
res = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ctx.extract4d(0.5, 1)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 10 * 1e3)
print(os.environ.get("CX4_SB_WGS", "-"), "level0 ms per call: min %.4f med %.4f" % (min(res), sorted(res)[2]), c)
