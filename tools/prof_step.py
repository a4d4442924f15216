#!/usr/bin/env python3
"""a few Level-0 extractions of the bench field, for rocprofv3 counter passes"""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")   # ablation flags and tuning knobs are refused otherwise
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
flags = int(sys.argv[2], 0) if len(sys.argv) > 2 else 1
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
print(ctx.extract3d(0.0, 1))
for _ in range(4):
    ctx.extract3d_async(0.0, flags)
ctx.synchronize()
