#!/usr/bin/env python3
"""time of the Level-1 post-passes (weld, tiny, clean, orient) after a Level-0 extraction"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
c = ctx.extract3d(0.0, 1)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    post = ctx.postprocess3d(0)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); pts, tris = ctx.download_level1(post); td = time.perf_counter() - t0
print(json.dumps({"size": size, "level0": c, "level1": post, "postprocess_ms": [round(t * 1e3, 2) for t in ts], "download_ms": round(td * 1e3, 1)}))
