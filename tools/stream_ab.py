#!/usr/bin/env python3
"""stream kernel alone (never runs the later stages: safe for ablation builds whose queues are garbage)"""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")   # ablation flags and tuning knobs are refused otherwise
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
FL = 1 | 0x10000 | 0x800000
out = []
for value, FL in ((0.0, FL), (1.0e6, FL), (0.0, FL | 0x2000000)):
    r = []
    for rnd in range(7):
        ctx.extract3d_async(value, FL)
        ctx.timing_enable(True)
        for _ in range(5):
            ctx.extract3d_async(value, FL)
        t = ctx.timing_read(); ctx.timing_enable(False)
        r.append(t["stream_ms"] / t["n"])
    out.append(sorted(r)[3])
print("%s stream: active field %.3f ms, no crossings %.3f ms, active without tolerance path %.3f ms" % (os.environ.get("TAG", ""), out[0], out[1], out[2]))
