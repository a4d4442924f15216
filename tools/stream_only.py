#!/usr/bin/env python3
"""stream kernel alone: bench field at an isovalue nothing crosses vs the bench isovalue"""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")   # ablation flags and tuning knobs are refused otherwise
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
print(ctx.extract3d(0.0, 1))
for value in (0.0, 1.0e6):
    r = []
    for rnd in range(7):
        ctx.extract3d_async(value, 1)
        ctx.timing_enable(True)
        for _ in range(5):
            ctx.extract3d_async(value, 1)
        t = ctx.timing_read(); ctx.timing_enable(False)
        r.append(tuple(t[k] / t["n"] for k in ("stream_ms", "scan_ms", "cells_ms", "emit_ms")))
    med = [sorted(x[c] for x in r)[3] for c in range(4)]
    print("value %g: stream %.3f scan %.3f verts %.3f tris %.3f" % (value, *med))
