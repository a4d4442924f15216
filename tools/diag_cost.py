#!/usr/bin/env python3
"""cost of reproducing the reference's CPython set order (CX_DIAG_CPYTHON310) in the triangle kernel: K2 time with
flags 1 (exact) and 0 (canonical diagonal)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
for fl, name in ((1, "cpython310"), (0, "canonical"), (1, "cpython310"), (0, "canonical")):
    ctx.extract3d(0.0, fl)
    r = []
    for rnd in range(7):
        ctx.extract3d_async(0.0, fl)
        ctx.timing_enable(True)
        for _ in range(5):
            ctx.extract3d_async(0.0, fl)
        t = ctx.timing_read(); ctx.timing_enable(False)
        r.append((t["emit_ms"] / t["n"], t["total_ms"] / t["n"]))
    print("%-11s K2 %.4f ms   total %.4f ms" % (name, sorted(x[0] for x in r)[3], sorted(x[1] for x in r)[3]), flush=True)
