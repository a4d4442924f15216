#!/usr/bin/env python3
"""K1/K2 timing of the full path (median of rounds), for A/B builds and modes."""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")   # ablation flags and tuning knobs are refused otherwise
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
extra = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0   # debug flag bits added to both modes
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
print(ctx.extract3d(0.0, 1))
for fl, name in ((1 | extra, "full"), (1 | 0x10000 | 0x800000, "phaseA")):
    r = []
    for rnd in range(7):
        ctx.extract3d_async(0.0, fl)
        ctx.timing_enable(True)
        for _ in range(5):
            ctx.extract3d_async(0.0, fl)
        t = ctx.timing_read(); ctx.timing_enable(False)
        r.append(tuple(t[k] / t["n"] for k in ("classify_ms", "emit_ms", "stream_ms", "scan_ms", "cells_ms")))
    med = [sorted(x[c] for x in r)[3] for c in range(5)]
    print("%s %-8s K1 %.3f (stream %.3f scan %.3f verts %.3f) | K2 %.3f | sum %.3f" % (
        os.environ.get("TAG", ""), name, med[0], med[2], med[3], med[4], med[1], med[0] + med[1]))
