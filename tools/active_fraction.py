#!/usr/bin/env python3
"""Level-0 kernel times of the staged pipeline against the ACTIVE FRACTION of the volume: the bench generator at several smoothing
depths (fewer passes = a rougher field = more surface), 512^3, isovalue 0.  Separates what the pass over the samples costs (4 B per
sample, whatever the field) from what a surface cell costs: time = a + b x cells."""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
rows = []
for passes in (11200, 5600, 2800, 1400, 700, 350, 175):
    A = synthetic.smooth_noise_torch((size,) * 3, 1235, passes, dev)
    ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
    c = ctx.extract3d(0.0, 1)
    ctx.reserve(int(c["n_cells"] * 1.05) + 1024, int(c["n_vertices"] * 1.05) + 1024, int(c["n_triangles"] * 1.05) + 1024)
    res = []
    for rnd in range(5):
        ctx.extract3d_async(0.0, 1)
        ctx.timing_enable(True)
        for k in range(6):
            ctx.extract3d_async(0.0, 1)
        t = ctx.timing_read(); ctx.timing_enable(False)
        res.append(tuple(t[k] / t["n"] for k in ("total_ms", "stream_ms", "scan_ms", "cells_ms", "emit_ms")))
    med = [sorted(x[i] for x in res)[2] for i in range(5)]
    frac = c["n_border_voxels"] / float((size - 1) ** 3)
    rows.append((passes, frac, c["n_cells"], c["n_vertices"], c["n_triangles"], med))
    print("passes %5d  active %.4f  cells %9d  V %9d  T %9d | total %.3f  stream %.3f  scan %.3f  vertices %.3f  triangles %.3f ms  | %.0f G voxels/s, frac of 8 TB/s %.3f"
          % (passes, frac, c["n_cells"], c["n_vertices"], c["n_triangles"], *med, size ** 3 / med[0] / 1e6, 4.0 * size ** 3 / (med[0] * 1e-3) / 8e12), flush=True)
    ctx.close(); del A; torch.cuda.empty_cache()
# least squares: total = a + b * cells
import numpy as np
x = np.array([r[2] for r in rows], dtype=np.float64); y = np.array([r[5][0] for r in rows])
b, a = np.polyfit(x, y, 1)
print("fit: total_ms = %.4f + %.4f per million surface cells  (one stream, kernel durations)" % (a, b * 1e6))
