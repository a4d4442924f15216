#!/usr/bin/env python3
"""Does it matter where the samples come from?  Per-kernel times of the staged pipeline on a 256^3 field (67 MB) extracted over and
over from ONE buffer (the field stays in the 256 MB Infinity Cache between the stream kernel and the vertex stage) against the same
extractions cycling through 8 buffers (each extraction's samples come from HBM): what the vertex stage's sample gathers would gain
if they followed the stream kernel closely enough to hit the cache."""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
passes = int(sys.argv[2]) if len(sys.argv) > 2 else (700 if size <= 256 else 1400)
dev = torch.device("cuda", 0)
base = synthetic.smooth_noise_torch((size,) * 3, 1235, passes, dev)
grids = [base] + [base.clone() for _ in range(7)]
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(base.data_ptr(), tuple(base.shape), keepalive=base)
c = ctx.extract3d(0.0, 1)
print(c)
ctx.reserve(int(c["n_cells"] * 1.1), int(c["n_vertices"] * 1.1), int(c["n_triangles"] * 1.1))
for nrot in (1, 8, 1, 8):
    res = []
    for rnd in range(5):
        for k in range(8):
            g = grids[k % nrot]
            ctx.adopt_device_grid(g.data_ptr(), tuple(g.shape), keepalive=g)
            ctx.extract3d_async(0.0, 1)
        ctx.timing_enable(True)
        for k in range(16):
            g = grids[k % nrot]
            ctx.adopt_device_grid(g.data_ptr(), tuple(g.shape), keepalive=g)
            ctx.extract3d_async(0.0, 1)
        t = ctx.timing_read(); ctx.timing_enable(False)
        res.append(tuple(t[k] / t["n"] for k in ("total_ms", "stream_ms", "scan_ms", "cells_ms", "emit_ms")))
    med = [sorted(x[c] for x in res)[len(res) // 2] for c in range(5)]
    print("%d^3, %d buffer(s): total %.4f | stream %.4f scan %.4f verts %.4f tris %.4f" % (size, nrot, *med))
