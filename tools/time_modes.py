#!/usr/bin/env python3
"""per-kernel timing (median of rounds, interleaved in one process) of the Level-0 paths: fused / staged emit,
CPython-order / canonical diagonals"""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")   # ablation flags and tuning knobs are refused otherwise
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
grids = [synthetic.smooth_noise_torch((size,) * 3, 1235 + r, 1400, torch.device("cuda", 0)) for r in range(2)]
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(grids[0].data_ptr(), tuple(grids[0].shape), keepalive=grids[0])
c = ctx.extract3d(0.0, 1)
print(c, "path", ctx.level0_path())
ctx.reserve(int(c["n_cells"] * 1.1), int(c["n_vertices"] * 1.1), int(c["n_triangles"] * 1.1))
modes = [(1 | 0x400, "fused cpython"), (0x400, "fused canonical"), (1 | 0x200, "staged cpython"), (0x200, "staged canonical")]
if len(sys.argv) > 2 and sys.argv[2] == "quick":
    modes = [(1 | 0x400, "fused"), (0x400, "fused canon"), (1, "staged"), (0, "staged canon")]
if len(sys.argv) > 2 and sys.argv[2] == "tiled":
    modes = [(1 | 0x800, "tiled"), (1 | 0x200, "staged"), (0x800, "tiled canon"), (0x200, "staged canon")]
if len(sys.argv) > 2 and sys.argv[2] == "tiled_ablate":
    F = 0x800
    modes = [(F | 1, "tiled"), (F, "tiled canon"), (F | 0x1000000, "canon no_vloads"), (F | 0x400000, "canon no_tris"), (F | 0x80000, "canon no_verts"),
             (F | 0x200000, "canon no_lookup"), (F | 0x1000000 | 0x80000, "canon no_vloads no_verts"), (F | 0x1000000 | 0x80000 | 0x400000, "canon alu+lds only")]
if len(sys.argv) > 2 and sys.argv[2] == "staged":
    modes = [(1, "staged"), (0, "staged canon")]
if len(sys.argv) > 2 and sys.argv[2] == "hash":
    modes = [(1, "staged"), (0, "staged canon"), (1 | 0x200000, "cpython no_lookup"), (0x200000, "canon no_lookup")]
if len(sys.argv) > 2 and sys.argv[2] == "fused":
    modes = [(1 | 0x400, "fused"), (0x400, "fused canon")]
if len(sys.argv) > 2 and sys.argv[2] == "ablate_staged":
    S = 0x200
    modes = [(S | 1, "staged"), (S, "staged canon"), (S | 0x40000, "canon no_celltab"), (S | 0x40000 | 0x100000, "canon no_celltab no_cells"),
             (S | 0x200000, "canon no_lookup"), (S | 0x400000, "canon no_tris"), (S | 0x200000 | 0x400000, "canon no_lookup no_tris"),
             (S | 0x80000, "canon no_verts"), (S | 0x1000000, "canon no_vloads")]
if len(sys.argv) > 2 and sys.argv[2] == "ablate":
    F = 0x400
    modes = [(F | 1, "fused"), (F, "fused canon"), (F | 0x200000, "canon no_lookup"), (F | 0x1000000, "canon no_vloads"),
             (F | 0x80000, "canon no_verts"), (F | 0x400000, "canon no_tris"), (F | 0x80000 | 0x400000, "canon no_stores"),
             (F | 0x200000 | 0x1000000, "canon no_loads"), (F | 0x200000 | 0x1000000 | 0x80000 | 0x400000, "canon alu only")]
res = {name: [] for _, name in modes}
for rnd in range(7):
    for fl, name in modes:
        ctx.adopt_device_grid(grids[rnd % 2].data_ptr(), tuple(grids[0].shape), keepalive=grids[rnd % 2])
        ctx.extract3d_async(0.0, fl)
        ctx.timing_enable(True)
        for k in range(6):
            g = grids[(rnd + k) % 2]
            ctx.adopt_device_grid(g.data_ptr(), tuple(g.shape), keepalive=g)
            ctx.extract3d_async(0.0, fl)
        t = ctx.timing_read(); ctx.timing_enable(False)
        res[name].append(tuple(t[k] / t["n"] for k in ("total_ms", "stream_ms", "scan_ms", "cells_ms", "emit_ms")))
for _, name in modes:
    r = res[name]
    med = [sorted(x[c] for x in r)[len(r) // 2] for c in range(5)]
    print("%s %-16s total %.3f | stream %.3f scan %.3f verts/mesh %.3f tris %.3f" % (os.environ.get("TAG", ""), name, *med))
