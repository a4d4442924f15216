#!/usr/bin/env python3
"""what one rank of an N-way strong-scaling run does: Level 0 of a slab of the 512^3 bench field (own planes + 1 halo plane),
per-kernel medians (HIP events); python tools/slab_time.py [planes ...]"""
import os, sys, time
os.environ.setdefault("CX_DEBUG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
A = synthetic.smooth_noise_torch((512,) * 3, 1235, 1400, torch.device("cuda", 0))
base = None
for planes in [int(x) for x in (sys.argv[1:] or (512, 256, 128, 64))]:
    lo = (512 - planes) // 2
    S = A[lo:lo + planes + (1 if lo + planes < 512 else 0)].contiguous()
    ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.adopt_device_grid(S.data_ptr(), tuple(S.shape), keepalive=S)
    c = ctx.extract3d(0.0, 1)
    for _ in range(5): ctx.extract3d_async(0.0, 1)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20): ctx.extract3d_async(0.0, 1)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    ctx.timing_enable(True)
    for _ in range(10): ctx.extract3d_async(0.0, 1)
    t = ctx.timing_read(); ctx.timing_enable(False)
    k = {n: t[n] / t["n"] for n in ("stream_ms", "scan_ms", "cells_ms", "emit_ms")}
    base = base if planes != 512 else min(ts)
    print("planes %3d: %.4f ms/extraction (512 planes / this = %.2f x) | stream %.3f scan %.3f verts %.3f tris %.3f | triangles %d" % (
        planes, min(ts), (base / min(ts)) if base else 0.0, k["stream_ms"], k["scan_ms"], k["cells_ms"], k["emit_ms"], c["n_triangles"]), flush=True)
