#!/usr/bin/env python3
"""diagnostic: winding consistency of the morph triangles, evaluated as surfaces at several times"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from contourist_amd import _ffi
from test_gpu_fullsize import edge_consistency
dev = torch.device("cuda", 0)
for shape in ((32, 32, 32, 16), (64, 64, 64, 32), (128, 128, 128, 64)):
    ax = [torch.arange(n, device=dev, dtype=torch.float32) / (n - 1) for n in shape]
    X, Y, Z, T = torch.meshgrid(*ax, indexing="ij")
    A = torch.exp(-(((X - 0.30 - 0.35 * T) ** 2 + (Y - 0.35 - 0.2 * T) ** 2 + (Z - 0.5) ** 2) / (2 * 0.12 ** 2))) + \
        torch.exp(-(((X - 0.70 + 0.30 * T) ** 2 + (Y - 0.65 + 0.2 * T) ** 2 + (Z - 0.45 - 0.1 * T) ** 2) / (2 * 0.10 ** 2)))
    for axis in range(4):
        for idx in (0, 1, -1, -2):
            A.select(axis, idx).fill_(0.0)
    A = A.contiguous()
    ctx = _ffi.Context(0)
    ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
    c = ctx.extract4d(0.5, 1)
    post = ctx.postprocess4d(100)
    mt = ctx.morph_triangles()
    tmin, tmax = float(mt[0][:, 3].min()), float(mt[0][:, 3].max())
    for frac in (0.13, 0.37, 0.52, 0.81):
        t = tmin + frac * (tmax - tmin)
        pts, tris = ctx.morph_eval(t)
        m, same, other = edge_consistency(tris) if len(tris) else (0, 0, 0)
        print(shape, "t=%.3f" % t, "triangles", len(tris), "manifold", m, "same-direction", same, "non-2", other, flush=True)
    ctx.close()
    del A, X, Y, Z, T
    torch.cuda.empty_cache()
