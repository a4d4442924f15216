#!/usr/bin/env python3
"""diagnostic: consistency of the Level-1 winding on thick spherical shells of several sizes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from contourist_amd import _ffi
from test_gpu_fullsize import edge_consistency, sphere_field
dev = torch.device("cuda", 0)
for n in (64, 128, 256, 512):
    shape = (n, n, n)
    s = n / 512.0
    c = (250.25 * s, 260.5 * s, 255.75 * s)
    r2 = sphere_field(shape, c, torch, dev)
    A = ((r2 - (200.0 * s) ** 2) * (r2 - (110.0 * s) ** 2) * 1e-4).contiguous()
    ctx = _ffi.Context(0)
    ctx.adopt_device_grid(A.data_ptr(), shape, keepalive=A)
    ctx.extract3d(0.0, 1)
    for flags in (0, 1):
        post = ctx.postprocess3d(flags)
        pts, tris = ctx.download_level1(post)
        m, same, other = edge_consistency(tris)
        t = np.asarray(tris, dtype=np.int64)
        print(n, "flags", flags, post, "manifold", m, "same-direction", same, "non-2 edges", other, flush=True)
    ctx.close()

# the bench field (smooth noise, closed interior) at 256^3 and 384^3
from contourist_amd import synthetic
for n in (256, 384):
    A = synthetic.smooth_noise_torch((n,) * 3, 1235, 1400, dev)
    ctx = _ffi.Context(0)
    ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
    ctx.extract3d(0.0, 1)
    post = ctx.postprocess3d(0)
    pts, tris = ctx.download_level1(post)
    m, same, other = edge_consistency(tris)
    print("noise", n, post, "manifold", m, "same-direction", same, "non-2 edges", other, flush=True)
    ctx.close()
