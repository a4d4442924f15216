#!/usr/bin/env python3
"""throughput with 1 / 2 / 3 extractions in flight (one context + stream each) on the bench field"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, dev)
for nctx in (1, 2, 3):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nctx)]
    ctxs = []
    for s in streams:
        c = _ffi.Context(0, stream=s.cuda_stream)
        c.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
        cnt = c.extract3d(0.0, 1)
        c.reserve(int(cnt["n_cells"] * 1.05) + 1024, int(cnt["n_vertices"] * 1.05) + 1024, int(cnt["n_triangles"] * 1.05) + 1024)
        c.extract3d(0.0, 1)
        ctxs.append(c)
    best = 1e9
    for rnd in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 30
        for i in range(steps):
            ctxs[i % nctx].extract3d_async(0.0, 1)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print("in flight %d: %.3f ms per extraction, %.0f Mvoxels/s" % (nctx, best * 1e3, size ** 3 / best / 1e6))
    del ctxs
