#!/usr/bin/env python3
"""config 4: the 4-D march with 1..4 extractions in flight (cx_extract4d_async / cx_counts4d_get on that many contexts)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
shape = (128, 128, 128, 64)
A = synthetic.moving_blobs_torch(shape, 1236, torch.device("cuda", 0))
ctxs = []
for k in range(4):
    c = _ffi.Context(0)
    c.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
    c.extract4d(0.5, 1)
    ctxs.append(c)
for nfl in (1, 2, 3, 4):
    pair = ctxs[:nfl]
    best = None
    for rep in range(3):
        K = 24
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            c = pair[i % nfl]
            if i >= nfl:
                c.counts4d()
            c.extract4d_async(0.5, 1)
        for c in pair:
            c.counts4d()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        best = dt if best is None else min(best, dt)
    print("%d in flight: %.4f ms per extraction (frac %.3f)" % (nfl, best * 1e3, 4 * A.numel() / best / 8e12), flush=True)
