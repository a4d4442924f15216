#!/usr/bin/env python3
"""thin slabs: consecutive extractions on ONE stream vs alternating between TWO contexts / streams (independent volumes)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
A = synthetic.smooth_noise_torch((512,) * 3, 1235, 1400, torch.device("cuda", 0))
for planes in [int(x) for x in (sys.argv[1:] or (64, 128, 256, 512))]:
    lo = (512 - planes) // 2
    S = A[lo:lo + planes + (1 if lo + planes < 512 else 0)].contiguous()
    S2 = S.clone()
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs = []
    for k, g in enumerate((S, S2, S)):
        c = _ffi.Context(0, stream=streams[k].cuda_stream)
        c.adopt_device_grid(g.data_ptr(), tuple(g.shape), keepalive=g)
        c.extract3d(0.0, 1)
        ctxs.append(c)
    torch.cuda.synchronize()
    def run(n, which):
        t0 = time.perf_counter()
        for i in range(n):
            ctxs[which[i % len(which)]].extract3d_async(0.0, 1)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    run(20, [0]); run(20, [0, 1])
    one = min(run(40, [0]) for _ in range(5))
    two = min(run(40, [0, 1]) for _ in range(5))
    three = min(run(42, [0, 1, 2]) for _ in range(5))
    print("planes %3d: one stream %.4f ms/extraction, two streams %.4f, three %.4f" % (planes, one, two, three), flush=True)
