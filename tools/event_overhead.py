#!/usr/bin/env python3
"""wall time per extraction with and without the per-kernel HIP events"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
dev = torch.device("cuda", 0)
A = synthetic.smooth_noise_torch((512,) * 3, 1235, 1400, dev)
c = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
c.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
c.extract3d(0.0, 1)
for rnd in range(3):
    for on in (False, True):
        c.timing_enable(on)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40):
            c.extract3d_async(0.0, 1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
        extra = ""
        if on:
            t = c.timing_read(); extra = " kernels sum %.3f" % (t["total_ms"] / t["n"])
        print("events %-5s %.3f ms per extraction%s" % (on, dt * 1e3, extra))
