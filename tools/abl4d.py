#!/usr/bin/env python3
"""ablation timings of the 4-D Level-0 kernels (debug knob CX4_ABL, one process per setting)."""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from contourist_amd import _ffi, synthetic
shape = (128, 128, 128, 64)
A = synthetic.moving_blobs_torch(shape, 1236, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
flags = int(sys.argv[1])
ctx.extract4d(0.5, flags); ctx.extract4d(0.5, flags)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    c = ctx.extract4d(0.5, flags)
torch.cuda.synchronize()
print((time.perf_counter() - t0) / 10 * 1e3, c)
''' % ROOT
for name, abl, flags in (("full cpython", 0, 1), ("canonical diagonals", 0, 0), ("no tet stores", 2, 1), ("no celltab gathers", 4, 1),
                         ("no stores, no gathers", 6, 1), ("no stores/gathers/hash", 6, 0)):
    env = dict(os.environ, CX_DEBUG="1", CX4_ABL=str(abl))
    out = subprocess.run([sys.executable, "-c", CHILD, str(flags)], env=env, capture_output=True, text=True)
    print("%-28s %s %s" % (name, out.stdout.strip(), out.stderr.strip()[-300:]), flush=True)
