#!/usr/bin/env python3
"""A/B timing of 4-D Level-0 builds (tools/variants.sh): python tools/ab4d.py v0 v1 ... ; each in its own process,
per-kernel times from HIP events are not available through the C ABI, so: whole extract4d, cpython and canonical diagonals."""
import os, subprocess, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from contourist_amd import _ffi, synthetic
shape = (128, 128, 128, 64)
A = synthetic.moving_blobs_torch(shape, 1236, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
out = []
for flags in (1, 0):
    ctx.extract4d(0.5, flags); ctx.extract4d(0.5, flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        c = ctx.extract4d(0.5, flags)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 20 * 1e3)
print("cpython %%.4f canonical %%.4f ms  tets %%d" %% (out[0], out[1], c["n_tetrahedra"]))
''' % ROOT
for rep in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, CX_DEBUG="1")
        if name != "default":
            env["CX_LIB_PATH"] = os.path.join(ROOT, "contourist_amd", "lib", "variants", "lib_%s.so" % name)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("%-10s %s %s" % (name, out.stdout.strip(), out.stderr.strip()[-200:].replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory", "")), flush=True)
