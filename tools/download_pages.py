#!/usr/bin/env python3
"""the Level-1 download of the 512^3 bench mesh (541 MB) into numpy arrays of three kinds: np.empty (fresh pages, touched first by
the copy), the same arrays a second time (pages in place), and anonymous mappings advised MADV_HUGEPAGE."""
import os, sys, time, mmap, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
n = 512
A = torch.from_numpy(synthetic.smooth_noise_host((n, n, n), 1235, 1400)).cuda()
ctx = _ffi.Context(0)
ctx.adopt_device_grid(A.data_ptr(), (n, n, n), keepalive=A)
ctx.extract3d(0.0, 1)
post = ctx.postprocess3d()
nv, nt = post["n_vertices"], post["n_triangles"]
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "|", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
def dl(p, t):
    t0 = time.perf_counter()
    ctx._check(ctx.lib.cx_level1_download(ctx.handle, p.ctypes.data, t.ctypes.data))
    return (time.perf_counter() - t0) * 1e3
def huge(nbytes):
    size = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20) + (2 << 20)
    m = mmap.mmap(-1, size)
    m.madvise(mmap.MADV_HUGEPAGE)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    off = (-base) % (2 << 20)
    return m, off
for rep in range(3):
    p, t = np.empty((nv, 3)), np.empty((nt, 3), dtype=np.int32)
    a = dl(p, t); b = dl(p, t)
    mp, op = huge(nv * 24); mt, ot = huge(nt * 12)
    ph = np.frombuffer(mp, dtype=np.float64, count=nv * 3, offset=op).reshape(nv, 3)
    th = np.frombuffer(mt, dtype=np.int32, count=nt * 3, offset=ot).reshape(nt, 3)
    c = dl(ph, th); d = dl(ph, th)
    assert np.array_equal(ph, p) and np.array_equal(th, t)
    print("np.empty first %.1f ms, again %.1f ms | hugepage-advised first %.1f ms, again %.1f ms   (%.0f MB)" % (a, b, c, d, (nv * 24 + nt * 12) / 1e6), flush=True)
    del ph, th
