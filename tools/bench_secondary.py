#!/usr/bin/env python3
"""secondary calls of the 3-D path on the 512^3 bench mesh, warm: Level 1 with Laplacian smoothing (tetrahedral.py:329-351), Level 1 of
the assembled mesh (cx_postprocess3d_mesh: what the slab paths call), cx_surface_geometry on a caller's mesh (clean + orient),
the float64 Level-0 points, the mesh file written from the device buffers."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = torch.from_numpy(synthetic.smooth_noise_host((n, n, n), 1235, 1400 if n == 512 else 300)).cuda()
ctx = _ffi.Context(0)
ctx.adopt_device_grid(A.data_ptr(), (n, n, n), keepalive=A)
counts = ctx.extract3d(0.0, 1)
def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("%-52s %8.2f ms" % (name, best * 1e3), flush=True)
    return r
post = timed("cx_postprocess3d (Level 1)", lambda: ctx.postprocess3d())
timed("cx_postprocess3d_ex, smooth = 0.5", lambda: ctx.postprocess3d(0, 0.5))
xyz = timed("cx_level0_points_f64 (12.4 M points to the host)", lambda: ctx.level0_points_f64(counts))
_x, keys, tris = ctx.download_level0(counts)
order = np.argsort(keys.astype(np.int64), kind="stable")
rank = np.empty(len(order), dtype=np.int64); rank[order] = np.arange(len(order))
mp, mt = np.ascontiguousarray(xyz[order]), np.ascontiguousarray(rank[tris].astype(np.int32))     # (prepared outside the timed call)
pm0 = timed("cx_postprocess3d_mesh (a mesh of any origin)", lambda: ctx.postprocess3d_mesh(mp, mt, [n - 1] * 3))
pm = timed("cx_postprocess3d_mesh (flagged: a mesh of the march)", lambda: ctx.postprocess3d_mesh(mp, mt, [n - 1] * 3, _ffi.CX_MESH_OF_THE_MARCH))
assert pm == pm0
post = ctx.postprocess3d()
pts, t1 = ctx.download_level1(post)
import ctypes
def sg():
    p2, t2 = pts.copy(), t1.copy()
    nv, nt = ctypes.c_int64(len(p2)), ctypes.c_int64(len(t2))
    t0 = time.perf_counter()
    ctx._check(ctx.lib.cx_surface_geometry(ctx.handle, p2.ctypes.data, ctypes.byref(nv), t2.ctypes.data, ctypes.byref(nt), 1))
    return time.perf_counter() - t0
sg(); print("%-52s %8.2f ms" % ("cx_surface_geometry (caller's mesh: clean + orient)", min(sg(), sg()) * 1e3), flush=True)
ctx.postprocess3d()
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
timed("cx_level1_write ply (541 MB)", lambda: ctx.write_level1(os.path.join(d, "m.ply"), "ply"), reps=2)
os.remove(os.path.join(d, "m.ply")); os.rmdir(d)
print(post, pm)
