#!/usr/bin/env python3
"""config 4 of BASELINE.json: 128x128x128x64 fp32 4-D field, marching pentatopes on one GPU.
Reports M hyper-voxels/s of the Level-0 march (classify + tetrahedra emit) and of find_tetrahedra."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic

shape = tuple(int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 128, 128, 64)))
dev = torch.device("cuda", 0)
A = synthetic.moving_blobs_torch(shape, 1236, dev)
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
v = 0.5
c = ctx.extract4d(v, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    c = ctx.extract4d(v, 1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
# two extractions in flight (two contexts on the same field, cx_extract4d_async / cx_counts4d_get): what the 3-D bench line reports
ctx2 = _ffi.Context(0)
ctx2.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
ctx2.extract4d(v, 1)
pair = [ctx, ctx2]
K2 = 12
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K2):              # K2 extractions start and end inside the timed region
    c_ = pair[i & 1]
    if i >= 2:
        c2f = c_.counts4d()
    c_.extract4d_async(v, 1)
for c_ in pair:
    c2f = c_.counts4d()
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / K2
assert c2f == c
ctx2.close()
t0 = time.perf_counter()
post = ctx.postprocess4d(100)
torch.cuda.synchronize()
dp_first = time.perf_counter() - t0          # includes the one-time device allocations of the post-pass buffers
t0 = time.perf_counter()
post = ctx.postprocess4d(100)                # again on the same Level-0 mesh, buffers in place
torch.cuda.synchronize()
dp = time.perf_counter() - t0
# morph triangles (B4/B5) and the per-t surfaces (B6) at n3 times: "per-t isosurface stream"
t0 = time.perf_counter()
mt = ctx.morph_triangles()
torch.cuda.synchronize()
dm_first = time.perf_counter() - t0          # includes the one-time device allocations of the post-pass buffers
out = __import__("numpy").zeros(8, dtype="int64")
t0 = time.perf_counter()
ctx._check(ctx.lib.cx_morph_triangles(ctx.handle, out.ctypes.data))     # again, buffers in place, nothing downloaded
torch.cuda.synchronize()
dm = time.perf_counter() - t0
ts = [float(x) for x in torch.linspace(float(mt[0][:, 3].min()), float(mt[0][:, 3].max()), shape[3])] if len(mt[0]) else []
ntris_t = 0
t0 = time.perf_counter()
for tt in ts:
    ntris_t += ctx.morph_eval(tt, download=False)[1]
torch.cuda.synchronize()
de = time.perf_counter() - t0
# the same stream in ONE call (cx_morph_eval_many): warm once (buffers), then timed
import numpy as np
cm = ctx.morph_eval_many(ts, download=False) if ts else np.zeros((0, 2), dtype=np.int64)
torch.cuda.synchronize()
t0 = time.perf_counter()
KM = 5
for _ in range(KM):
    cm = ctx.morph_eval_many(ts, download=False) if ts else cm
torch.cuda.synchronize()
dem = (time.perf_counter() - t0) / KM
assert int(cm[:, 1].sum()) == ntris_t, (int(cm[:, 1].sum()), ntris_t)
# ... and with all surfaces brought to the host in one transfer (cx_morph_eval_many_download_all)
_s = ctx.morph_eval_many(ts) if ts else []
torch.cuda.synchronize()
t0 = time.perf_counter()
_s = ctx.morph_eval_many(ts) if ts else []
torch.cuda.synchronize()
dem_dl = time.perf_counter() - t0
dl_bytes = sum(p_.nbytes + t_.nbytes for p_, t_ in _s)
del _s
n = A.numel()
# CPU baseline: the oracle's C restatement (1 thread) on a slab of the same field
cpu = None
try:
    from oracle import level0_4d
    planes = min(12, shape[0])
    lo = (shape[0] - planes) // 2     # a slab through the middle, where the surface is
    host = A[lo:lo + planes].contiguous().cpu().numpy()
    t0 = time.perf_counter()
    O = level0_4d.march4d(host, v, diag_mode=1)
    tc = time.perf_counter() - t0
    cpu = {"value": host.size / tc / 1e6, "unit": "Mhypervoxels/s", "cores": 1, "kind": "port",
           "sample": "planes %d:%d of the field (%d samples, %d tetrahedra) in %.1f s, oracle/march4d_oracle.c single thread" % (lo, lo + planes, host.size, len(O["tets"]), tc)}
except Exception as e:   # the oracle is test infrastructure; the bench line stands without it
    cpu = {"error": str(e)}
print(json.dumps({"workload": "%dx%dx%dx%d fp32, two moving blobs + noise, v=%g" % (shape + (v,)), "counts": c, "post": post,
                  "level0_ms": dt * 1e3, "level0_two_in_flight_ms": dt2 * 1e3, "hbm_frac_input_bytes_two_in_flight": 4 * n / dt2 / 8e12, "Mhypervoxels_per_s": n / dt / 1e6, "hbm_frac_input_bytes": 4 * n / dt / 8e12,
                  "postprocess_ms": dp * 1e3, "postprocess_first_call_ms": dp_first * 1e3, "morph_triangles_ms": dm * 1e3, "morph_triangles_first_call_with_download_ms": dm_first * 1e3, "morph_triangles": int(len(mt[2])),
                  "per_t_surfaces": {"times": len(ts), "triangles": int(ntris_t), "ms": de * 1e3,
                                     "Mtriangles_per_s": ntris_t / de / 1e6 if de > 0 else 0.0},
                  "per_t_surfaces_one_call": {"times": len(ts), "triangles": int(cm[:, 1].sum()), "points": int(cm[:, 0].sum()), "ms": dem * 1e3,
                                              "Mtriangles_per_s": int(cm[:, 1].sum()) / dem / 1e6 if dem > 0 else 0.0,
                                              "ms_with_download": dem_dl * 1e3, "download_MB": dl_bytes / 1e6},
                  "cpu_baseline": cpu}))
