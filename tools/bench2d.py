#!/usr/bin/env python3
"""2-D multi-level contour lines on one GPU: samples already in HBM, all levels in one
pass.  Reports M samples/s of the whole extraction (count, scan, emit, growth groups, chains, ranking, ordered output),
with the oracle (pure Python restatement of the reference, one core) timed beside it on a small window."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from contourist_amd import _ffi

n, m, nlev = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 8192, 8)))
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(4242)
# smooth field: coarse noise (one value per 32 samples) interpolated bicubically, plus 2 % fine noise
coarse = torch.randn((1, 1, n // 32 + 2, m // 32 + 2), device=dev, generator=g)
t = torch.nn.functional.interpolate(coarse, size=(n, m), mode="bicubic", align_corners=True)[0, 0]
t = t / t.std() + 0.02 * torch.randn((n, m), device=dev, generator=g)
t = t.contiguous()
values = np.linspace(-1.5, 1.5, nlev)
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
c = _ffi.CxCounts2D()
import ctypes
vals = np.ascontiguousarray(values, dtype=np.float64)
def run(flags=0):
    ctx._check(ctx.lib.cx_contour2d_extract(ctx.handle, t.data_ptr(), 1, n, m, vals.ctypes.data, len(vals), None, 0, flags, None, ctypes.byref(c)))
run()
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
run(_ffi.CX2_ALL_CHAINS)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    run(_ffi.CX2_ALL_CHAINS)
torch.cuda.synchronize()
dt_all = (time.perf_counter() - t0) / K
run()
cpu = None
try:
    from oracle import contour2d as o2
    w = 96
    host = t[:w, :w].cpu().numpy()
    t0 = time.perf_counter()
    npts = 0
    for v in values:
        npts += sum(len(p) for _, p, _ in o2.contours(host, float(v), None, "build"))
    tc = time.perf_counter() - t0
    cpu = {"value": host.size / tc / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%dx%d window, %d levels, %d points in %.1f s, oracle/contour2d.py (pure Python)" % (w, w, nlev, npts, tc)}
except Exception as e:
    cpu = {"error": str(e)}
print(json.dumps({"workload": "%dx%d fp32 smooth noise, %d levels" % (n, m, nlev), "points": c.n_points, "polylines": c.n_chains,
                  "crossings": c.n_pairs, "ms": dt * 1e3, "ms_all_chains": dt_all * 1e3, "Msamples_per_s": n * m / dt / 1e6,
                  "hbm_frac_input_bytes": 4.0 * n * m / dt / 8e12, "cpu_baseline": cpu}))
