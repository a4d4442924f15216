#!/usr/bin/env python3
"""where a vertex-stage wave spends its time (build with CX_EXTRA_FLAGS=-DCX_S3_STAMPS; 100 MHz s_memrealtime ticks)"""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
size = 512
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
print(ctx.extract3d(0.0, 1))
nw = 2560 * 4 * 8
ctx._check(ctx.lib.cx_debug_stamps(ctx.handle, nw, None))
for _ in range(3):
    ctx.extract3d_async(0.0, 1)
ctx.synchronize()
buf = np.zeros(nw, dtype=np.uint64)
ctx._check(ctx.lib.cx_debug_stamps(ctx.handle, nw, buf.ctypes.data))
S = buf.reshape(-1, 8).astype(np.int64)
S = S[S[:, 0] > 0]
t0 = S[:, 0].min()
life = (S[:, 1] - S[:, 0]) / 100.0
print("waves", len(S), "kernel span %.1f us" % ((S[:, 1].max() - t0) / 100.0))
print("wave life us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
print("start us p50 %.1f p90 %.1f max %.1f | end p10 %.1f p50 %.1f p90 %.1f" % (*np.percentile((S[:, 0] - t0) / 100.0, [50, 90]), (S[:, 0].max() - t0) / 100.0, *np.percentile((S[:, 1] - t0) / 100.0, [10, 50, 90])))
r = S[:, 6].astype(float)
print("rounds per wave: mean %.1f p50 %.0f p90 %.0f max %.0f" % (r.mean(), *np.percentile(r, [50, 90]), r.max()))
tot = S[:, 2:6].sum(axis=0) / 100.0
print("us per round: front(issue) %.2f | corners+interp %.2f | wait(pin) %.2f | stores %.2f | sum %.2f" % (*(tot / r.sum()), tot.sum() / r.sum()))
T = (S[:, 1].max() - t0)
for k in range(10):
    t = t0 + T * (k + 0.5) / 10
    print("t=%3d%% running %5d" % (10 * k + 5, int(((S[:, 0] <= t) & (S[:, 1] > t)).sum())))
