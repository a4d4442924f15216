#!/usr/bin/env python3
"""where the waves of the tile kernel (cx_k_tile_emit) spend their lives: per wave start / pass A done / end (10 ns ticks)"""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
size = 512
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
FL = 1 | 0x800
print(ctx.extract3d(0.0, FL), "path", ctx.level0_path())
nblocks = 3072 + 64
nw = nblocks * 16 + nblocks * 2 * 16
ctx._check(ctx.lib.cx_debug_stamps(ctx.handle, nw, None))
for rep in range(2):
    ctx.extract3d_async(0.0, FL); ctx.synchronize()
buf = np.zeros(nw, dtype=np.uint64)
ctx._check(ctx.lib.cx_debug_stamps(ctx.handle, nw, buf.ctypes.data))
nb = int(os.environ.get("NBLOCKS", "3072"))
K = buf[nb * 16:nb * 16 + nb * 2 * 16].reshape(-1, 4).astype(np.int64)     # per wave
K = K[K[:, 0] > 0]
t0 = K[:, 0].min()
tick = 0.01   # us
span = (K[:, 2].max() - t0) * tick
print("waves that started", len(K), "span %.1f us" % span)
live = K[K[:, 3] > 0]
print("waves with entries", len(live), "entries mean %.0f p50 %.0f p90 %.0f max %d" % (live[:, 3].mean(), *np.percentile(live[:, 3], [50, 90]), live[:, 3].max()))
dA = (live[:, 1] - live[:, 0]) * tick
dB = (live[:, 2] - live[:, 1]) * tick
print("pass A (incl. set-up): mean %.1f p50 %.1f p90 %.1f max %.1f us" % (dA.mean(), *np.percentile(dA, [50, 90]), dA.max()))
print("pass B (incl. barriers): mean %.1f p50 %.1f p90 %.1f max %.1f us" % (dB.mean(), *np.percentile(dB, [50, 90]), dB.max()))
r = np.maximum(1, (live[:, 3] + 63) // 64)
print("per round of 64 entries: pass A %.2f us, pass B %.2f us (mean over waves of time / rounds)" % ((dA / r).mean(), (dB / r).mean()))
empty = K[K[:, 3] == 0]
if len(empty):
    print("waves without entries: %d, life mean %.1f us" % (len(empty), ((empty[:, 2] - empty[:, 0]) * tick).mean()))
st, en = (K[:, 0] - t0) * tick, (K[:, 2] - t0) * tick
mid = (K[:, 1] - t0) * tick
for k in range(20):
    t = span * (k + 0.5) / 20
    print("t=%6.1f us  waves running %5d  (in pass A %5d, in pass B %5d)" % (t, int(((st <= t) & (en > t)).sum()), int(((st <= t) & (mid > t)).sum()), int(((mid <= t) & (en > t)).sum())))
