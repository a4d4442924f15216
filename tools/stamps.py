#!/usr/bin/env python3
"""where a classify wave spends its life (diagnostic s_memtime stamps; shares, not absolute times)"""
import os, sys
os.environ.setdefault("CX_DEBUG", "1")   # ablation flags and tuning knobs are refused otherwise
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
size = 512
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
print(ctx.extract3d(0.0, 1))
nw = 4096 * 4 * 4
ctx._check(ctx.lib.cx_debug_stamps(ctx.handle, nw, None))
ctx.extract3d_async(0.0, 1); ctx.synchronize()
buf = np.zeros(nw, dtype=np.uint64)
ctx._check(ctx.lib.cx_debug_stamps(ctx.handle, nw, buf.ctypes.data))
S = buf.reshape(-1, 4).astype(np.int64)
S = S[S[:, 0] > 0]
t0 = S[:, 0].min()
d = S[:, 1] - S[:, 0]
print("stream waves", len(S), "span", S[:, 1].max() - t0)
print("wave life: mean %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (d.mean(), *np.percentile(d, [10, 50, 90, 99]), d.max()))
st = S[:, 0] - t0
en = S[:, 1] - t0
print("start p50 %d p90 %d max %d | end p50 %d p90 %d p99 %d max %d" % (np.median(st), np.percentile(st, 90), st.max(), np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max()))
# waves still running over time (20 bins)
T = en.max()
for k in range(0, 20):
    t = T * (k + 0.5) / 20
    print("t=%5.1f%%  running %5d" % (100 * (k + 0.5) / 20, int(((st <= t) & (en > t)).sum())))
