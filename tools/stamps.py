#!/usr/bin/env python3
"""where a classify wave spends its life (diagnostic s_memtime stamps; shares, not absolute times)"""
import os, sys
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
print("waves", len(S), "kernel span (ticks)", S[:, 3].max() - t0)
for name, a, b in (("phaseA", 0, 1), ("barrier+reserve", 1, 2), ("emit", 2, 3), ("life", 0, 3)):
    d = S[:, b] - S[:, a]
    print("%-16s mean %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f" % (name, d.mean(), np.median(d), np.percentile(d, 90), d.max()))
st = S[:, 0] - t0
print("start time p50 %d p90 %d max %d" % (np.median(st), np.percentile(st, 90), st.max()))
