#!/usr/bin/env python3
"""config 4: how long the morph triangles and their segments live (in layers of the time axis), and how wide the windows of ids
are that cx_morph_eval_many tests for one time -- against the triangles that actually exist then."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
shape = (128, 128, 128, 64)
A = synthetic.moving_blobs_torch(shape, 1236, torch.device("cuda", 0))
ctx = _ffi.Context(0)
ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
ctx.extract4d(0.5, 1); ctx.postprocess4d(100)
pts, segs, tris, _ = ctx.morph_triangles()
t = pts[:, 3]
lo, hi = t[segs[:, 0]], t[segs[:, 1]]
tl, th = lo[tris].max(axis=1), hi[tris].min(axis=1)
layer = (t.max() - t.min()) / (shape[3] - 1)
for name, d in (("segments", hi - lo), ("triangles", th - tl)):
    d = d / layer
    print(name, len(d), "life in layers: mean %.3f  p50 %.3f  p99 %.3f  p99.99 %.3f  max %.3f" % (d.mean(), np.percentile(d, 50), np.percentile(d, 99), np.percentile(d, 99.99), d.max()),
          " longer than 1.01 layers: %d, than 2.01: %d" % ((d > 1.01).sum(), (d > 2.01).sum()))
ts = np.linspace(t.min(), t.max(), shape[3])
c = ctx.morph_eval_many(ts, download=False)
print("visible per time: mean %.0f triangles, %.0f points; all triangles / times = %.0f" % (c[:, 1].mean(), c[:, 0].mean(), len(tris) / len(ts)))
