"""GPU box (host side numpy): how local are the edges of the Level-1 mesh of the 512^3 bench field in triangle-id order?
fraction of interior edges whose two triangles lie in the same block of B consecutive triangles"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from contourist_amd import _ffi, synthetic   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = synthetic.smooth_noise_torch((n,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
ctx.extract3d(0.0, 1)
post = ctx.postprocess3d(0)
pts, tris = ctx.download_level1(post)
t = tris.astype(np.int64)
nt = len(t)
e = np.concatenate([np.stack([t[:, a], t[:, (a + 1) % 3]], axis=1) for a in range(3)])
key = np.minimum(e[:, 0], e[:, 1]) << 32 | np.maximum(e[:, 0], e[:, 1])
tid = np.tile(np.arange(nt, dtype=np.int64), 3)
order = np.argsort(key, kind="stable")
k, ti = key[order], tid[order]
same = k[1:] == k[:-1]
first = np.ones(len(k), dtype=bool)
first[1:] = ~same
cnt = np.diff(np.append(np.flatnonzero(first), len(k)))
print("triangles", nt, "distinct edges", int(first.sum()), "with 1 / 2 / 3+ triangles:", int((cnt == 1).sum()), int((cnt == 2).sum()), int((cnt > 2).sum()))
pa, pb = ti[:-1][same], ti[1:][same]
d = np.abs(pa - pb)
for B in (256, 1024, 2048, 8192, 65536):
    print("block", B, "pairs in one block: %.3f" % float(np.mean(pa // B == pb // B)))
print("id distance percentiles 50/75/90/99:", np.percentile(d, [50, 75, 90, 99]).astype(int).tolist())
