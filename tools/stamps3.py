#!/usr/bin/env python3
"""where a vertex-stage wave spends its time (needs a -DCX_S3_STAMPS build; diagnostic only)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from contourist_amd import _ffi, synthetic
A = synthetic.smooth_noise_torch((512,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
print(ctx.extract3d(0.0, 1))
nw = 16384 * 8
names = ["entries+drain", "decode/prefix/slots", "issue loads", "wait loads", "interp+stores issue"]
for label, fl in (("full", 0), ("no celltab", 0x40000), ("no cell records", 0x100000), ("no vertex stores", 0x80000),
                  ("no stores at all", 0x40000 | 0x100000 | 0x80000), ("no sample loads", 0x1000000)):
    ctx._check(ctx.lib.cx_debug_stamps(ctx.handle, nw, None))
    ctx.extract3d_async(0.0, 1 | 0x800000 | fl); ctx.synchronize()
    buf = np.zeros(nw, dtype=np.uint64)
    ctx._check(ctx.lib.cx_debug_stamps(ctx.handle, nw, buf.ctypes.data))
    S = buf.reshape(-1, 8).astype(np.float64)
    S = S[S[:, 7] > 0]
    rounds = S[:, 6].sum()
    print("%-18s per round: " % label + "  ".join("%s %5.0f" % (nm.split("/")[0][:12], S[:, k].sum() / rounds) for k, nm in enumerate(names)) +
          "  | batch/round %6.0f" % (S[:, 5].sum() / rounds))
