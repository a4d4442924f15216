"""GPU: cx_extract3d_levels (8 levels of the 512^3 bench grid) with 1..4 streams for the emit stages (CX_DEBUG=1 CX_LEVELS_STREAMS=n)"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
code = r'''
import sys, time, os
sys.path.insert(0, %r)
import torch
from contourist_amd import _ffi, synthetic
dev = torch.device("cuda", 0)
A = synthetic.smooth_noise_torch((512,) * 3, 1235, 1400, dev)
sample = A.flatten()[:: max(1, A.numel() // (1 << 22))].float()
levels = [float(torch.quantile(sample, q / 100.0)) for q in range(20, 100, 10)]
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
ctx.extract3d_levels(levels, 1)
best = 1e9
for rnd in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(3):
        ctx.extract3d_levels(levels, 1)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 3)
print("streams", os.environ.get("CX_LEVELS_STREAMS", "default"), "ms for 8 levels: %%.3f" %% (best * 1e3))
''' % ROOT
for n in ("0", "2", "3", "4"):
    env = dict(os.environ, CX_DEBUG="1", CX_LEVELS_STREAMS=n)
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
