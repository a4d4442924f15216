"""GPU: a few cx_extract3d_levels calls (8 levels of the 512^3 bench grid) for profilers -- no subprocesses, nothing else"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch   # noqa: E402
from contourist_amd import _ffi, synthetic   # noqa: E402

dev = torch.device("cuda", 0)
A = synthetic.smooth_noise_torch((512,) * 3, 1235, 1400, dev)
sample = A.flatten()[:: max(1, A.numel() // (1 << 22))].float()
levels = [float(torch.quantile(sample, q / 100.0)) for q in range(20, 100, 10)]
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
for _ in range(4):
    ctx.extract3d_levels(levels, 1)
torch.cuda.synchronize()
print("done")
