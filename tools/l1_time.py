"""GPU: Level-1 post-pass time of the 512^3 bench mesh on a warm context (for tools/ab_build.sh: AB_TOOL=tools/l1_time.py)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch   # noqa: E402
from contourist_amd import _ffi, synthetic   # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, torch.device("cuda", 0))
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
ctx.extract3d(0.0, 1)
ts = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    post = ctx.postprocess3d(0)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("TAG", ""), "Level 1 ms:", " ".join("%.2f" % t for t in ts), "| triangles", post["n_triangles"], "components", post["n_components"])
