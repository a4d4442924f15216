#!/usr/bin/env python3
"""config 5 of BASELINE.json: the config-3 grid, 8 isovalues at the 20/30/.../90th percentiles.  Two ways on the resident
grid: all levels in ONE call (cx_extract3d_levels: one pass over the samples for all levels) and the levels one by one
(cx_extract3d_async per level).  Reports Mvoxels/s with the grid counted once and Mvoxel-levels/s."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contourist_amd import _ffi, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
A = synthetic.smooth_noise_torch((size,) * 3, 1235, 1400, dev)
sample = A.flatten()[:: max(1, A.numel() // (1 << 22))].float()
levels = [float(torch.quantile(sample, q / 100.0)) for q in range(20, 100, 10)]
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
counts = ctx.extract3d_levels(levels, 1)        # sizes every level's buffers
best_one = 1e9
for rnd in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(3):
        ctx.extract3d_levels(levels, 1)
    torch.cuda.synchronize()
    best_one = min(best_one, (time.perf_counter() - t0) / 3)
ctx2 = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx2.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
ctx2.reserve(int(max(c["n_cells"] for c in counts) * 1.05) + 1024, int(max(c["n_vertices"] for c in counts) * 1.05) + 1024,
             int(max(c["n_triangles"] for c in counts) * 1.05) + 1024)
for v in levels:
    ctx2.extract3d_async(v, 1)
best = 1e9
for rnd in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(3):
        for v in levels:
            ctx2.extract3d_async(v, 1)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 3)
out = {"workload": "%d^3 fp32 smooth noise, 8 isovalues at the 20..90th percentiles, Level 0 of every level on the resident grid" % size,
       "levels": levels, "triangles_per_level": [c["n_triangles"] for c in counts],
       "active_voxel_fraction_per_level": [c["n_border_voxels"] / float((size - 1) ** 3) for c in counts],
       "one_call": {"ms_all_levels": best_one * 1e3, "Mvoxels_per_s_grid_once": size ** 3 / best_one / 1e6,
                    "Mvoxel_levels_per_s": 8 * size ** 3 / best_one / 1e6,
                    "note": "cx_extract3d_levels: one stream-kernel launch for all levels, synchronous (counts come back between the scan and the emit stages)"},
       "level_by_level": {"ms_all_levels": best * 1e3, "Mvoxels_per_s_grid_once": size ** 3 / best / 1e6,
                          "Mvoxel_levels_per_s": 8 * size ** 3 / best / 1e6}}
print(json.dumps(out))
