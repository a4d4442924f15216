#!/usr/bin/env python3
"""Timing ablations of the Level-0 kernels (interleaved rounds in one process, guide rule 24)."""
import argparse
import os
import sys
os.environ.setdefault("CX_DEBUG", "1")   # ablation flags and tuning knobs are refused otherwise

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from contourist_amd import _ffi, synthetic  # noqa: E402

VARIANTS = [
    ("full", 0),
    ("full_canonical_diag", -1),
    ("generic", _ffi.CX_KERNEL_GENERIC),
    ("phaseA_only", 0x10000 | 0x800000),
    ("count_only", 0x20000 | 0x800000),
    ("no_celltab", 0x40000),
    ("no_verts", 0x80000),
    ("no_cells_no_emit", 0x100000 | 0x800000),
    ("no_vloads", 0x1000000),
    ("no_vloads_no_verts", 0x1000000 | 0x80000),
    ("no_vloads_no_stores", 0x1000000 | 0x40000 | 0x80000 | 0x100000 | 0x800000),
    ("no_stores_K1", 0x40000 | 0x80000 | 0x100000 | 0x800000),
    ("emit_no_lookup", 0x200000),
    ("emit_no_tris", 0x400000),
    ("emit_no_lookup_no_tris", 0x600000),
]

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--passes", type=int, default=1400)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
A = synthetic.smooth_noise_torch((args.size,) * 3, 1235, args.passes, dev)
ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
c = ctx.extract3d(0.0, 1)
print("counts", c)
res = {name: [] for name, _ in VARIANTS}
for rnd in range(args.rounds):
    for name, fl in VARIANTS:
        print("round", rnd, name, flush=True)
        fl = 0 if fl == -1 else (1 | fl)
        ctx.extract3d_async(0.0, fl)       # warm
        ctx.timing_enable(True)
        for _ in range(3):
            ctx.extract3d_async(0.0, fl)
        t = ctx.timing_read()
        ctx.timing_enable(False)
        res[name].append(tuple(t[k] / t["n"] for k in ("classify_ms", "emit_ms", "stream_ms", "scan_ms", "cells_ms")))
for name, _ in VARIANTS:
    med = [sorted(x[c] for x in res[name])[len(res[name]) // 2] for c in range(5)]
    print("%-24s K1 %.3f (stream %.3f scan %.3f verts %.3f) | K2 %.3f ms" % (name, med[0], med[2], med[3], med[4], med[1]))
