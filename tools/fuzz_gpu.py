#!/usr/bin/env python3
"""Randomised Level-0 parity on the GPU: random shapes (ragged rows, several tiles in every direction), fields from white noise to
smooth, random isovalues (some equal to samples), both diagonal modes, the staged and the tile path -- every mesh against
oracle/march_oracle.c (crossing edges, triangle key triples exactly; coordinates 1e-6).  python tools/fuzz_gpu.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from contourist_amd import _ffi
from oracle import level0
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
ctx = _ffi.Context(0)
t0 = time.time(); ncase = 0; nbad = 0; ntri = 0; paths = {}
last_note = t0
while time.time() - t0 < budget:
    if time.time() - last_note > 60.0:      # (a GPU box takes a run that says nothing for minutes to be hung)
        last_note = time.time()
        print("... %.0f s" % (time.time() - t0), flush=True)
    kind = rng.randint(0, 4)
    if kind == 0:
        shape = tuple(int(x) for x in rng.randint(2, 24, size=3))
    elif kind == 1:
        shape = (int(rng.randint(2, 40)), int(rng.randint(2, 40)), int(rng.randint(4, 600)))
    elif kind == 2:
        shape = (int(rng.randint(8, 90)), int(rng.randint(8, 70)), int(rng.randint(8, 300)))
    else:
        shape = (int(rng.randint(30, 160)), int(rng.randint(16, 48)), int(rng.choice([64, 128, 255, 256, 257, 260, 512])))
    A = rng.standard_normal(shape)
    for _ in range(int(rng.randint(0, 6))):
        for ax in range(3):
            A = 0.25 * np.roll(A, 1, ax) + 0.5 * A + 0.25 * np.roll(A, -1, ax)
    A = (A / max(A.std(), 1e-9)).astype(np.float32)
    if rng.rand() < 0.3:
        A = (np.round(A * 4) / 4).astype(np.float32)          # many samples equal to candidate isovalues
    v = float(np.float32(rng.choice([0.0, 0.25, -0.5, float(rng.uniform(-1.2, 1.2))])))
    diag = int(rng.randint(0, 2))
    extra = int(rng.choice([0, _ffi.CX_KERNEL_TILED, _ffi.CX_KERNEL_STAGED, _ffi.CX_KERNEL_FUSED]))
    if shape[2] < 4:
        extra = 0
    origin = tuple(int(x) for x in rng.randint(0, 700, size=3)) if rng.rand() < 0.3 else ((-1, -1, -1) if rng.rand() < 0.1 else (0, 0, 0))
    ctx.set_origin(*origin)
    ctx.upload_grid(A)
    if shape[2] >= 4 and rng.rand() < 0.15:
        # several isovalues in one call (cx_extract3d_levels): every level against the oracle
        vals = sorted(set([v] + [float(np.float32(x)) for x in rng.uniform(-1.0, 1.0, size=int(rng.randint(1, 4)))]))
        cs = ctx.extract3d_levels(vals, diag)
        for li, lv in enumerate(vals):
            ctx.select_level(li)
            xyz, keys, tris = ctx.download_level0(cs[li])
            O = level0.march3d(A, lv, diag_mode=diag, origin=origin)
            ko = level0.edge_keys_from_pairs(O["pairs"], shape)
            co = level0.canonical_level0(ko, O["xyz"], O["tris"])
            ok = len(keys) == len(ko) and len(tris) == len(O["tris"]) and (len(tris) == 0 or (tris.min() >= 0 and tris.max() < len(keys)))
            if ok:
                ch = level0.canonical_level0(keys.astype(np.int64), xyz, tris.astype(np.int64))
                ok = np.array_equal(co[0], ch[0]) and np.array_equal(co[2], ch[2])
            ncase += 1; ntri += len(tris); paths["levels"] = paths.get("levels", 0) + 1
            if not ok:
                nbad += 1
                print("MISMATCH (levels) shape", shape, "values", vals, "level", li, "diag", diag, "origin", origin, cs[li], len(ko), len(O["tris"]), flush=True)
        continue
    c = ctx.extract3d(v, diag | extra)
    p = ctx.level0_path(); paths[p] = paths.get(p, 0) + 1
    xyz, keys, tris = ctx.download_level0(c)
    O = level0.march3d(A, v, diag_mode=diag, origin=origin)
    ko = level0.edge_keys_from_pairs(O["pairs"], shape)
    co = level0.canonical_level0(ko, O["xyz"], O["tris"])
    if len(tris) and (tris.min() < 0 or tris.max() >= len(keys)):
        nbad += 1; ncase += 1
        print("INDEX OUT OF RANGE shape", shape, "v", v, "diag", diag, "extra", hex(extra), "path", p, c, "max index", int(tris.max()), "oracle", len(ko), len(O["tris"]), flush=True)
        np.save("gpurun_out/t9/fuzz_bad_%d.npy" % ncase, A)
        continue
    ch = level0.canonical_level0(keys.astype(np.int64), xyz, tris.astype(np.int64))
    ok = (np.array_equal(co[0], ch[0]) and np.array_equal(co[2], ch[2]) and
          (len(co[1]) == 0 or np.all(np.abs(ch[1] - co[1]) <= 1e-6 * np.abs(co[1]) + 1e-6)) and c["n_border_voxels"] == O["nborder_mixed"])
    ncase += 1; ntri += len(tris)
    if not ok:
        nbad += 1
        print("MISMATCH shape", shape, "v", v, "diag", diag, "extra", hex(extra), "origin", origin, "path", p, c, len(ko), len(O["tris"]), flush=True)
ctx.close()
print("fuzz: %d cases, %d triangles, %d mismatches, kernels by path %s, %.0f s" % (ncase, ntri, nbad, dict(sorted(paths.items(), key=str)), time.time() - t0))
sys.exit(1 if nbad else 0)
