#!/usr/bin/env python3
"""Randomised Level-1 parity on the GPU: weld / tiny collapse / clean / orient of random small fields (closed interior and open,
rough to smooth, values rounded so that samples equal the isovalue) against oracle/postpass.py's canonical pipeline: counts after
every stage, triangles as weld-bucket triples, windings.  python tools/fuzz_gpu_level1.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from contourist_amd import _ffi
from oracle import level0, postpass
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
ctx = _ffi.Context(0)
t0 = time.time(); ncase = 0; nbad = 0; ntri = 0; nexc = 0
last_note = t0
while time.time() - t0 < budget:
    if time.time() - last_note > 60.0:      # (a GPU box takes a run that says nothing for minutes to be hung)
        last_note = time.time()
        print("... %.0f s" % (time.time() - t0), flush=True)
    shape = tuple(int(x) for x in rng.randint(6, 26, size=3))
    if rng.rand() < 0.3:
        shape = (shape[0], shape[1], int(rng.randint(4, 70)))
    B = rng.standard_normal(shape)
    for _ in range(int(rng.randint(0, 5))):
        for ax in range(3):
            B = 0.25 * np.roll(B, 1, ax) + 0.5 * B + 0.25 * np.roll(B, -1, ax)
    B = (B / max(B.std(), 1e-9)).astype(np.float32)
    rounded = rng.rand() < 0.25
    if rounded:
        B = (np.round(B * 4) / 4).astype(np.float32)
    closed = rng.rand() < 0.6
    if closed:
        fill = np.float32(B.min() - 1.0)
        for ax in range(3):
            sl = [slice(None)] * 3
            for idx in (0, 1, -1, -2):
                sl[ax] = idx
                B[tuple(sl)] = fill
    v = float(np.float32(rng.choice([0.0, 0.25, float(np.round(rng.uniform(-0.9, 0.9), 3))])))
    ctx.upload_grid(B)
    ctx.extract3d(v, _ffi.CX_DIAG_CPYTHON310 | int(rng.choice([0, _ffi.CX_KERNEL_TILED])))
    post = ctx.postprocess3d(0)
    pts, tris = ctx.download_level1(post)
    O = level0.march3d(B, v, diag_mode=1)
    ncase += 1
    if len(O["tris"]) == 0:
        if post["n_triangles"] != 0:
            nbad += 1; print("MISMATCH (empty) shape", shape, "v", v, post, flush=True)
        continue
    corner = np.array(shape) - 1
    ko = level0.edge_keys_from_pairs(O["pairs"], shape)
    L1 = postpass.level1_from_level0(ko, O["xyz"], O["tris"], corner)
    ok = post["n_after_weld"] == L1["n_after_weld"] and post["n_after_tiny"] == L1["n_after_tiny"] and len(tris) == len(L1["triangles"])
    cmp = None
    if ok:
        cmp = postpass.compare_level1(L1, pts, tris, corner, reach=0)
        ok = not cmp["missing"] and not cmp["extra"] and not cmp["winding"]
        nexc += int(cmp["excused_rows"])
    ntri += len(tris)
    if not ok:
        nbad += 1
        print("MISMATCH shape", shape, "v", v, "closed", closed, "rounded", rounded, post, "oracle", L1["n_after_weld"], L1["n_after_tiny"], len(L1["triangles"]),
              None if cmp is None else {k: (len(cmp[k]) if hasattr(cmp[k], "__len__") else cmp[k]) for k in ("missing", "extra", "winding", "excused_rows")}, flush=True)
ctx.close()
print("fuzz Level 1: %d cases, %d triangles, %d mismatches, %d excused rows, %.0f s" % (ncase, ntri, nbad, nexc, time.time() - t0))
sys.exit(1 if nbad else 0)
