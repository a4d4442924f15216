#!/usr/bin/env python3
"""Randomised 4-D Level-0 parity on the GPU: random shapes (rows that are / are not whole bitmap words, minimum sizes), fields from white
noise to smooth, values rounded so that samples equal the isovalue (zeros of either sign included), random origins for the CPython-order
2-3 splits, both split modes -- every mesh against oracle/march4d_oracle.c (crossing edges and tetrahedra as key quadruples exactly,
coordinates 1e-6, border voxels).  python tools/fuzz_gpu4d.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from contourist_amd import _ffi
from oracle import level0_4d
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
ctx = _ffi.Context(0)
t0 = time.time(); ncase = 0; nbad = 0; ntet = 0
last_note = t0
while time.time() - t0 < budget:
    if time.time() - last_note > 60.0:      # (a GPU box takes a run that says nothing for minutes to be hung)
        last_note = time.time()
        print("... %.0f s" % (time.time() - t0), flush=True)
    kind = rng.randint(0, 3)
    if kind == 0:
        shape = tuple(int(x) for x in rng.randint(2, 9, size=4))
    elif kind == 1:
        shape = (int(rng.randint(2, 12)), int(rng.randint(2, 12)), int(rng.randint(2, 12)), int(rng.choice([2, 3, 31, 32, 33, 40, 64, 65])))
    else:
        shape = (int(rng.randint(6, 20)), int(rng.randint(6, 20)), int(rng.randint(6, 20)), int(rng.randint(4, 24)))
    A = rng.standard_normal(shape)
    for _ in range(int(rng.randint(0, 5))):
        for ax in range(4):
            A = 0.25 * np.roll(A, 1, ax) + 0.5 * A + 0.25 * np.roll(A, -1, ax)
    A = (A / max(A.std(), 1e-9)).astype(np.float32)
    if rng.rand() < 0.35:
        A = (np.round(A * 4) / 4).astype(np.float32)
    v = float(np.float32(rng.choice([0.0, 0.25, -0.5, float(rng.uniform(-1.0, 1.0))])))
    diag = int(rng.randint(0, 2))
    origin = tuple(int(x) for x in rng.randint(0, 50, size=4)) if rng.rand() < 0.3 else (0, 0, 0, 0)
    ctx.set_origin4d(*origin)
    ctx.upload_grid4d(A)
    c = ctx.extract4d(v, diag)
    verts, keys, tets = ctx.download_level0_4d(c)
    O = level0_4d.march4d(A, v, diag_mode=diag, origin=origin)
    ko = level0_4d.edge_keys4(O["pairs"], shape)
    ok = c["n_vertices"] == len(ko) and c["n_tetrahedra"] == len(O["tets"]) and c["n_border_voxels"] == O["nborder_mixed"]
    if ok and len(ko):
        a = level0_4d.canonical4(keys.astype(np.int64), verts, tets.astype(np.int64))
        b = level0_4d.canonical4(ko, O["xyzt"], O["tets"])
        ok = np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.all(np.abs(a[1] - b[1]) <= 1e-6 * np.abs(b[1]) + 1e-6)
    ncase += 1; ntet += len(tets)
    if not ok:
        nbad += 1
        print("MISMATCH shape", shape, "v", v, "mode", diag, "origin", origin, c, "oracle", len(ko), len(O["tets"]), O["nborder_mixed"], flush=True)
ctx.close()
print("fuzz 4-D: %d cases, %d tetrahedra, %d mismatches, %.0f s" % (ncase, ntet, nbad, time.time() - t0))
sys.exit(1 if nbad else 0)
