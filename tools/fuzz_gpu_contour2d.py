#!/usr/bin/env python3
"""Randomised 2-D contour parity on the GPU (cx_contour2d_extract against oracle/contour2d.py): random shapes (thin, ragged), rough to
smooth fields, values rounded so that samples equal the levels (zeros of either sign), several levels per call -- every polyline as a
sequence of lattice pairs (modulo rotation and direction) and every point to 1e-12.  python tools/fuzz_gpu_contour2d.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from contourist_amd import _ffi
from oracle import contour2d as o2
from test_gpu_contour2d import device_chains
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = _ffi.Context(0)
t0 = time.time(); ncase = 0; nbad = 0; npts = 0
last_note = t0
while time.time() - t0 < budget:
    if time.time() - last_note > 60.0:      # (a GPU box takes a run that says nothing for minutes to be hung)
        last_note = time.time()
        print("... %.0f s" % (time.time() - t0), flush=True)
    shape = (int(rng.randint(2, 60)), int(rng.randint(2, 90)))
    B = rng.standard_normal(shape)
    for _ in range(int(rng.randint(0, 4))):
        for ax in range(2):
            B = 0.25 * np.roll(B, 1, ax) + 0.5 * B + 0.25 * np.roll(B, -1, ax)
    A = (B / max(B.std(), 1e-9)).astype(np.float32)
    if rng.rand() < 0.3:
        A = (np.round(A * 4) / 4).astype(np.float32)
    values = sorted(set(float(np.float32(x)) for x in list(rng.choice([0.0, 0.25, -0.5, 1.0], size=int(rng.randint(1, 3)))) + list(np.round(rng.uniform(-1.2, 1.2, size=int(rng.randint(0, 4))), 3))))
    got, npairs = device_chains(A, values, None, _ffi.CX2_ALL_CHAINS | _ffi.CX2_NO_DEDUPE, ctx=ctx)
    ok = True; total = 0
    for k, v in enumerate(values):
        want = o2.contours(A, v, "all", "build", dedupe=False)
        if o2.canonical_keys(got[k]) != o2.canonical_keys(want):
            ok = False; break
        where = {}
        for _, p, keys in want:
            for row, q in zip(keys, p):
                where[tuple(int(x) for x in row)] = q
        for _, p, keys in got[k]:
            for row, q in zip(keys, p):
                if np.max(np.abs(where[tuple(int(x) for x in row)] - q)) > 1e-12:
                    ok = False
            total += len(p)
    ok = ok and total == npairs
    ncase += 1; npts += total
    if not ok:
        nbad += 1
        print("MISMATCH shape", shape, "levels", values, flush=True)
ctx.close()
print("fuzz 2-D contours: %d cases, %d points, %d mismatches, %.0f s" % (ncase, npts, nbad, time.time() - t0))
sys.exit(1 if nbad else 0)
