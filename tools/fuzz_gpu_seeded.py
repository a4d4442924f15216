#!/usr/bin/env python3
"""Randomised parity of the seeded selection (cx_select_seeded3d_ex: the two-step unions, the per-triangle keep kernel) on the GPU:
random multi-component fields (several thousand surface voxels: many blocks of records, pairs across blocks), random seed segments
taken from crossing edges, random in_range boxes -- the kept triangle mask against oracle/seeds.py.
python tools/fuzz_gpu_seeded.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from contourist_amd import _ffi
from oracle import level0, seeds
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
t0 = time.time(); last_note = t0; ncase = 0; nbad = 0; ntri = 0
ctx = _ffi.Context(0)
while time.time() - t0 < budget:
    if time.time() - last_note > 60.0:
        last_note = time.time()
        print("... %d cases, %d mismatches, %.0f s" % (ncase, nbad, time.time() - t0), flush=True)
    shape = tuple(int(x) for x in rng.randint(18, 34, size=3))
    A = rng.standard_normal(shape)
    for _ in range(int(rng.randint(2, 5))):
        for ax in range(3):
            A = 0.25 * np.roll(A, 1, ax) + 0.5 * A + 0.25 * np.roll(A, -1, ax)
    A = (A / max(A.std(), 1e-9)).astype(np.float32)
    A += np.float32(1e-4) * rng.standard_normal(shape).astype(np.float32)      # (no sample equal to the isovalue)
    v = float(np.float32(rng.uniform(-0.7, 0.7)))
    ctx.upload_grid(A)
    counts = ctx.extract3d(v, 1)
    if counts["n_triangles"] < 50:
        continue
    xyz, keys, tris = ctx.download_level0(counts)
    keys = keys.astype(np.int64)
    lin, d = keys >> 3, keys & 7
    q = np.stack([lin // (shape[1] * shape[2]), (lin // shape[2]) % shape[1], lin % shape[2]], axis=1)
    dv = np.stack([(d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1)
    O = level0.march3d(A, v, diag_mode=1)
    ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
    order_o, order_d = np.argsort(ko), np.argsort(keys)
    for trial in range(3):
        pick = rng.choice(len(keys), size=int(rng.randint(1, 4)), replace=False)
        eps = [[tuple(int(x) for x in q[p]), tuple(int(x) for x in q[p] + dv[p])] for p in pick]
        want, _ = seeds.select(A, v, eps, ko, O["tris"])
        got = ctx.select_seeded(eps)
        tk, vk = ctx.seeded_masks(counts)
        # compare as sets of key triples
        kept_d = set(tuple(sorted(int(keys[i]) for i in t)) for t in np.asarray(tris)[tk])
        kept_o = set(tuple(sorted(int(ko[i]) for i in t)) for t in np.asarray(O["tris"])[np.asarray(want, dtype=bool)])
        ncase += 1; ntri += len(kept_o)
        if got["triangles_kept"] != int(np.sum(want)) or kept_d != kept_o:
            nbad += 1
            print("MISMATCH shape", shape, "v", v, "eps", eps, got, int(np.sum(want)), flush=True)
print("fuzz seeded selection: %d cases, %d triangles kept, %d mismatches, %.0f s" % (ncase, ntri, nbad, time.time() - t0))
sys.exit(1 if nbad else 0)
