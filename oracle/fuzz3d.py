#!/usr/bin/env python3
"""oracle/fuzz3d.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

Random small closed-interior fields: the Level-0 snapshot of the REAL reference (make_goldens.run_reference) against
the C restatement (oracle/march_oracle.c through level0.march3d, CPython-order diagonals) -- crossing edges, float64
points and triangles must be identical -- and the triangle counts after the weld against oracle/postpass.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg            # noqa: E402
from oracle import level0, postpass  # noqa: E402


def main(nfields=30):
    bad = []
    for seed in range(nfields):
        rng = np.random.RandomState(500 + seed)
        shape = tuple(int(x) for x in rng.randint(9, 15, size=3))
        B = rng.standard_normal(shape)
        for _ in range(int(rng.randint(1, 4))):
            for ax in range(3):
                B = 0.25 * np.roll(B, 1, ax) + 0.5 * B + 0.25 * np.roll(B, -1, ax)
        B = B / B.std()
        A = mg.close_interior(B.astype(np.float32), float(B.min()) - 1.0)
        v = float(np.round(rng.uniform(-0.8, 0.8), 3))
        R = mg.run_reference(A, v)
        O = level0.march3d(A, v, diag_mode=1)
        kr = level0.edge_keys_from_pairs(R["l0_pairs"], A.shape)
        ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
        a = level0.canonical_level0(kr, R["l0_xyz"], R["l0_tris"])
        b = level0.canonical_level0(ko, O["xyz"], O["tris"])
        ok = np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1])
        L1 = postpass.level1_from_level0(ko, O["xyz"], O["tris"], np.array(A.shape) - 1)
        ok = ok and L1["n_after_weld"] == int(R["n_tris_after_weld"])
        if not ok:
            bad.append((seed, shape, v))
        print("seed %d shape %s v=%g: %d vertices %d triangles %s" % (seed, shape, v, len(kr), len(R["l0_tris"]), "ok" if ok else "DIFFERENT"), flush=True)
    print("fields %d, identical %d, different %s" % (nfields, nfields - len(bad), bad))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
