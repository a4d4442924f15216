#!/usr/bin/env python3
"""oracle/make_goldens_segments.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

FunctionGrid.find_contour_crossing_grid_segments (grid_field.py:64-84) of the REAL reference on a small callable field at
skip = 1, 2, 3: the crossing lattice segments IN THE ORDER the reference lists them (the seeded growth that
search_for_endpoints(skip > 1) feeds with this list is order-sensitive) -> tests/golden/crossing_segments.npz."""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def field(x, y, z):
    return np.sin(2.1 * x) + np.cos(1.7 * y) * 0.8 + 0.6 * np.sin(1.3 * z + 0.4)


MINS, MAXES, DELTA, VALUE = [-1, -1, -1], [1, 1.2, 0.9], [0.2, 0.25, 0.3], 0.3

if __name__ == "__main__":
    np.int = int
    np.float = float
    np.sometrue = np.any
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from contourist import grid_field
    out = {}
    for skip in (1, 2, 3):
        g = grid_field.FunctionGrid(MINS, MAXES, DELTA, field)
        maxf, minf, segs = g.find_contour_crossing_grid_segments(VALUE, skip)
        out["skip%d" % skip] = np.array([list(p) + list(q) for p, q in segs], dtype=np.int32).reshape(-1, 6)
        out["range%d" % skip] = np.array([maxf, minf], dtype=np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "crossing_segments.npz"), **out)
    print({k: v.shape for k, v in out.items()})
