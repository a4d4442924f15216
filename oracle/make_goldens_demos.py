#!/usr/bin/env python3
"""oracle/make_goldens_demos.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

The reference's own demo calls of the 3-D path (contourist/html_demo.py:163-282: test_centered, test_json2,
test_sphere, test_hyperbola, test_torus, test_wave) run on the REAL reference; what get_points_and_triangles()
returned goes to tests/golden_demos/*.npz (points, triangles, the call's parameters).  The fields are re-stated in
tests/test_gpu_demos.py (they are one-line formulas), nothing of the reference's source is stored."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..")))
sys.path.insert(0, HERE)
import make_goldens as mg   # noqa: E402

OUT = os.path.normpath(os.path.join(HERE, "..", "tests", "golden_demos"))


def demos(tetrahedral):
    from numpy.linalg import norm
    import math
    D = {}
    D["centered"] = lambda: tetrahedral.TriangulatedIsosurfaces((-1, -1, -1), (1, 1, 1), (0.25, 0.2, 0.33), lambda x, y, z: norm([x, y, z]), 1.3, [])
    D["sphere"] = lambda: tetrahedral.Grid3DContour(10, 10, 10, lambda x, y, z: norm([x - 5, y - 5, z - 5]), 6.0, [[(0, 0, 0), (5, 5, 5)]])
    D["hyperbola"] = lambda: tetrahedral.Grid3DContour(50, 50, 50, lambda x, y, z: x * y * z, 100.0, [[(0, 0, 0), (20, 20, 20)]])
    c = np.array((5.0, 0.0))

    def shift_torus(x, y, z):
        return norm(c - np.array((norm((x - 15, y - 15)), z - 15)))
    D["torus"] = lambda: tetrahedral.Grid3DContour(30, 30, 30, shift_torus, 5 / 3.0, [[(0, 0, 0), (20, 15, 15)]])
    D["wave"] = lambda: tetrahedral.Grid3DContour(40, 40, 40, lambda x, y, z: 1.1 + math.sin(((x - 20) ** 2 + (y - 20) ** 2) * 0.02) - z, 0,
                                                 [[(20, 20, 0), (20, 20, 20)]])
    # the field of the reference's own unit test (test/test_tetrahedral.py:13-37) in lattice coordinates through
    # Grid3DContour: the dot at the corner of the grid is reached from a start voxel OUTSIDE the grid
    def dots(x, y, z):
        return 1 if (x == y == z == 0 or x == y == z == 4) else -1
    D["grid_two_dots"] = lambda: tetrahedral.Grid3DContour(8, 8, 8, dots, 0, [[(0, 0, 0), (0, 0, 8)]])
    # linear_interpolate=False (the default of contour_doodle.implicit_surface, contour_doodle.py:13-21): the crossing
    # points are refined by regula falsi on the callable (tetrahedral.py:488-505)
    d = 3.0 / 32
    D["sphere_nonlinear"] = lambda: tetrahedral.TriangulatedIsosurfaces([-1.5] * 3, [1.5 - d] * 3, [d] * 3, lambda x, y, z: x * x + y * y + z * z,
                                                                        1.0, [], linear_interpolate=False)
    D["quartic_nonlinear"] = lambda: tetrahedral.TriangulatedIsosurfaces([-1.2] * 3, [1.2] * 3, [0.15] * 3,
                                                                         lambda x, y, z: x ** 4 + y ** 4 + z ** 4 - 0.6 * x * y, 0.5, [],
                                                                         linear_interpolate=False)
    return D


def main():
    grid_field, surface_geometry, tetrahedral, triangulated = mg.reference_modules()
    os.makedirs(OUT, exist_ok=True)
    for name, make in demos(tetrahedral).items():
        t0 = time.time()
        obj = make()
        if name == "centered" or name.endswith("_nonlinear"):
            obj.search_for_endpoints()      # (the stale 2-D assert of the shared ctor rules out passing its end points)
        pts, tris = obj.get_points_and_triangles()
        pts = np.array(pts, dtype=np.float64).reshape(-1, 3)
        tris = np.array(tris, dtype=np.int64).reshape(-1, 3)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), points=pts, triangles=tris)
        print("%-10s %d points %d triangles (%.1f s)" % (name, len(pts), len(tris), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
