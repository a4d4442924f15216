#!/usr/bin/env python3
"""oracle/make_goldens.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

Imports the REAL reference (AaronWatters/contourist, Python) from /root/reference, runs its
tetrahedral pipeline stage by stage on small dense fp32 fields and writes input + output vectors
to tests/golden/*.npz.  Nothing of the reference's source is copied: only numbers leave.

Stages follow GridContour3d.get_points_and_triangles (contourist/tetrahedral.py:528-552):
    find_initial_voxels / expand_voxels  ->  enumerate_voxel_triangles   == "Level 0" snapshot
    quantize_interpolations -> remove_tiny_simplices -> extract_surface_geometry
    (clean_triangles + orient_triangles)                                == "Level 1" output

Every fixture is generated NORDERS times with the reference's surface_voxels iterated in
different orders (native, sorted, reversed, shuffled); the fixture records whether its Level-1
canonical form is invariant (`l1_order_invariant`), see SURVEY.md section 7 hard part 3.

usage:  python oracle/make_goldens.py [name ...]
"""
import os
import random
import sys
import time

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.normpath(os.path.join(HERE, "..", "tests", "golden"))
sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..")))


def reference_modules():
    """SURVEY.md Appendix B recipe: numpy alias shim + import from /root/reference."""
    if not os.path.isdir(REFERENCE):
        raise RuntimeError("reference not present; goldens can only be regenerated where it is mounted")
    np.int = int
    np.float = float
    np.sometrue = np.any
    sys.dont_write_bytecode = True
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    from contourist import grid_field, surface_geometry, tetrahedral, triangulated  # noqa
    return grid_field, surface_geometry, tetrahedral, triangulated


def as_reference_function(A, outside):
    n0, n1, n2 = A.shape

    def f(x, y, z):
        i, j, k = int(x), int(y), int(z)
        if i < 0 or j < 0 or k < 0 or i >= n0 or j >= n1 or k >= n2:
            return float(outside)
        return float(A[i, j, k])     # python float => all reference arithmetic is float64
    return f


# ---- synthetic fields (fp32, closed interior) ------------------------------------------------

def close_interior(A, below):
    """force the outermost 2 samples to `below` so no crossing touches the array boundary"""
    A = A.copy()
    for ax in range(3):
        sl = [slice(None)] * 3
        for idx in (0, 1, -1, -2):
            sl[ax] = idx
            A[tuple(sl)] = below
    return A


def smooth_noise(n, seed, passes):
    rng = np.random.RandomState(seed)
    A = rng.standard_normal((n, n, n))
    for _ in range(passes):
        for ax in range(3):
            A = 0.25 * np.roll(A, 1, ax) + 0.5 * A + 0.25 * np.roll(A, -1, ax)
    A = A / A.std()
    return A.astype(np.float32)


def sphere_field(n, center, radius2_scale=1.0):
    g = np.arange(n, dtype=np.float64)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    return (((X - center[0]) ** 2 + (Y - center[1]) ** 2 + (Z - center[2]) ** 2) * radius2_scale).astype(np.float32)


def fields():
    F = {}
    # config 1 of BASELINE.json: f = x^2+y^2+z^2 on mins=-1.5, delta=3/32, grid_dimensions 32^3 (33^3 samples)
    d = 3.0 / 32
    g = -1.5 + d * np.arange(33)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    F["sphere32"] = dict(A=(X * X + Y * Y + Z * Z).astype(np.float32), value=1.0,
                         mins=[-1.5] * 3, delta=[d] * 3)
    # inverted sphere: inside is HIGH -> reference normals are anti-parallel to grad f
    F["inv_sphere20"] = dict(A=(-sphere_field(21, (10.2, 9.9, 10.4))).astype(np.float32), value=-30.5)
    # nested shells: |r^2 - 49| < w  -> two concentric components with opposite grad orientation
    r2 = sphere_field(25, (12.1, 12.3, 11.8)).astype(np.float64)
    F["shells24"] = dict(A=np.abs(r2 - 49.0).astype(np.float32), value=22.0)
    # two disjoint blobs
    a = sphere_field(28, (8.3, 8.1, 8.4)).astype(np.float64)
    b = sphere_field(28, (19.2, 18.7, 19.1)).astype(np.float64)
    F["blobs27"] = dict(A=np.minimum(a, b * 1.7).astype(np.float32), value=17.3)
    # closed-interior smooth noise at several isovalues
    n24 = smooth_noise(24, 7, 3)
    lo = float(n24.min()) - 1.0
    F["noise24_v0"] = dict(A=close_interior(n24, lo), value=0.0)
    F["noise24_v07"] = dict(A=close_interior(n24, lo), value=0.7)
    n32 = smooth_noise(32, 11, 4)
    lo = float(n32.min()) - 1.0
    F["noise32_v0"] = dict(A=close_interior(n32, lo), value=0.0)
    F["noise32_vm05"] = dict(A=close_interior(n32, lo), value=-0.5)
    # non-cubic shape, non-representable isovalue
    rng = np.random.RandomState(5)
    B = rng.standard_normal((14, 20, 26))
    for _ in range(2):
        for ax in range(3):
            B = 0.25 * np.roll(B, 1, ax) + 0.5 * B + 0.25 * np.roll(B, -1, ax)
    B = (B / B.std()).astype(np.float32)
    F["noise_14x20x26"] = dict(A=close_interior(B, float(B.min()) - 1.0), value=0.1)
    # tiny amplitude: exercises the 1e-8 absolute tolerances (np.allclose) of the reference
    F["tiny_amp16"] = dict(A=(close_interior(smooth_noise(16, 3, 2), -4.0) * np.float32(2e-9)), value=0.0)
    F["tiny_amp16b"] = dict(A=(close_interior(smooth_noise(16, 3, 2), -4.0) * np.float32(1.2e-8)), value=0.0)
    # relative tolerance regime: values ~100, differences ~1e-3 (1e-5 * 100)
    F["rel_tol16"] = dict(A=(np.float32(100.0) + close_interior(smooth_noise(16, 4, 2), -4.0) * np.float32(1.2e-3)), value=100.0)
    # smooth_interpolations (tetrahedral.py:329-351)
    F["sphere32_smooth05"] = dict(A=F["sphere32"]["A"], value=1.0, mins=[-1.5] * 3, delta=[d] * 3, smooth=0.5)
    F["noise24_v07_smooth03"] = dict(A=F["noise24_v07"]["A"], value=0.7, smooth=0.3)
    # samples exactly equal to the isovalue (f == v counts as HIGH; strict search test differs)
    Q = np.round(close_interior(smooth_noise(18, 21, 2), -3.0) * 2.0) / 2.0
    F["quantised18"] = dict(A=Q.astype(np.float32), value=0.5)
    return F


# ---- staged reference run ----------------------------------------------------------------------

def run_reference(A, value, order="native", seed=0, mins=None, delta=None, smooth=None):
    grid_field, surface_geometry, tetrahedral, triangulated = reference_modules()
    A = np.ascontiguousarray(A, dtype=np.float32)
    shape = A.shape
    outside = float(A.min()) - 1.0
    if mins is None:
        mins, delta = [0.0] * 3, [1.0] * 3
    mins = np.array(mins, dtype=float)
    delta = np.array(delta, dtype=float)
    fa = as_reference_function(A, outside)

    def f(x, y, z):   # world -> nearest grid index (exact for our fixtures)
        return fa(round((x - mins[0]) / delta[0]), round((y - mins[1]) / delta[1]), round((z - mins[2]) / delta[2]))
    maxes = mins + delta * (np.array(shape) - 2)      # grid_dimensions = shape-1  (grid_field.py:26-27)
    t0 = time.time()
    S = tetrahedral.TriangulatedIsosurfaces(list(mins), list(maxes), list(delta), f, float(value), [], smooth=smooth)
    assert tuple(S.grid.grid_dimensions) == tuple(n - 1 for n in shape), (S.grid.grid_dimensions, shape)
    S.search_for_endpoints()
    n_crossing_segments = len(S.grid_endpoints)
    t_search = time.time() - t0
    cm = S.contour_maker
    cm.find_initial_voxels()
    while cm.new_surface_voxels:
        cm.expand_voxels()
    voxels = list(cm.surface_voxels)
    if order == "sorted":
        voxels = sorted(voxels)
    elif order == "reversed":
        voxels = sorted(voxels, reverse=True)
    elif order == "shuffled":
        voxels = sorted(voxels)
        random.Random(seed).shuffle(voxels)
    for triple in voxels:
        cm.enumerate_voxel_triangles(triple)
    # ---- Level 0 snapshot
    pair_list = list(cm.interpolated_contour_pairs.keys())
    pair_index = {p: n for n, p in enumerate(pair_list)}
    l0_pairs = np.array([list(p[0]) + list(p[1]) for p in pair_list], dtype=np.int32).reshape(-1, 6)
    l0_xyz = np.array([cm.interpolated_contour_pairs[p] for p in pair_list], dtype=np.float64).reshape(-1, 3)
    l0_tris = np.array([[pair_index[p] for p in s] for s in cm.simplex_sets], dtype=np.int64).reshape(-1, 3)
    out = dict(A=A, value=np.float64(value), mins=mins, delta=delta,
               surface_voxels=np.array(sorted(cm.surface_voxels), dtype=np.int32).reshape(-1, 3),
               n_crossing_segments=np.int64(n_crossing_segments),
               l0_pairs=l0_pairs, l0_xyz=l0_xyz, l0_tris=l0_tris)
    # ---- post passes
    cm.quantize_interpolations()
    out["n_tris_after_weld"] = np.int64(len(cm.simplex_sets))
    if smooth:
        cm.smooth_interpolations(smooth)          # tetrahedral.py:547-550
        out["smooth"] = np.float64(smooth)
    cm.remove_tiny_simplices()
    out["n_tris_after_tiny"] = np.int64(len(cm.simplex_sets))
    geometry = cm.extract_surface_geometry(True)
    grid_points = np.array(geometry.vertices, dtype=np.float64).reshape(-1, 3)
    tris = np.array(geometry.oriented_triangles, dtype=np.int64).reshape(-1, 3)
    out["l1_grid_points"] = grid_points
    out["l1_points"] = np.array([S.grid.from_grid_coordinates(p) for p in grid_points], dtype=np.float64).reshape(-1, 3)
    out["l1_triangles"] = tris
    out["t_search_s"] = np.float64(t_search)
    out["t_total_s"] = np.float64(time.time() - t0)
    return out


def l1_canonical(grid_points, triangles, shape):
    """Level-1 canonical form: each triangle as a triple of weld-bucket ids
    (trunc(p*expander), tetrahedral.py:192-196), rotated to min-first (winding kept), sorted."""
    from oracle import postpass
    return postpass.canonical_level1(grid_points, triangles, np.array(shape) - 1)


def make(name, spec, outdir=GOLDEN_DIR):
    A, value = spec["A"], spec["value"]
    kw = dict(mins=spec.get("mins"), delta=spec.get("delta"), smooth=spec.get("smooth"))
    base = run_reference(A, value, "native", **kw)
    canon0 = l1_canonical(base["l1_grid_points"], base["l1_triangles"], A.shape)
    invariant = True
    counts = [len(base["l1_triangles"])]
    for order, seed in (("sorted", 0), ("reversed", 0), ("shuffled", 1), ("shuffled", 2)):
        other = run_reference(A, value, order, seed, **kw)
        counts.append(len(other["l1_triangles"]))
        c = l1_canonical(other["l1_grid_points"], other["l1_triangles"], A.shape)
        if c.shape != canon0.shape or not np.array_equal(c, canon0):
            invariant = False
    base["l1_order_invariant"] = np.bool_(invariant)
    base["l1_count_band"] = np.array([min(counts), max(counts)], dtype=np.int64)
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, name + ".npz")
    np.savez_compressed(path, **base)
    print("%-16s shape=%s v=%g  L0: %d verts %d tris | weld %d tiny %d | L1: %d pts %d tris  invariant=%s band=%s  (%.1fs)" % (
        name, A.shape, value, len(base["l0_pairs"]), len(base["l0_tris"]), base["n_tris_after_weld"],
        base["n_tris_after_tiny"], len(base["l1_points"]), len(base["l1_triangles"]), invariant,
        base["l1_count_band"].tolist(), base["t_total_s"]))
    return path


def make_two_dots(outdir=GOLDEN_DIR):
    """The reference's own hot-path test (contourist/test/test_tetrahedral.py:13-37), replayed with the
    endpoint handed to get_contour_maker directly because the shared ctor's 2-D assert is stale
    (SURVEY.md section 4).  Stores the field as a dense array over grid indices -1..9 plus the
    expected integer-truncated triangle set the test asserts."""
    grid_field, surface_geometry, tetrahedral, triangulated = reference_modules()

    def two_dots(x, y, z):
        if x == y == z == -8 or x == y == z == 0:
            return 1
        return -1
    S = tetrahedral.TriangulatedIsosurfaces([-8] * 3, [8] * 3, [2] * 3, two_dots, 0, [])
    ep = S.to_grid_endpoint((-8, -8, -8), (-8, -8, 8))
    S.contour_maker = S.get_contour_maker([ep])
    points, triangles = S.get_points_and_triangles()
    ipoints = [tuple(int(i) for i in pt) for pt in points]
    got = sorted(sorted(ipoints[i] for i in tri) for tri in triangles)
    # dense samples on grid indices -1..9 (world -10..10): the un-range-checked seeds reach index -1
    idx = np.arange(-1, 10)
    A = np.full((11, 11, 11), -1.0, dtype=np.float32)
    for n, i in enumerate(idx):
        w = -8 + 2 * i
        if w == -8 or w == 0:
            A[n, n, n] = 1.0
    np.savez_compressed(os.path.join(outdir, "two_dots.npz"), A=A, value=np.float64(0.0),
                        mins=np.array([-10.0] * 3), delta=np.array([2.0] * 3),
                        expected_int_triangles=np.array(got, dtype=np.int64),
                        l1_points=np.array(points, dtype=np.float64),
                        l1_triangles=np.array(triangles, dtype=np.int64))
    print("two_dots: %d triangles" % len(got))


COARSE = dict(corner=511, center=(201.25, 310.5, 255.75), radius2=144.0,
              end_points=[[[201, 310, 255], [201, 310, 275]]])


def coarse_field(i, j, k):
    "fp32 sample of the coarse-regime fixture at lattice point (i, j, k); works on scalars and on arrays"
    cx, cy, cz = COARSE["center"]
    return np.float32((i - cx) ** 2 + (j - cy) ** 2 + (k - cz) ** 2)


def make_coarse(outdir=GOLDEN_DIR):
    """Grid3DContour(511, 511, 511, f, v, end points) on a sphere of radius 12 (tetrahedral.py:104-107): the COARSE
    regime of the post-passes -- weld bucket 1/int(10000/511) = 1/19 voxel, tiny threshold 0.05 voxel.
    Only the surface neighbourhood is ever evaluated by the reference; the fixture stores the parameters of the field,
    the Level-0 snapshot, the triangle counts after every stage for 5 voxel orders and the native Level-1 mesh."""
    grid_field, surface_geometry, tetrahedral, triangulated = reference_modules()
    c = COARSE["corner"]

    def f(x, y, z):
        return float(coarse_field(float(x), float(y), float(z)))
    runs = []
    for order, seed in (("native", 0), ("sorted", 0), ("reversed", 0), ("shuffled", 1), ("shuffled", 2)):
        t0 = time.time()
        cm = tetrahedral.Grid3DContour(c, c, c, f, COARSE["radius2"], [[tuple(a), tuple(b)] for a, b in COARSE["end_points"]])
        cm.find_initial_voxels()
        while cm.new_surface_voxels:
            cm.expand_voxels()
        voxels = list(cm.surface_voxels)
        if order == "sorted":
            voxels = sorted(voxels)
        elif order == "reversed":
            voxels = sorted(voxels, reverse=True)
        elif order == "shuffled":
            voxels = sorted(voxels)
            random.Random(seed).shuffle(voxels)
        for triple in voxels:
            cm.enumerate_voxel_triangles(triple)
        out = {}
        if order == "native":
            pair_list = list(cm.interpolated_contour_pairs.keys())
            pair_index = {p: n for n, p in enumerate(pair_list)}
            out["l0_pairs"] = np.array([list(p[0]) + list(p[1]) for p in pair_list], dtype=np.int32).reshape(-1, 6)
            out["l0_xyz"] = np.array([cm.interpolated_contour_pairs[p] for p in pair_list], dtype=np.float64).reshape(-1, 3)
            out["l0_tris"] = np.array([[pair_index[p] for p in s] for s in cm.simplex_sets], dtype=np.int64).reshape(-1, 3)
        n0 = len(cm.simplex_sets)
        cm.quantize_interpolations()
        n1 = len(cm.simplex_sets)
        cm.remove_tiny_simplices()
        n2 = len(cm.simplex_sets)
        geometry = cm.extract_surface_geometry(True)
        out["l1_grid_points"] = np.array(geometry.vertices, dtype=np.float64).reshape(-1, 3)
        out["l1_triangles"] = np.array(geometry.oriented_triangles, dtype=np.int64).reshape(-1, 3)
        out["counts"] = (n0, n1, n2, len(out["l1_triangles"]))
        runs.append(out)
        print("coarse %-9s emitted %d  weld %d  tiny %d  final %d  (%.1fs)" % ((order,) + out["counts"] + (time.time() - t0,)))
    base = runs[0]
    stage = np.array([r["counts"] for r in runs], dtype=np.int64)
    np.savez_compressed(os.path.join(outdir, "coarse_sphere_r12_corner511.npz"),
                        corner=np.int64(c), center=np.array(COARSE["center"]), value=np.float64(COARSE["radius2"]),
                        end_points=np.array(COARSE["end_points"], dtype=np.int32),
                        l0_pairs=base["l0_pairs"], l0_xyz=base["l0_xyz"], l0_tris=base["l0_tris"],
                        l1_grid_points=base["l1_grid_points"], l1_triangles=base["l1_triangles"],
                        stage_counts=stage)


if __name__ == "__main__":
    names = sys.argv[1:]
    F = fields()
    if names == ["coarse"]:
        make_coarse()
        sys.exit(0)
    if not names or "two_dots" in names:
        make_two_dots()
    for name, spec in F.items():
        if names and name not in names:
            continue
        make(name, spec)
