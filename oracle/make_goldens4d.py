#!/usr/bin/env python3
"""oracle/make_goldens4d.py -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

4-D goldens from the REAL reference.  contourist/pentatopes.py and morph_geometry.py are Python-2
source, so the package is copied to a temp dir OUTSIDE the repo, translated there with lib2to3 and
imported from there; only numbers are written to tests/golden4d/*.npz.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.normpath(os.path.join(HERE, "..", "tests", "golden4d"))
sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..")))
_TMP = [None]


def reference_modules4d():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError("reference not present")
    if _TMP[0] is None:
        tmp = tempfile.mkdtemp(prefix="contourist_ref4d_")
        shutil.copytree(os.path.join(REFERENCE, "contourist"), os.path.join(tmp, "contourist"))
        pk = os.path.join(tmp, "contourist")
        subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", "pentatopes.py", "morph_geometry.py", "field2d.py",
                               "html_demo.py", "lasso.py"], cwd=pk, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        src = open(os.path.join(pk, "pentatopes.py")).read().replace("for l in 0,1]", "for l in (0,1)]")
        open(os.path.join(pk, "pentatopes.py"), "w").write(src)
        _TMP[0] = tmp
    np.int = int
    np.float = float
    np.sometrue = np.any
    sys.dont_write_bytecode = True
    if _TMP[0] not in sys.path:
        sys.path.insert(0, _TMP[0])
    from contourist import pentatopes
    return pentatopes


def close4(A, fill):
    """force the outermost 2 samples of every axis to `fill` so that no crossing touches the array
    boundary (the reference evaluates f outside the array near boundary crossings)"""
    A = A.copy()
    for ax in range(4):
        sl = [slice(None)] * 4
        for idx in (0, 1, -1, -2):
            sl[ax] = idx
            A[tuple(sl)] = fill
    return A


def fields4d():
    F = {}
    g = np.arange(11, dtype=np.float64)
    X, Y, Z, T = np.meshgrid(g, g, g, np.arange(9, dtype=np.float64), indexing="ij")
    # sphere whose radius grows with t, closed interior in all four axes
    A = ((X - 4.9) ** 2 + (Y - 5.1) ** 2 + (Z - 5.0) ** 2 - 0.9 * T)
    F["paraboloid_11x11x11x9"] = dict(A=close4(A, 60.0).astype(np.float32), value=2.1)
    rng = np.random.RandomState(3)
    B = rng.standard_normal((9, 8, 10, 8))
    for _ in range(2):
        for ax in range(4):
            B = 0.25 * np.roll(B, 1, ax) + 0.5 * B + 0.25 * np.roll(B, -1, ax)
    B = B / B.std()
    F["noise_9x8x10x8"] = dict(A=close4(B, float(B.min()) - 1.0).astype(np.float32), value=0.15)
    # two blobs that merge over time
    C = np.minimum((X - 3.7) ** 2 + (Y - 4.6) ** 2 + (Z - 4.8) ** 2, (X - 6.4) ** 2 + (Y - 5.4) ** 2 + (Z - 5.1) ** 2) - 0.45 * T
    F["merge_11x11x11x9"] = dict(A=close4(C, 60.0).astype(np.float32), value=1.2)
    # two blobs that never meet; explicit end points reach one of them (seeded growth, tetrahedral.py:396-463 with the
    # 80 offsets of pentatopes.py:32-39)
    g10 = np.arange(12, dtype=np.float64)
    X2, Y2, Z2, T2 = np.meshgrid(g10, g10, g10, np.arange(7, dtype=np.float64), indexing="ij")
    E = np.minimum((X2 - 3.2) ** 2 + (Y2 - 3.4) ** 2 + (Z2 - 3.1) ** 2 + 0.3 * (T2 - 3.0) ** 2,
                   (X2 - 8.1) ** 2 + (Y2 - 7.9) ** 2 + (Z2 - 8.3) ** 2 + 0.3 * (T2 - 3.0) ** 2)
    F["two_blobs_seeded_12x12x12x7"] = dict(A=close4(E, 60.0).astype(np.float32), value=2.6,
                                            end_points=[[[3, 3, 3, 3], [3, 3, 6, 3]]])
    # the reference's own demo field (pentatopes.py:528-551 `test0`: period-3 pattern of spheres morphing into bars,
    # value 2.0, with samples exactly equal to the value), on a lattice large enough to close its rim
    g = np.arange(13, dtype=np.float64)
    X, Y, Z, T = np.meshgrid(g % 3, g % 3, g % 3, np.arange(9, dtype=np.float64), indexing="ij")
    p1, p2 = 0.5 * (8 - T), 0.5 * T
    D = p1 * np.sqrt(X * X + Y * Y + Z * Z) + p2 * np.minimum(np.sqrt(X * X + Y * Y), np.sqrt(X * X + Z * Z))
    F["test0_style_13x13x13x9"] = dict(A=close4(D, 60.0).astype(np.float32), value=2.0)
    return F


def run_reference4d(A, value, end_points=None):
    pentatopes = reference_modules4d()
    A = np.ascontiguousarray(A, dtype=np.float32)
    shape = A.shape
    outside = float(A[0, 0, 0, 0])     # beyond the array the field continues with its boundary fill value

    def f(x, y, z, t):
        idx = (int(x), int(y), int(z), int(t))
        if any(i < 0 or i >= n for i, n in zip(idx, shape)):
            return float(outside)
        return float(A[idx])
    t0 = time.time()
    if end_points is None:
        M = pentatopes.MorphingIsoSurfaces([0.0] * 4, [n - 2 for n in shape], [1.0] * 4, f, float(value), [])
        assert tuple(M.grid.grid_dimensions) == tuple(n - 1 for n in shape)
        M.search_for_endpoints()
        cm = M.contour_maker
    else:
        # the reference's own way of calling the 4-D march (test0, pentatopes.py:528-551): explicit end points
        cm = pentatopes.GridContour4D([n - 1 for n in shape], f, float(value), [[list(a), list(b)] for a, b in end_points])
    cm.find_initial_voxels()
    while cm.new_surface_voxels:
        cm.expand_voxels()
    for quad in cm.surface_voxels:
        cm.enumerate_voxel_tetrahedra(quad)
    pair_list = list(cm.interpolated_contour_pairs.keys())
    pair_index = {p: n for n, p in enumerate(pair_list)}
    out = dict(A=A, value=np.float64(value),
               surface_voxels=np.array(sorted(cm.surface_voxels), dtype=np.int32).reshape(-1, 4),
               l0_pairs=np.array([list(p[0]) + list(p[1]) for p in pair_list], dtype=np.int32).reshape(-1, 8),
               l0_xyzt=np.array([cm.interpolated_contour_pairs[p] for p in pair_list], dtype=np.float64).reshape(-1, 4),
               l0_tets=np.array([[pair_index[p] for p in s] for s in cm.simplex_sets], dtype=np.int64).reshape(-1, 4))
    # B3: bin_times / drop_instant_tetrahedra / remove_tiny_simplices (pentatopes.py:107, 122-125)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        cm.bin_times()
        out["b3_xyzt_binned"] = np.array([cm.interpolated_contour_pairs[p] for p in pair_list], dtype=np.float64).reshape(-1, 4)
        cm.drop_instant_tetrahedra()
        out["n_tets_after_drop"] = np.int64(len(cm.simplex_sets))
        cm.remove_tiny_simplices(epsilon=1e-3)
        out["n_tets_after_tiny"] = np.int64(len(cm.simplex_sets))
        # B4/B5: morph triangles (pentatopes.py:314-368, morph_geometry.py:5-89, 145-237)
        mt = cm.collect_morph_triangles()
    pts = np.array(mt.points4d, dtype=np.float64).reshape(-1, 4)
    # reference vertex numbering = dict order of interpolated_contour_pairs at that time; map to edge keys
    pl = list(cm.interpolated_contour_pairs.keys())
    out["mt_point_pairs"] = np.array([list(p[0]) + list(p[1]) for p in pl], dtype=np.int32).reshape(-1, 8)
    out["mt_points4d"] = pts
    out["mt_segments"] = np.array(mt.segment_point_indices, dtype=np.int64).reshape(-1, 2)
    out["mt_triangles"] = np.array([list(t) for t in mt.triangle_segment_indices], dtype=np.int64).reshape(-1, 3)
    out["t_total_s"] = np.float64(time.time() - t0)
    return out


def test0_field(x, y, z, t):
    "the field of the reference's own 4-D demo (pentatopes.py:528-551, `test0`), restated: works on scalars and arrays"
    x, y, z = np.mod(x, 3), np.mod(y, 3), np.mod(z, 3)
    p1, p2 = 0.5 * (8 - t), 0.5 * t
    return p1 * np.sqrt(x * x + y * y + z * z) + p2 * np.minimum(np.sqrt(x * x + y * y), np.sqrt(x * x + z * z))


TEST0_END_POINTS = [([0] * 4, [4] * 3 + [0]), ([3, 2, 1, 0], [3, 3, 3, 8])]


def make_test0():
    """the reference's own call: GridContour4D([8]*4, function, 2.0, endpoints) with a CALLABLE evaluated in float64
    (two of its start voxels lie outside the grid).  Stores the Level-0 snapshot only (edges and tetrahedra)."""
    import contextlib
    import io
    pentatopes = reference_modules4d()

    def f(x, y, z, t):
        return float(test0_field(float(x), float(y), float(z), float(t)))
    with contextlib.redirect_stdout(io.StringIO()):
        G = pentatopes.GridContour4D([8] * 4, f, 2.0, [(list(a), list(b)) for a, b in TEST0_END_POINTS])
        G.find_initial_voxels()
        while G.new_surface_voxels:
            G.expand_voxels()
        for q in G.surface_voxels:
            G.enumerate_voxel_tetrahedra(q)
    pair_list = list(G.interpolated_contour_pairs.keys())
    pair_index = {p: n for n, p in enumerate(pair_list)}
    pairs = np.array([list(p[0]) + list(p[1]) for p in pair_list], dtype=np.int32).reshape(-1, 8)
    tets = np.array([[pair_index[p] for p in s] for s in G.simplex_sets], dtype=np.int64).reshape(-1, 4)
    sv = np.array(sorted(tuple(int(x) for x in v) for v in G.surface_voxels), dtype=np.int32)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "reference_test0_seeded.npz"), l0_pairs=pairs, l0_tets=tets, surface_voxels=sv,
                        end_points=np.array(TEST0_END_POINTS, dtype=np.int32), value=np.float64(2.0))
    print("reference_test0_seeded: %d hyper-voxels, %d vertices, %d tetrahedra" % (len(sv), len(pairs), len(tets)))


def open_rim_field(x, y, z, t):
    "a ball that leaves the grid through three of its faces and moves in time; works on scalars and arrays"
    return np.sqrt((x - 0.8) ** 2 + (y - 3.1) ** 2 + (z - 5.7) ** 2 + 0.6 * (t - 1.2) ** 2) - 2.55


OPEN_RIM = dict(mins=[0.0] * 4, maxes=[5.5, 5.5, 6.5, 3.5], delta=[1.0] * 4, value=0.0)   # grid_dimensions (6, 6, 7, 4)


def make_open_rim():
    """MorphingIsoSurfaces(mins, maxes, delta, CALLABLE, value, []).search_for_endpoints() of the reference on a surface that
    LEAVES the grid: the exhaustive search starts from every crossing lattice segment and does not range-check the voxels
    it starts from, so hyper-voxels one lattice step outside the grid get tetrahedra too.  Level-0 snapshot only."""
    import contextlib
    import io
    pentatopes = reference_modules4d()

    def f(x, y, z, t):
        return float(open_rim_field(float(x), float(y), float(z), float(t)))
    with contextlib.redirect_stdout(io.StringIO()):
        M = pentatopes.MorphingIsoSurfaces(OPEN_RIM["mins"], OPEN_RIM["maxes"], OPEN_RIM["delta"], f, OPEN_RIM["value"], [])
        M.search_for_endpoints()
        G = M.contour_maker
        G.find_initial_voxels()
        while G.new_surface_voxels:
            G.expand_voxels()
        for q in G.surface_voxels:
            G.enumerate_voxel_tetrahedra(q)
    pair_list = list(G.interpolated_contour_pairs.keys())
    pair_index = {p: n for n, p in enumerate(pair_list)}
    pairs = np.array([list(p[0]) + list(p[1]) for p in pair_list], dtype=np.int32).reshape(-1, 8)
    tets = np.array([[pair_index[p] for p in s] for s in G.simplex_sets], dtype=np.int64).reshape(-1, 4)
    sv = np.array(sorted(tuple(int(x) for x in v) for v in G.surface_voxels), dtype=np.int32)
    gd = np.array([int(n) for n in M.grid.grid_dimensions], dtype=np.int32)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "reference_open_rim_seeded.npz"), l0_pairs=pairs, l0_tets=tets, surface_voxels=sv,
                        grid_dimensions=gd, value=np.float64(OPEN_RIM["value"]))
    outside = int(((sv < 0) | (sv >= gd)).any(axis=1).sum())
    print("reference_open_rim_seeded: grid %s, %d hyper-voxels (%d outside the grid), %d vertices, %d tetrahedra" % (
        tuple(gd), len(sv), outside, len(pairs), len(tets)))


def refined_field(x, y, z, t):
    "a closed blob well inside the grid, not linear along the lattice edges; works on scalars and arrays"
    return (x - 3.4) ** 2 + 0.8 * (y - 3.1) ** 2 + 1.3 * (z - 2.8) ** 2 + 2.0 * (t - 2.2) ** 2 + 0.3 * np.sin(1.3 * x + 0.7 * t) - 3.9


REFINED = dict(mins=[0.0] * 4, maxes=[6.5, 6.5, 5.5, 4.5], delta=[1.0] * 4, value=0.0)   # grid_dimensions (7, 7, 6, 5)


def make_refined():
    """MorphingIsoSurfaces(mins, maxes, delta, CALLABLE, value, [], linear_interpolate=False) of the reference: every crossing
    point is refined with up to 5 regula-falsi steps on the callable (tetrahedral.py:488-505).  Level-0 snapshot with the refined
    points, then bin_times / drop_instant / tiny collapse (pentatopes.py:107, 122-125)."""
    import contextlib
    import io
    pentatopes = reference_modules4d()

    def f(x, y, z, t):
        return float(refined_field(float(x), float(y), float(z), float(t)))
    with contextlib.redirect_stdout(io.StringIO()):
        M = pentatopes.MorphingIsoSurfaces(REFINED["mins"], REFINED["maxes"], REFINED["delta"], f, REFINED["value"], [],
                                           linear_interpolate=False)
        # (the reference's constructor loses the flag: the base constructor it calls last resets it to its default True,
        # pentatopes.py:82-83 -- set it again, as a caller who wants the refinement has to)
        M.linear_interpolate = False
        M.search_for_endpoints()
        G = M.contour_maker
        assert not G.linear_interpolate
        G.find_initial_voxels()
        while G.new_surface_voxels:
            G.expand_voxels()
        for q in G.surface_voxels:
            G.enumerate_voxel_tetrahedra(q)
        pair_list = list(G.interpolated_contour_pairs.keys())
        pair_index = {p: n for n, p in enumerate(pair_list)}
        out = dict(l0_pairs=np.array([list(p[0]) + list(p[1]) for p in pair_list], dtype=np.int32).reshape(-1, 8),
                   l0_xyzt=np.array([G.interpolated_contour_pairs[p] for p in pair_list], dtype=np.float64).reshape(-1, 4),
                   l0_tets=np.array([[pair_index[p] for p in s] for s in G.simplex_sets], dtype=np.int64).reshape(-1, 4),
                   surface_voxels=np.array(sorted(tuple(int(x) for x in v) for v in G.surface_voxels), dtype=np.int32))
        G.bin_times()
        out["b3_xyzt_binned"] = np.array([G.interpolated_contour_pairs[p] for p in pair_list], dtype=np.float64).reshape(-1, 4)
        G.drop_instant_tetrahedra()
        out["n_tets_after_drop"] = np.int64(len(G.simplex_sets))
        G.remove_tiny_simplices(epsilon=1e-3)
        out["n_tets_after_tiny"] = np.int64(len(G.simplex_sets))
    gd = np.array([int(n) for n in M.grid.grid_dimensions], dtype=np.int32)
    sv = out["surface_voxels"]
    assert not ((sv < 0) | (sv >= gd)).any(), "the fixture is meant to stay inside the grid"
    np.savez_compressed(os.path.join(GOLDEN_DIR, "reference_refined_seeded.npz"), grid_dimensions=gd, value=np.float64(REFINED["value"]), **out)
    lin = out["l0_pairs"][:, :4] + 0.5 * (out["l0_pairs"][:, 4:] - out["l0_pairs"][:, :4])
    print("reference_refined_seeded: grid %s, %d hyper-voxels, %d vertices, %d tetrahedra (%d after drop, %d after tiny); refined points move up to %.3f from the edge midpoints" % (
        tuple(int(n) for n in gd), len(sv), len(out["l0_pairs"]), len(out["l0_tets"]), out["n_tets_after_drop"], out["n_tets_after_tiny"],
        float(np.abs(out["l0_xyzt"] - lin).max())))


if __name__ == "__main__":
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    names = sys.argv[1:]
    if names == ["refined"]:
        make_refined()
        sys.exit(0)
    if names == ["test0"]:
        make_test0()
        sys.exit(0)
    if names == ["open_rim"]:
        make_open_rim()
        sys.exit(0)
    for name, spec in fields4d().items():
        if names and name not in names:
            continue
        G = run_reference4d(spec["A"], spec["value"], spec.get("end_points"))
        if "end_points" in spec:
            G["end_points"] = np.array(spec["end_points"], dtype=np.int32)
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **G)
        print("%-22s shape=%s v=%g  hypervoxels %d  verts %d  tets %d | after drop %d after tiny %d | morph: %d segments %d triangles (%.1fs)" % (
            name, spec["A"].shape, spec["value"], len(G["surface_voxels"]), len(G["l0_pairs"]), len(G["l0_tets"]),
            G["n_tets_after_drop"], G["n_tets_after_tiny"], len(G["mt_segments"]), len(G["mt_triangles"]), G["t_total_s"]))
