"""GPU: 4-D hyper-voxel march (pentatopes) through the C ABI vs the 4-D oracle and vs the vectors of the
real reference: edge sets, tetrahedra sets (incl. the hash-order 2-3 splits) bit-exact, coordinates
within 1e-6 relative."""
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
G4 = os.path.join(ROOT, "tests", "golden4d")


def names():
    # (fixtures named *seeded* were made with explicit end points: the reference reached a subset; own tests below)
    return sorted(f[:-4] for f in os.listdir(G4) if f.endswith(".npz") and "seeded" not in f) if os.path.isdir(G4) else []


def check(A, v, diagonal, diag_mode):
    from contourist_amd import pentatopes
    from oracle import level0_4d
    corner = tuple(n - 1 for n in A.shape)
    R = pentatopes.GridContour4D(corner, A, v, diagonal=diagonal).march()
    O = level0_4d.march4d(A, v, diag_mode=diag_mode)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    co = level0_4d.canonical4(ko, O["xyzt"], O["tets"])
    ch = level0_4d.canonical4(R["keys"].astype(np.int64), R["xyzt"], R["tetrahedra"].astype(np.int64))
    assert R["counts"]["n_vertices"] == len(ko) and R["counts"]["n_tetrahedra"] == len(O["tets"])
    assert R["counts"]["n_border_voxels"] == O["nborder_mixed"]
    assert np.array_equal(co[0], ch[0])
    assert np.all(np.abs(ch[1] - co[1]) <= 1e-6 * np.abs(co[1]) + 1e-6)
    assert np.array_equal(co[2], ch[2])
    return R, ch


@pytest.mark.parametrize("name", names())
def test_golden4d(name):
    from oracle import level0_4d
    G = np.load(os.path.join(G4, name + ".npz"))
    A, v = G["A"], float(G["value"])
    R, ch = check(A, v, "cpython310", 1)
    check(A, v, "canonical", 0)
    kr = level0_4d.edge_keys4(G["l0_pairs"], A.shape)
    cr = level0_4d.canonical4(kr, G["l0_xyzt"], G["l0_tets"])
    assert np.array_equal(cr[0], ch[0]) and np.array_equal(cr[2], ch[2])        # the real reference's snapshot


@pytest.mark.parametrize("shape,seed", [((3, 4, 5, 6), 1), ((2, 2, 2, 2), 2), ((9, 5, 4, 7), 3), ((6, 6, 6, 17), 4),
                                        ((5, 4, 3, 33), 5), ((4, 3, 5, 64), 6), ((3, 3, 4, 70), 7), ((18, 17, 16, 20), 8),
                                        ((5, 4, 3, 32), 9), ((3, 2, 2, 96), 10), ((12, 11, 10, 32), 11)])
def test_random_open_boundary_4d(shape, seed):
    rng = np.random.RandomState(seed)
    A = rng.standard_normal(shape).astype(np.float32)
    check(A, 0.1, "cpython310", 1)


def test_tolerances_4d():
    rng = np.random.RandomState(5)
    B = (rng.standard_normal((5, 6, 5, 6)) * 3e-9).astype(np.float32)
    check(B, 0.0, "cpython310", 1)
    C = np.round(rng.standard_normal((5, 5, 6, 6)) * 2) / 2
    check(C.astype(np.float32), 0.5, "cpython310", 1)


@pytest.mark.parametrize("name", names())
def test_find_tetrahedra_post_steps(name):
    """bin_times / drop_instant_tetrahedra / remove_tiny_simplices on the device vs the oracle and the reference"""
    from contourist_amd import pentatopes
    from oracle import level0_4d, postpass4d
    G = np.load(os.path.join(G4, name + ".npz"))
    A, v = G["A"], float(G["value"])
    corner = np.array(A.shape) - 1
    R = pentatopes.GridContour4D(tuple(corner), A, v).find_tetrahedra()
    O = level0_4d.march4d(A, v, diag_mode=1)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    W = postpass4d.find_tetrahedra_post(ko, O["xyzt"], O["tets"], corner)
    assert R["counts"]["n_after_drop"] == W["n_after_drop"] == int(G["n_tets_after_drop"])
    assert R["counts"]["n_after_tiny"] == W["n_after_tiny"] == int(G["n_tets_after_tiny"])
    kh = R["keys"].astype(np.int64)
    assert np.array_equal(R["points4d"][np.argsort(kh)], W["xyzt"][np.argsort(ko)])          # float64, bit for bit
    got = level0_4d.canonical4(kh, R["points4d"], R["tetrahedra"].astype(np.int64))[2]
    want = level0_4d.canonical4(ko, W["xyzt"], W["tets"])[2]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", names())
def test_morph_triangles_device(name):
    """collect_morph_triangles on the device vs the oracle's canonical restatement and the reference's output"""
    from contourist_amd import pentatopes
    from oracle import level0_4d, postpass4d
    G = np.load(os.path.join(G4, name + ".npz"))
    A, v = G["A"], float(G["value"])
    corner = np.array(A.shape) - 1
    maker = pentatopes.GridContour4D(tuple(corner), A, v)
    R = maker.find_tetrahedra()
    MT = maker.collect_morph_triangles()
    kh = R["keys"].astype(np.int64)
    O = level0_4d.march4d(A, v, diag_mode=1)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    W = postpass4d.find_tetrahedra_post(ko, O["xyzt"], O["tets"], corner)
    M = postpass4d.collect_morph_triangles(ko, W["xyzt"], W["tets"])
    # same canonical rules on both sides => identical segments (with direction) and identical triangles
    got_seg = set((int(kh[i]), int(kh[j])) for i, j in MT.segment_point_indices)
    want_seg = set((int(M["keys"][i]), int(M["keys"][j])) for i, j in M["segments"])
    assert got_seg == want_seg
    assert len(MT.triangle_segment_indices) == len(M["triangles"]) == len(G["mt_triangles"])

    def tri_sets(keys, segs, tris):
        sk = [tuple(sorted((int(keys[i]), int(keys[j])))) for i, j in segs]
        return set(frozenset(sk[s] for s in t) for t in tris)
    assert tri_sets(kh, MT.segment_point_indices, MT.triangle_segment_indices) == tri_sets(M["keys"], M["segments"], M["triangles"])
    ot, label, flags = postpass4d.orient_morph_triangles(M)
    common, agree = postpass4d.winding_agreement(kh, MT.segment_point_indices, MT.triangle_segment_indices, M["keys"], M["segments"], ot)
    # the march's table emits every tetrahedron wound low -> high (tools/gen_tables.py orient_tet, exact for any sample values
    # of a sign pattern) and every slice is wound from that (cx_post.hip cxp_morph_slices): the same windings as the
    # reference's flood fill, also on the fixture with samples EQUAL to the isovalue (test0_style: zero-length edges)
    assert common == len(ot) and agree == common, (common, agree)
    # and against the real reference
    rk = level0_4d.edge_keys4(G["mt_point_pairs"], A.shape)
    assert set((int(rk[i]), int(rk[j])) for i, j in G["mt_segments"]) == got_seg
    assert postpass4d.morph_polygons(rk, G["mt_segments"], G["mt_triangles"]) == \
        postpass4d.morph_polygons(kh, MT.segment_point_indices, MT.triangle_segment_indices)
    common, agree = postpass4d.winding_agreement(kh, MT.segment_point_indices, MT.triangle_segment_indices, rk, G["mt_segments"], G["mt_triangles"])
    assert common > 0.7 * len(ot) and agree == common, (common, agree)
    # Where two triangles are alone on a segment at some time, both windings are consistent (opposite directions along the
    # shared edge) -- for the device and for the reference:
    bad_d, seen_d = postpass4d.forced_pair_violations(kh, MT.segment_point_indices, MT.triangle_segment_indices, MT.points4d)
    bad_r, seen_r = postpass4d.forced_pair_violations(rk, G["mt_segments"], G["mt_triangles"], G["mt_points4d"])
    assert bad_r == 0 and seen_d > 0 and bad_d == 0, (bad_d, seen_d)
    print(name, "4-D winding: common", common, "agree", agree, "(%.2f %%)" % (100.0 * agree / common), "forced pairs checked", seen_d, seen_r,
          "inconsistent on the device", bad_d)
    assert np.array_equal(MT.points4d[np.argsort(kh)], G["mt_points4d"][np.argsort(rk)])
    # B6: the surface at a time t is a closed 3-D mesh where it exists (every edge shared by two triangles)
    tmid = 0.5 * (MT.min_value + MT.max_value) + 0.013
    pts, tris = MT.triangles_at(tmid)
    assert len(tris) > 0 and tris.max() < len(pts)


@pytest.mark.parametrize("name", names())
def test_per_t_surfaces_against_the_viewer_oracle(name):
    """B6: the surface at time t from the morph triangles -- cx_morph_eval against oracle/morph_eval.py, the restatement of
    the reference's viewer (misc/morph_triangles.js:53-84 triangle intervals, :117-147 active set, :156-178 points on the
    segments), on the device's own morph triangles AND on the reference's (tests/golden4d: mt_*)."""
    from contourist_amd import pentatopes, morph_geometry
    from oracle import morph_eval
    G = np.load(os.path.join(G4, name + ".npz"))
    A, v = G["A"], float(G["value"])
    maker = pentatopes.GridContour4D(tuple(np.array(A.shape) - 1), A, v)
    maker.find_tetrahedra()
    MT = maker.collect_morph_triangles()

    def tri_points(points, tris):
        "triangles as triples of 3-D points, rotated to start at the smallest one (winding kept), sorted"
        out = []
        for t in np.asarray(tris):
            p = [tuple(np.round(points[k], 9).tolist()) for k in t]
            r = p.index(min(p))
            out.append((p[r], p[(r + 1) % 3], p[(r + 2) % 3]))
        return sorted(out)
    n_nonempty = 0
    lo, hi = float(MT.min_value), float(MT.max_value)
    for frac in (0.013, 0.137, 0.291, 0.419, 0.503, 0.677, 0.811, 0.953):      # generic times: no vertex time is hit exactly
        t = lo + frac * (hi - lo)
        pd, td = maker.triangles_at(t)                                          # device: cx_morph_eval
        W = morph_eval.surface_at(MT.points4d, MT.segment_point_indices, MT.triangle_segment_indices, t)
        assert len(td) == len(W["faces"]) and len(pd) == len(W["points"])
        assert tri_points(pd, td) == tri_points(W["points"], W["faces"])
        # the host-side numpy evaluation of the mirrored class agrees as well
        ph, th = MT.triangles_at(t)
        assert np.array_equal(th, td) and np.allclose(ph, pd, rtol=0, atol=1e-12)
        n_nonempty += len(td) > 0
    assert n_nonempty >= 3
    if "mt_points4d" in G:
        # the reference's own morph triangles through the mirrored class (host) and the oracle
        R = morph_geometry.MorphTriangles(G["mt_points4d"], G["mt_segments"], G["mt_triangles"])
        for frac in (0.213, 0.577):
            t = lo + frac * (hi - lo)
            ph, th = R.triangles_at(t)
            W = morph_eval.surface_at(R.points4d, R.segment_point_indices, R.triangle_segment_indices, t)
            assert tri_points(ph, th) == tri_points(W["points"], W["faces"])


def test_time_slices_are_consistently_wound_at_size():
    """64 x 64 x 64 x 32 (two moving blobs): every time slice of the morph triangles is a surface whose manifold edges
    are run in opposite directions by their two triangles (the reference's orient_triangles aims at exactly that,
    surface_geometry.py:110-138) -- none may be run in the same direction"""
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi
    from test_gpu_fullsize import edge_consistency
    dev = torch.device("cuda", 0)
    shape = (64, 64, 64, 32)
    ax = [torch.arange(n, device=dev, dtype=torch.float32) / (n - 1) for n in shape]
    X, Y, Z, T = torch.meshgrid(*ax, indexing="ij")
    A = torch.exp(-(((X - 0.30 - 0.35 * T) ** 2 + (Y - 0.35 - 0.2 * T) ** 2 + (Z - 0.5) ** 2) / (2 * 0.12 ** 2))) + \
        torch.exp(-(((X - 0.70 + 0.30 * T) ** 2 + (Y - 0.65 + 0.2 * T) ** 2 + (Z - 0.45 - 0.1 * T) ** 2) / (2 * 0.10 ** 2)))
    for axis in range(4):
        for idx in (0, 1, -1, -2):
            A.select(axis, idx).fill_(0.0)
    A = A.contiguous()
    ctx = _ffi.Context(0)
    try:
        ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
        ctx.extract4d(0.5, 1)
        ctx.postprocess4d(100)
        mt = ctx.morph_triangles()
        tmin, tmax = float(mt[0][:, 3].min()), float(mt[0][:, 3].max())
        for frac in (0.13, 0.37, 0.52, 0.81):
            pts, tris = ctx.morph_eval(tmin + frac * (tmax - tmin))
            manifold, same, other = edge_consistency(tris)
            assert len(tris) > 10000 and manifold > 1.3 * len(tris) and same == 0
    finally:
        ctx.close()


def test_seeded_growth_4d():
    """explicit end points in 4-D (cx_select_seeded4d): the device keeps exactly the tetrahedra the restated search keeps
    and the mirrored GridContour4D(corner, samples, value, end points) returns what the real reference returned"""
    from contourist_amd import _ffi, pentatopes
    from oracle import level0_4d, seeds
    G = np.load(os.path.join(G4, "two_blobs_seeded_12x12x12x7.npz"))
    A, v, eps = G["A"], float(G["value"]), G["end_points"]
    corner = np.array(A.shape) - 1
    maker = pentatopes.GridContour4D(tuple(corner), A, v, [[tuple(a), tuple(b)] for a, b in eps.tolist()])
    L = maker.march()
    kh = L["keys"].astype(np.int64)
    keep, surf = seeds.select4d(A, v, eps, kh, L["tetrahedra"].astype(np.int64))
    R = maker.find_tetrahedra()
    kh = R["keys"].astype(np.int64)      # find_tetrahedra marches again: the 4-D vertex numbering is per extraction
    assert maker.seeded["tetrahedra_kept"] == int(keep.sum()) == len(G["l0_tets"]) and maker.seeded["groups_kept"] == 1
    assert R["counts"]["n_after_drop"] == int(G["n_tets_after_drop"]) and R["counts"]["n_after_tiny"] == int(G["n_tets_after_tiny"])
    kr = level0_4d.edge_keys4(G["l0_pairs"], A.shape)
    MT = maker.collect_morph_triangles()
    assert len(MT.triangle_segment_indices) == len(G["mt_triangles"]) and len(MT.segment_point_indices) == len(G["mt_segments"])
    rk = level0_4d.edge_keys4(G["mt_point_pairs"], A.shape)
    # with explicit end points the morph triangles carry only the points the reference's search interpolated (its
    # interpolated_contour_pairs): exactly as many as the reference's own MorphTriangles holds, numbered 0..n-1;
    # maker.morph_vertex_ids maps them back to the Level-0 vertices
    assert len(MT.points4d) == len(G["mt_points4d"]) == len(maker.morph_vertex_ids)
    km = kh[maker.morph_vertex_ids]
    assert set(int(k) for k in km) == set(int(k) for k in rk)
    got_seg = set((int(km[i]), int(km[j])) for i, j in MT.segment_point_indices)
    assert got_seg == set((int(rk[i]), int(rk[j])) for i, j in G["mt_segments"])
    # end points on the other blob select the other component; both together everything
    ctx = maker.context()
    other = ctx.select_seeded4d([[(8, 8, 8, 3), (8, 8, 11, 3)]])
    both = ctx.select_seeded4d([[(3, 3, 3, 3), (3, 3, 6, 3)], [(8, 8, 8, 3), (8, 8, 11, 3)]])
    assert other["tetrahedra_kept"] == len(L["tetrahedra"]) - int(keep.sum()) and both["tetrahedra_kept"] == len(L["tetrahedra"])
    assert both["groups_kept"] == 2
    with pytest.raises(_ffi.CxError):
        ctx.select_seeded4d([[(0, 0, 0, 0), (1, 0, 0, 0)]])       # both on the same side


def test_reference_test0_call_on_device():
    """the reference's own 4-D demo call (pentatopes.py:528-551) through the mirrored class, with the CALLABLE: the device
    returns the reference's 26 100 tetrahedra, the 96 included that sit in two start voxels one lattice step OUTSIDE the
    grid (the reference does not range-check the voxels it starts from, tetrahedral.py:396-441): same edges, same
    tetrahedra with the CPython-order splits"""
    from contourist_amd import pentatopes
    from oracle.make_goldens4d import test0_field, TEST0_END_POINTS
    G = np.load(os.path.join(G4, "reference_test0_seeded.npz"))
    maker = pentatopes.GridContour4D((8, 8, 8, 8), test0_field, 2.0, [(tuple(a), tuple(b)) for a, b in TEST0_END_POINTS])
    assert maker.origin == (-1, -1, -1, -1) and maker.shape == (11, 11, 11, 11)
    L = maker.march()
    ctx = maker.context()
    assert ctx.select_seeded4d(maker.end_points, maker.voxel_range)["tetrahedra_kept"] == len(G["l0_tets"]) == 26100
    keep = ctx.seeded4d_mask(L["counts"]).astype(bool)
    lo, hi = pentatopes.unpack_edge_ids4(L["keys"], maker.shape)
    lo, hi = lo - 1, hi - 1                                     # array lattice -> the reference's
    code = lambda P: [tuple(int(x) for x in r) for r in P]
    dev_pair = [(a, b) for a, b in zip(code(lo), code(hi))]
    ref_pair = [tuple(sorted((tuple(int(x) for x in r[:4]), tuple(int(x) for x in r[4:])))) for r in G["l0_pairs"]]
    dev_tets = set(frozenset(dev_pair[v] for v in t) for t in L["tetrahedra"][keep])
    ref_tets = set(frozenset(ref_pair[v] for v in t) for t in G["l0_tets"])
    assert len(dev_tets) == int(keep.sum()) == 26100 and dev_tets == ref_tets
    P = G["l0_pairs"]
    inside = (P.min(axis=1) >= 0) & (P.max(axis=1) <= 8)
    assert int(inside[G["l0_tets"]].all(axis=1).sum()) == 26004       # the other 96 reach beyond the grid
    used = np.unique(L["tetrahedra"][keep])
    assert L["xyzt"][used].min() < 0.0 or L["xyzt"][used].max() > 8.0  # coordinates in the reference's lattice, rim included
    R = maker.find_tetrahedra()
    assert maker.seeded["tetrahedra_kept"] == 26100
    assert R["counts"]["n_after_tiny"] > 0 and R["points4d"][np.unique(R["tetrahedra"])].min() >= -1.0
    MT = maker.collect_morph_triangles()
    assert len(MT.triangle_segment_indices) > 0
    # the same field as an ARRAY cannot be evaluated outside itself: the growth keeps to the grid (26 004 tetrahedra)
    g = np.arange(9, dtype=np.float64)
    X, Y, Z, T = np.meshgrid(g, g, g, g, indexing="ij")
    A = test0_field(X, Y, Z, T).astype(np.float32)
    maker2 = pentatopes.GridContour4D((8, 8, 8, 8), A, 2.0, [(tuple(a), tuple(b)) for a, b in TEST0_END_POINTS])
    maker2.find_tetrahedra()
    assert maker2.seeded["tetrahedra_kept"] == 26004


def test_open_surface_with_callable_reaches_the_rim_4d():
    """MorphingIsoSurfaces(..., CALLABLE, ...).search_for_endpoints() on a surface that leaves the grid: the reference starts from
    every crossing lattice segment and does not range-check the hyper-voxels it starts from (tetrahedral.py:396-441), so 109 of
    its 352 hyper-voxels lie one lattice step OUTSIDE the grid.  Golden: the real reference (oracle/make_goldens4d.py open_rim)."""
    from contourist_amd import pentatopes
    from oracle.make_goldens4d import open_rim_field, OPEN_RIM
    G = np.load(os.path.join(G4, "reference_open_rim_seeded.npz"))
    M = pentatopes.MorphingIsoSurfaces(OPEN_RIM["mins"], OPEN_RIM["maxes"], OPEN_RIM["delta"], open_rim_field, OPEN_RIM["value"], [])
    assert tuple(int(n) for n in M.grid.grid_dimensions) == tuple(int(n) for n in G["grid_dimensions"])
    M.search_for_endpoints()
    maker = M.contour_maker
    assert maker.origin == (-1, -1, -1, -1) and maker.keep_in_range
    L = maker.march()
    ctx = maker.context()
    assert ctx.select_seeded4d(maker.end_points, maker.voxel_range, True)["tetrahedra_kept"] == len(G["l0_tets"])
    keep = ctx.seeded4d_mask(L["counts"]).astype(bool)
    lo, hi = pentatopes.unpack_edge_ids4(L["keys"], maker.shape)
    lo, hi = lo - 1, hi - 1
    code = lambda P: [tuple(int(x) for x in r) for r in P]
    dev_pair = [(a, b) for a, b in zip(code(lo), code(hi))]
    ref_pair = [tuple(sorted((tuple(int(x) for x in r[:4]), tuple(int(x) for x in r[4:])))) for r in G["l0_pairs"]]
    dev_tets = set(frozenset(dev_pair[v] for v in t) for t in L["tetrahedra"][keep])
    ref_tets = set(frozenset(ref_pair[v] for v in t) for t in G["l0_tets"])
    assert dev_tets == ref_tets and len(dev_tets) == int(keep.sum())
    sv = G["surface_voxels"]
    assert int(((sv < 0) | (sv >= G["grid_dimensions"])).any(axis=1).sum()) == 109
    assert ctx.seeded_mode() == "sequential"
    # one thread per end point pair (what more than 16 384 pairs get): no shared `visited` set.  The hyper-voxels of the GRID are
    # the same (all kept); in the RIM each point then takes its own voxel or its first border neighbour, where the reference's
    # later points move on to the next unvisited candidate -- fewer rim voxels; the caller can ask which mode ran
    par = ctx.select_seeded4d(maker.end_points, maker.voxel_range, True, parallel=True)
    assert ctx.seeded_mode() == "parallel"
    keep_p = ctx.seeded4d_mask(L["counts"]).astype(bool)
    lo4, _ = pentatopes.unpack_edge_ids4(L["keys"], maker.shape)
    cell = lo4[L["tetrahedra"]].min(axis=1) - 1                       # hyper-voxel of a tetrahedron, the reference's lattice
    in_grid = ((cell >= 0) & (cell < G["grid_dimensions"])).all(axis=1)
    assert np.array_equal(keep_p[in_grid], keep[in_grid]) and keep[in_grid].all()
    assert 0 < int(keep_p[~in_grid].sum()) <= int(keep[~in_grid].sum()) and par["tetrahedra_kept"] == int(keep_p.sum())
    ctx.select_seeded4d(maker.end_points, maker.voxel_range, True)    # back to the reference's order for what follows
    T = M.collect_morph_triangles()          # the whole path on the rimmed array, world coordinates
    assert len(T.triangle_segment_indices) > 0


def test_linear_interpolate_false_4d():
    """linear_interpolate=False in 4-D: every crossing point is refined with the reference's regula falsi on the CALLABLE
    (tetrahedral.py:488-505), then bin_times / drop_instant / tiny collapse run on the refined points.  Golden: the real
    reference (oracle/make_goldens4d.py refined)."""
    from contourist_amd import pentatopes
    from oracle.make_goldens4d import refined_field, REFINED
    G = np.load(os.path.join(G4, "reference_refined_seeded.npz"))
    M = pentatopes.MorphingIsoSurfaces(REFINED["mins"], REFINED["maxes"], REFINED["delta"], refined_field, REFINED["value"], [],
                                       linear_interpolate=False)
    M.search_for_endpoints()
    maker = M.contour_maker
    assert not maker.linear_interpolate     # (the surface comes within a lattice step of the t = 0 face: the array carries a rim)
    R = maker.find_tetrahedra()
    assert R["counts"]["n_after_drop"] == int(G["n_tets_after_drop"]) and R["counts"]["n_after_tiny"] == int(G["n_tets_after_tiny"])
    ctx = maker.context()
    L0 = ctx.download_level0_4d(maker._counts)
    keep = ctx.seeded4d_mask(maker._counts).astype(bool) if maker.end_points is not None and len(maker.end_points) else np.ones(len(L0[2]), bool)
    lo, hi = pentatopes.unpack_edge_ids4(R["keys"], maker.shape)
    lo, hi = lo + np.asarray(maker.origin), hi + np.asarray(maker.origin)
    used = np.zeros(len(lo), bool)
    used[L0[2][keep].ravel()] = True
    dev = dict(((tuple(int(x) for x in a), tuple(int(x) for x in b)), n) for n, (a, b) in enumerate(zip(lo, hi)) if used[n])
    ref_pair = [tuple(sorted((tuple(int(x) for x in r[:4]), tuple(int(x) for x in r[4:])))) for r in G["l0_pairs"]]
    assert set(dev) == set(ref_pair)
    idx = np.array([dev[p] for p in ref_pair])
    # the refined points themselves, and after bin_times: the host restates the reference's float64 iteration exactly
    refined = maker._refined_points(R["keys"])
    assert np.abs(refined[idx] - G["l0_xyzt"]).max() <= 1e-12
    assert np.abs(R["points4d"][idx] - G["b3_xyzt_binned"]).max() <= 1e-12
    dev_tets = set(frozenset(int(v) for v in t) for t in L0[2][keep])
    ref_tets = set(frozenset(int(idx[v]) for v in t) for t in G["l0_tets"])
    assert dev_tets == ref_tets
    # and it is not the linear interpolation
    M2 = pentatopes.MorphingIsoSurfaces(REFINED["mins"], REFINED["maxes"], REFINED["delta"], refined_field, REFINED["value"], [])
    M2.search_for_endpoints()
    R2 = M2.contour_maker.find_tetrahedra()
    o1, o2 = np.argsort(R["keys"]), np.argsort(R2["keys"])      # (the vertex ORDER of the 4-D march varies from run to run)
    assert np.array_equal(R2["keys"][o2], R["keys"][o1]) and np.abs(R2["points4d"][o2] - R["points4d"][o1]).max() > 1e-3
    T = M.collect_morph_triangles()
    assert len(T.triangle_segment_indices) > 0


def test_search_for_endpoints_with_skip_4d():
    """skip > 1 in 4-D: the coarse crossing search seeds the growth; a blob that fits between the coarse lattice points is
    not returned, the large one is (the reference's sparsity mode)"""
    from contourist_amd import pentatopes
    shape = (25, 25, 25, 9)
    ax = [np.arange(n, dtype=np.float64) for n in shape]
    X, Y, Z, T = np.meshgrid(*ax, indexing="ij")
    big = np.sqrt((X - 9.3) ** 2 + (Y - 9.1) ** 2 + (Z - 9.2) ** 2 + 0.5 * (T - 4.0) ** 2) - 5.0
    small = np.sqrt((X - 21.5) ** 2 + (Y - 21.5) ** 2 + (Z - 21.5) ** 2 + (T - 6.0) ** 2) - 1.1      # between the coarse points
    A = np.minimum(big, small).astype(np.float32)
    mins, delta = [0.0] * 4, [1.0] * 4

    def tets(skip):
        M = pentatopes.MorphingIsoSurfaces(mins, None, delta, A, 0.0, [])
        M.search_for_endpoints(skip)
        R = M.contour_maker.find_tetrahedra()
        return R, M
    R_all, _ = tets(1)
    R_big, M = tets(4)
    assert 0 < len(R_big["tetrahedra"]) < len(R_all["tetrahedra"])
    used = np.unique(R_big["tetrahedra"])
    assert np.all(np.linalg.norm(R_big["points4d"][used][:, :3] - np.array([9.3, 9.1, 9.2]), axis=1) < 6.5)   # only the big blob
    used_all = np.unique(R_all["tetrahedra"])
    assert (np.linalg.norm(R_all["points4d"][used_all][:, :3] - 21.5, axis=1) < 2.5).any()                    # exhaustive: both
    assert M.contour_maker.seeded["groups_kept"] == 1


@pytest.mark.parametrize("name,tag", [("two_blobs_seeded_12x12x12x7", "all"), ("two_blobs_seeded_12x12x12x7", "clipped"),
                                      ("paraboloid_11x11x11x9", "clipped")])
def test_device_morph_triangles_through_to_json_into_the_viewer(name, tag):
    """SURVEY 8(f) N2 on the GPU path, end to end: GridContour4D.collect_morph_triangles() on the DEVICE -> MorphTriangles.to_json()
    (morph_geometry.py:91-125) -> the bytes parsed the way the reference's viewer parses them (misc/morph_triangles.js:26-52:
    positions = shift + scale * integer) -> oracle/morph_eval.surface_at (the viewer's :53-204, pinned bit for bit by
    tests/test_oracle_viewer.py) at the viewer goldens' own times == cx_morph_eval on the device, and the header of the JSON
    (counts, shift, scale, min / max value) equals that of the bytes the REAL reference wrote for the fixture."""
    import gzip
    import json
    from contourist_amd import pentatopes
    from oracle import morph_eval
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G = np.load(os.path.join(G4, name + ".npz"))
    A, v = G["A"], float(G["value"])
    eps = [[tuple(a), tuple(b)] for a, b in G["end_points"].tolist()] if "end_points" in G.files else None
    maker = pentatopes.GridContour4D(tuple(np.array(A.shape) - 1), A, v, eps) if eps else pentatopes.GridContour4D(tuple(np.array(A.shape) - 1), A, v)
    maker.find_tetrahedra()
    MT = maker.collect_morph_triangles()
    lo, hi = float(MT.min_value), float(MT.max_value)
    kw = {} if tag == "all" else dict(min_value=lo + 0.25 * (hi - lo), max_value=hi - 0.125 * (hi - lo), maxint=4095)
    text = MT.to_json(**kw)
    mine = json.loads(text)
    ref = json.loads(gzip.open(os.path.join(root, "tests", "golden_wire", "%s.to_json.%s.txt.gz" % (name, tag))).read().decode("ascii"))
    # ---- the header: the device's morph triangles are the reference's (as sets), so every number derived from them agrees
    assert mine["description"] == ref["description"] and mine["counts"] == ref["counts"]
    assert mine["min_value"] == ref["min_value"] and mine["max_value"] == ref["max_value"]
    assert np.allclose(mine["shift"], ref["shift"], rtol=0, atol=1e-12) and np.allclose(mine["scale"], ref["scale"], rtol=1e-12, atol=0)
    assert len(mine["positions"]) == 4 * mine["counts"][0] and len(mine["segments"]) == 2 * mine["counts"][1] and len(mine["triangles"]) == 3 * mine["counts"][2]
    # the quantised positions are the reference's as a multiset (the numbering differs)
    qa = np.array(mine["positions"], dtype=np.int64).reshape(-1, 4)
    qb = np.array(ref["positions"], dtype=np.int64).reshape(-1, 4)
    assert np.array_equal(qa[np.lexsort(qa.T[::-1])], qb[np.lexsort(qb.T[::-1])])
    # ---- what the viewer draws from these bytes, against cx_morph_eval on the device
    P = np.array(mine["shift"], dtype=np.float64) + np.array(mine["scale"], dtype=np.float64) * qa
    seg = np.array(mine["segments"], dtype=np.int64).reshape(-1, 2)
    tri = np.array(mine["triangles"], dtype=np.int64).reshape(-1, 3)
    vpath = os.path.join(root, "tests", "golden_viewer", "viewer_%s_%s.npz" % (name, tag))
    times = np.load(vpath)["times"].tolist() if os.path.exists(vpath) else [lo + f * (hi - lo) for f in (0.3, 0.5, 0.7)]
    step = np.array(mine["scale"], dtype=np.float64)[:3]
    V = np.load(vpath) if os.path.exists(vpath) else None
    drawn = 0
    for n, t in enumerate(times):
        W = morph_eval.surface_at(P, seg, tri, t, float(mine["min_value"]), float(mine["max_value"]))
        pd, td = maker.triangles_at(t)                       # device: cx_morph_eval on the unquantised morph triangles
        # A triangle is drawn while t lies inside its segments' intervals; quantising the TIMES (scale[3] per step) can move an
        # interval's end across t, so the counts agree up to the triangles whose interval ends within one step of t
        tr_min, tr_max, valid = morph_eval.triangle_intervals(P, seg, tri)
        near = int(np.sum(valid & ((np.abs(tr_min - t) <= 2 * mine["scale"][3]) | (np.abs(tr_max - t) <= 2 * mine["scale"][3]))))
        assert abs(len(td) - len(W["faces"])) <= near, (len(td), len(W["faces"]), near)
        if V is not None:                                    # and the reference's own viewer on the reference's bytes drew as many
            assert abs(len(V["faces_%d" % n]) - len(W["faces"])) <= near
        if len(td) and near == 0:
            # every drawn point of the viewer lies within the quantisation step (plus its share of the time step) of a device point
            from scipy.spatial import cKDTree
            d, _ = cKDTree(np.asarray(pd)).query(W["points"])
            # a drawn point sits at lam = (t - lo) / (hi - lo) along its segment: the quantisation moves the end points by up to a
            # step in space, and lo / hi by up to a step in TIME, which moves lam by up to ~3 steps / (hi - lo)
            sid = np.asarray(W["segment_ids"], dtype=np.int64)
            a, b = P[seg[sid, 0]], P[seg[sid, 1]]
            dlam = np.minimum(1.0, 3.0 * mine["scale"][3] / np.maximum(b[:, 3] - a[:, 3], 1e-300))
            bound = 2.0 * np.linalg.norm(step) + np.linalg.norm(b[:, :3] - a[:, :3], axis=1) * dlam + 1e-9
            assert np.all(d <= bound), float((d - bound).max())
        drawn += len(td) > 0
    assert drawn >= 2


def test_per_t_surfaces_on_one_context_across_rebuilt_morphs():
    """cx_morph_eval keeps its flag bytes zeroed between calls (the compaction kernels clear what they consume; a full clear only for
    new morph triangles or a new buffer): surfaces at several times, then ANOTHER field's morph triangles on the same context (more
    segments and triangles, then fewer), then the first field again -- every surface equals the one a fresh context gives, and a
    repeated time gives the same arrays (B6, misc/morph_triangles.js:117-178)"""
    from contourist_amd import _ffi

    def field(shape, seed):
        rng = np.random.RandomState(seed)
        ax = [np.linspace(0.0, 1.0, n, dtype=np.float32) for n in shape]
        X, Y, Z, T = np.meshgrid(*ax, indexing="ij")
        A = np.exp(-(((X - 0.35 - 0.3 * T) ** 2 + (Y - 0.4) ** 2 + (Z - 0.5 + 0.1 * T) ** 2) / (2 * 0.16 ** 2)))
        A += 0.02 * rng.standard_normal(shape)
        for axis in range(4):
            for idx in (0, -1):
                np.moveaxis(A, axis, 0)[idx] = 0.0
        return np.ascontiguousarray(A.astype(np.float32))

    def surfaces(ctx, A, fracs):
        ctx.upload_grid4d(A)
        ctx.extract4d(0.5, 1)
        ctx.postprocess4d(100)
        mt = ctx.morph_triangles()
        tmin, tmax = float(mt[0][:, 3].min()), float(mt[0][:, 3].max())
        out = []
        for fr in fracs:
            p, t = ctx.morph_eval(tmin + fr * (tmax - tmin))
            out.append((p.copy(), t.copy()))
        return out

    def canon(points, tris):
        out = []
        for tr in np.asarray(tris):
            q = [tuple(np.round(points[k], 9).tolist()) for k in tr]
            r = q.index(min(q))
            out.append((q[r], q[(r + 1) % 3], q[(r + 2) % 3]))
        return sorted(out)

    fields = [field((14, 13, 12, 9), 1), field((20, 18, 16, 12), 2), field((9, 10, 11, 6), 3)]
    fracs = (0.21, 0.48, 0.21, 0.77, 0.48)
    fresh = []
    for A in fields:
        c = _ffi.Context(0)
        try:
            fresh.append(surfaces(c, A, fracs))
        finally:
            c.close()
    for S in fresh:
        assert sum(len(t) for _, t in S) > 0
        assert np.array_equal(S[0][1], S[2][1]) and np.array_equal(S[0][0], S[2][0])      # the same time twice
    ctx = _ffi.Context(0)
    try:
        for k in (0, 1, 2, 0, 1):
            got = surfaces(ctx, fields[k], fracs)
            for (p, t), (pf, tf) in zip(got, fresh[k]):
                # (the ORDER of the 4-D Level-0 output follows the workgroups' reservations: compare the surfaces as sets of triangles)
                assert len(t) == len(tf) and len(p) == len(pf)
                assert canon(p, t) == canon(pf, tf)
    finally:
        ctx.close()


def test_post_steps_and_morph_triangles_against_the_oracle_at_mid_size():
    """26 x 24 x 22 x 12 samples, two moving blobs + noise (277 k tetrahedra, 525 k morph triangles -- the reference's own run would take
    hours: the oracle's restatement stands in, pinned by the small fixtures above): bin_times / drop_instant / tiny collapse (B3),
    the slicing into morph triangles (B4: segments with direction, triangles as sets of segments) and the time-compatible windings
    (B5: every triangle the oracle's flood fill orients agrees) -- pentatopes.py:162-189, 314-368, morph_geometry.py:145-237"""
    from contourist_amd import pentatopes
    from oracle import level0_4d, postpass4d
    shape = (26, 24, 22, 12)
    ax = [np.linspace(0, 1, n, dtype=np.float32) for n in shape]
    X, Y, Z, T = np.meshgrid(*ax, indexing="ij")
    rng = np.random.RandomState(5)
    A = np.exp(-(((X - 0.35 - 0.3 * T) ** 2 + (Y - 0.45) ** 2 + (Z - 0.5 + 0.1 * T) ** 2) / (2 * 0.17 ** 2))) + \
        np.exp(-(((X - 0.72 + 0.2 * T) ** 2 + (Y - 0.6) ** 2 + (Z - 0.4) ** 2) / (2 * 0.12 ** 2)))
    A += 0.01 * rng.standard_normal(shape)
    for axis in range(4):
        for idx in (0, -1):
            np.moveaxis(A, axis, 0)[idx] = 0.0
    A = np.ascontiguousarray(A.astype(np.float32))
    v = 0.5
    corner = np.array(shape) - 1
    maker = pentatopes.GridContour4D(tuple(corner), A, v)
    R = maker.find_tetrahedra()
    MT = maker.collect_morph_triangles()
    kh = R["keys"].astype(np.int64)
    O = level0_4d.march4d(A, v, diag_mode=1)
    ko = level0_4d.edge_keys4(O["pairs"], shape)
    W = postpass4d.find_tetrahedra_post(ko, O["xyzt"], O["tets"], corner)
    assert R["counts"]["n_after_drop"] == W["n_after_drop"] and R["counts"]["n_after_tiny"] == W["n_after_tiny"] and W["n_after_tiny"] > 100000
    assert np.array_equal(R["points4d"][np.argsort(kh)], W["xyzt"][np.argsort(ko)])          # float64, bit for bit
    got = level0_4d.canonical4(kh, R["points4d"], R["tetrahedra"].astype(np.int64))[2]
    want = level0_4d.canonical4(ko, W["xyzt"], W["tets"])[2]
    assert np.array_equal(got, want)
    M = postpass4d.collect_morph_triangles(ko, W["xyzt"], W["tets"])
    seg_d = np.asarray(MT.segment_point_indices, dtype=np.int64)
    seg_o = np.asarray(M["segments"], dtype=np.int64)
    # segments with direction: pairs of edge ids
    sd = np.stack([kh[seg_d[:, 0]], kh[seg_d[:, 1]]], axis=1)
    so = np.stack([np.asarray(M["keys"])[seg_o[:, 0]], np.asarray(M["keys"])[seg_o[:, 1]]], axis=1)
    sd = sd[np.lexsort(sd.T[::-1])]; so = so[np.lexsort(so.T[::-1])]
    assert np.array_equal(sd, so)
    assert len(MT.triangle_segment_indices) == len(M["triangles"]) > 300000
    ot, label, flags = postpass4d.orient_morph_triangles(M)
    common, agree = postpass4d.winding_agreement(kh, MT.segment_point_indices, MT.triangle_segment_indices, M["keys"], M["segments"], ot)
    assert common == len(ot) and agree == common, (common, agree, len(ot))
    bad_d, seen_d = postpass4d.forced_pair_violations(kh, MT.segment_point_indices, MT.triangle_segment_indices, MT.points4d)
    assert seen_d > 0 and bad_d == 0, (bad_d, seen_d)


def _blob_field(shape, seed):
    rng = np.random.RandomState(seed)
    ax = [np.linspace(0.0, 1.0, n, dtype=np.float32) for n in shape]
    X, Y, Z, T = np.meshgrid(*ax, indexing="ij")
    A = np.exp(-(((X - 0.35 - 0.3 * T) ** 2 + (Y - 0.4) ** 2 + (Z - 0.5 + 0.1 * T) ** 2) / (2 * 0.16 ** 2)))
    A += 0.02 * rng.standard_normal(shape)
    for axis in range(4):
        for idx in (0, -1):
            np.moveaxis(A, axis, 0)[idx] = 0.0
    return np.ascontiguousarray(A.astype(np.float32))


@pytest.mark.parametrize("shape,nbins", [((14, 13, 12, 9), 100), ((22, 20, 18, 40), 100), ((12, 12, 12, 7), 1000)])
def test_per_t_stream_in_one_call_equals_the_single_surfaces(shape, nbins):
    """cx_morph_eval_many (config 4's per-t isosurface stream in ONE set of launches): every surface is bit for bit what
    cx_morph_eval returns for that time and what MorphTriangles.triangles_at computes on the host from the downloaded morph
    triangles (misc/morph_triangles.js:26-140) -- times unsorted, repeated, on vertex times exactly, outside the range;
    and the morph triangles / segments come back sorted by the bin of their start time (the windows the evaluation relies on)"""
    from contourist_amd import _ffi, morph_geometry
    A = _blob_field(shape, 5)
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid4d(A)
        ctx.extract4d(0.5, 1)
        ctx.postprocess4d(nbins)
        pts, segs, tris, ncomp = ctx.morph_triangles()
        assert len(tris) > 1000
        MT = morph_geometry.MorphTriangles(pts, segs, tris)
        tmin, tmax = float(pts[:, 3].min()), float(pts[:, 3].max())
        # sorted by start-time bin (256 bins over [tmin, tmax]); segments point from low t to high t
        s_lo, s_hi = pts[segs[:, 0], 3], pts[segs[:, 1], 3]
        assert np.all(s_lo <= s_hi)
        inv_width = 256.0 / (tmax - tmin)             # (the device's own formula: bins meet vertex times exactly at 1/4, 1/2, 3/4)
        sbin = np.clip(((s_lo - tmin) * inv_width).astype(np.int64), 0, 255)
        tbin = np.clip(((s_lo[tris].max(axis=1) - tmin) * inv_width).astype(np.int64), 0, 255)
        assert np.all(np.diff(sbin) >= 0) and np.all(np.diff(tbin) >= 0)
        vertex_times = np.unique(pts[:, 3])
        fr = np.array([0.52, 0.013, 0.97, 0.52, 0.31, 0.0, 1.0, 0.744, 0.25])
        times = list(tmin + fr * (tmax - tmin)) + [float(vertex_times[len(vertex_times) // 2]), float(vertex_times[1]),
                                                   tmin - 1.0, tmax + 0.5]
        many = ctx.morph_eval_many(times)
        counts = ctx.morph_eval_many(times, download=False)
        assert len(many) == len(times) and counts.shape == (len(times), 2)
        nonempty = 0
        for i, t in enumerate(times):
            p1, t1 = ctx.morph_eval(t)
            ph, th = MT.triangles_at(t)
            pm, tm = many[i]
            assert counts[i, 0] == len(pm) and counts[i, 1] == len(tm)
            assert np.array_equal(tm, t1) and np.array_equal(pm, p1)
            assert np.array_equal(tm, th) and np.allclose(pm, ph, rtol=0, atol=1e-12)
            if len(tm):
                assert tm.min() >= 0 and tm.max() < len(pm) and len(np.unique(tm)) == len(pm)
            nonempty += len(tm) > 0
        assert nonempty >= 9
        assert len(many[-1][1]) == 0 and len(many[-2][1]) == 0
        # surface by surface (cx_morph_eval_many_download) == the views of the one transfer (cx_morph_eval_many_download_all)
        ctx.morph_eval_many(times, download=False)
        for i in (0, 3, len(times) - 3, len(times) - 1):
            pi = np.empty((int(counts[i, 0]), 3), dtype=np.float64)
            ti = np.empty((int(counts[i, 1]), 3), dtype=np.int32)
            ctx._check(ctx.lib.cx_morph_eval_many_download(ctx.handle, i, pi.ctypes.data, ti.ctypes.data))
            assert np.array_equal(pi, many[i][0]) and np.array_equal(ti, many[i][1])
        # the stream again after the single calls, and an empty list of times
        again = ctx.morph_eval_many(times)
        for (pa, ta), (pm, tm) in zip(again, many):
            assert np.array_equal(ta, tm) and np.array_equal(pa, pm)
        assert ctx.morph_eval_many([]) == []
        # into arrays of the caller (kept from stream to stream)
        P = np.full((int(counts[:, 0].sum()) + 5, 3), -1.0)
        T = np.full((int(counts[:, 1].sum()) + 7, 3), -1, dtype=np.int32)
        mine = ctx.morph_eval_many(times, out=(P, T))
        for (pa, ta), (pm, tm) in zip(mine, many):
            assert np.array_equal(ta, tm) and np.array_equal(pa, pm) and (len(pa) == 0 or np.shares_memory(pa, P))
        assert np.all(P[-5:] == -1.0) and np.all(T[-7:] == -1)
        with pytest.raises(ValueError):
            ctx.morph_eval_many(times, out=(P[:3], T))
        # device addresses of a surface: what the download copies from
        torch = pytest.importorskip("torch")
        ctx.morph_eval_many(times, download=False)
        i = int(np.argmax(counts[:, 1]))
        dp, dt = ctx.morph_eval_device_ptrs(i)
        assert dp and dt
        import ctypes
        host = np.empty((int(counts[i, 1]), 3), dtype=np.int32)
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        assert hip.hipMemcpy(host.ctypes.data, dt, host.nbytes, 2) == 0
        assert np.array_equal(host, many[i][1])
        assert ctx.morph_eval_device_ptrs(len(times) - 1) == (0, 0)
    finally:
        ctx.close()


def test_per_t_stream_argument_errors():
    """cx_morph_eval_many: a NaN among the times, a negative count and a call before any morph triangles are refused with
    an error code and a message (no launch); the context keeps working afterwards"""
    from contourist_amd import _ffi
    ctx = _ffi.Context(0)
    try:
        out = np.zeros((4, 2), dtype=np.int64)
        ts = np.array([0.5, 1.5], dtype=np.float64)
        # nothing extracted yet: no post state at all
        assert ctx.lib.cx_morph_eval_many(ctx.handle, ts.ctypes.data, 2, out.ctypes.data) != 0
        A = _blob_field((12, 11, 10, 8), 9)
        ctx.upload_grid4d(A)
        ctx.extract4d(0.5, 1)
        ctx.postprocess4d(100)
        # post steps done, morph triangles not yet built: nothing to evaluate, empty surfaces
        assert ctx.lib.cx_morph_eval_many(ctx.handle, ts.ctypes.data, 2, out.ctypes.data) == 0 and not out.any()
        pts, segs, tris, _ = ctx.morph_triangles()
        assert ctx.lib.cx_morph_eval_many(ctx.handle, ts.ctypes.data, -1, out.ctypes.data) != 0
        bad = np.array([0.5, np.nan], dtype=np.float64)
        with pytest.raises(_ffi.CxError):
            ctx.morph_eval_many(bad)
        assert ctx.lib.cx_morph_eval_many_download(ctx.handle, 7, None, None) != 0
        t = 0.5 * (pts[:, 3].min() + pts[:, 3].max())
        p, q = ctx.morph_eval(t)
        assert len(q) > 0 and q.max() < len(p)
        # more times than the pinned staging of descriptors and totals holds (1 151): the pageable path
        many_t = np.linspace(pts[:, 3].min() - 0.1, pts[:, 3].max() + 0.1, 1500)
        counts = ctx.morph_eval_many(many_t, download=False)
        assert counts.shape == (1500, 2) and counts[:, 1].max() > 0 and counts[0, 1] == 0 and counts[-1, 1] == 0
        for i in (1, 377, 750, 1123, 1498):
            pi = np.empty((int(counts[i, 0]), 3), dtype=np.float64)
            ti = np.empty((int(counts[i, 1]), 3), dtype=np.int32)
            ctx._check(ctx.lib.cx_morph_eval_many_download(ctx.handle, i, pi.ctypes.data, ti.ctypes.data))
            p1, t1 = ctx.morph_eval(float(many_t[i]))
            assert np.array_equal(t1, ti) and np.array_equal(p1, pi)
            ctx.morph_eval_many(many_t, download=False)      # (the single call replaced the stream's surfaces)
        # a new post-pass (here: of a smaller field) takes the morph triangles of the old one away: nothing to evaluate until they are rebuilt
        ctx.upload_grid4d(_blob_field((8, 8, 8, 6), 2))
        ctx.extract4d(0.5, 1)
        ctx.postprocess4d(100)
        p, q = ctx.morph_eval(t)
        assert len(p) == 0 and len(q) == 0
        pts2, segs2, tris2, _ = ctx.morph_triangles()
        p, q = ctx.morph_eval(0.5 * (pts2[:, 3].min() + pts2[:, 3].max()))
        assert len(q) > 0 and q.max() < len(p)
    finally:
        ctx.close()


def test_per_t_stream_at_mid_size_equals_the_host_evaluation():
    """64 x 64 x 64 x 32 (two moving blobs, ~1.5 M morph triangles): the per-t stream of 40 times in one call -- windows of many blocks,
    offsets across blocks and across times -- against MorphTriangles.triangles_at on the host for every time, and the single calls"""
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi, morph_geometry
    dev = torch.device("cuda", 0)
    shape = (64, 64, 64, 32)
    ax = [torch.arange(n, device=dev, dtype=torch.float32) / (n - 1) for n in shape]
    X, Y, Z, T = torch.meshgrid(*ax, indexing="ij")
    A = torch.exp(-(((X - 0.30 - 0.35 * T) ** 2 + (Y - 0.35 - 0.2 * T) ** 2 + (Z - 0.5) ** 2) / (2 * 0.12 ** 2))) + \
        torch.exp(-(((X - 0.70 + 0.30 * T) ** 2 + (Y - 0.65 + 0.2 * T) ** 2 + (Z - 0.45 - 0.1 * T) ** 2) / (2 * 0.10 ** 2)))
    for axis in range(4):
        for idx in (0, 1, -1, -2):
            A.select(axis, idx).fill_(0.0)
    A = A.contiguous()
    ctx = _ffi.Context(0)
    try:
        ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
        ctx.extract4d(0.5, 1)
        ctx.postprocess4d(100)
        pts, segs, tris, _ = ctx.morph_triangles()
        assert len(tris) > 500000
        MT = morph_geometry.MorphTriangles(pts, segs, tris)
        tmin, tmax = float(pts[:, 3].min()), float(pts[:, 3].max())
        rng = np.random.RandomState(3)
        times = np.concatenate([np.linspace(tmin, tmax, 32), rng.uniform(tmin, tmax, size=6), [tmin + 0.4 * (tmax - tmin)] * 2])
        rng.shuffle(times)
        many = ctx.morph_eval_many(times)
        total = 0
        for i, t in enumerate(times):
            ph, th = MT.triangles_at(float(t))
            pm, tm = many[i]
            assert np.array_equal(tm, th) and np.allclose(pm, ph, rtol=0, atol=1e-12), (i, t, len(tm), len(th))
            total += len(tm)
        assert total > 1000000
        for i in (0, 7, 21):
            p1, t1 = ctx.morph_eval(float(times[i]))
            assert np.array_equal(t1, many[i][1]) and np.array_equal(p1, many[i][0])
    finally:
        ctx.close()


def test_extract4d_async_equals_the_synchronous_march():
    """cx_extract4d_async + cx_counts4d_get: two contexts with one extraction in flight each (different isovalues of two fields), the
    same counts, vertices and tetrahedra as cx_extract4d, three times over; cx_counts4d_get without an extraction in flight is refused"""
    from contourist_amd import _ffi
    from oracle import level0_4d
    fields = [(_blob_field((18, 16, 14, 10), 4), 0.5), (_blob_field((12, 13, 14, 9), 6), 0.42)]

    def canon(ctx, counts, shape):
        xyzt, keys, tets = ctx.download_level0_4d(counts)
        return level0_4d.canonical4(keys.astype(np.int64), xyzt, tets.astype(np.int64))
    want = []
    for A, v in fields:
        c = _ffi.Context(0)
        try:
            c.upload_grid4d(A)
            counts = c.extract4d(v, 1)
            want.append((counts, canon(c, counts, A.shape)))
        finally:
            c.close()
    ctxs = [_ffi.Context(0), _ffi.Context(0)]
    try:
        with pytest.raises(_ffi.CxError):
            ctxs[0].counts4d()
        for rep in range(3):
            for c, (A, v) in zip(ctxs, fields):
                if rep == 0:
                    c.upload_grid4d(A)
                c.extract4d_async(v, 1)
            for c, (A, v), (counts0, canon0) in zip(ctxs, fields, want):
                counts = c.counts4d()
                assert counts == counts0
                got = canon(c, counts, A.shape)
                assert np.array_equal(got[0], canon0[0]) and np.array_equal(got[2], canon0[2]) and np.array_equal(got[1], canon0[1])
        with pytest.raises(_ffi.CxError):
            ctxs[1].counts4d()
    finally:
        for c in ctxs:
            c.close()


def test_extract4d_grows_its_buffers_sync_and_async():
    """white noise at its median: nearly every hyper-voxel is active, far more than a fresh context reserves (one cell in eight):
    cx_extract4d repeats the march with larger buffers, and so does cx_counts4d_get after cx_extract4d_async -- both give the
    oracle's mesh"""
    from contourist_amd import _ffi
    from oracle import level0_4d
    rng = np.random.RandomState(11)
    A = np.ascontiguousarray(rng.standard_normal((20, 18, 16, 14)).astype(np.float32))
    O = level0_4d.march4d(A, 0.0, diag_mode=1)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    want = level0_4d.canonical4(ko, O["xyzt"], O["tets"])
    # beyond the first reservation (cells: a sample in eight + 4096; tetrahedra: 4 per sample + 4096)
    assert O["nborder_mixed"] > A.size // 8 + 4096 and len(O["tets"]) > 4 * A.size + 4096
    for use_async in (False, True):
        ctx = _ffi.Context(0)
        try:
            ctx.upload_grid4d(A)
            if use_async:
                ctx.extract4d_async(0.0, 1)
                counts = ctx.counts4d()
            else:
                counts = ctx.extract4d(0.0, 1)
            assert counts["n_tetrahedra"] == len(O["tets"]) and counts["n_vertices"] == len(ko)
            xyzt, keys, tets = ctx.download_level0_4d(counts)
            got = level0_4d.canonical4(keys.astype(np.int64), xyzt, tets.astype(np.int64))
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
            assert np.all(np.abs(got[1] - want[1]) <= 1e-6 * np.abs(want[1]) + 1e-6)
        finally:
            ctx.close()
