"""CPU: the oracle (oracle/march_oracle.c + oracle/postpass.py) against vectors produced by the
real reference (oracle/make_goldens.py) and against the reference's own golden triangle set
(contourist/test/test_tetrahedral.py:29-36)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_names
from oracle import level0, postpass


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.mark.parametrize("name", golden_names())
def test_level0_exact(name):
    G = load(name)
    A, v = G["A"], float(G["value"])
    shape = A.shape
    O = level0.march3d(A, v, diag_mode=1)
    assert O["nborder"] == len(G["surface_voxels"])           # dense scan == the reference's BFS here
    kr = level0.edge_keys_from_pairs(G["l0_pairs"], shape)
    ko = level0.edge_keys_from_pairs(O["pairs"], shape)
    cr = level0.canonical_level0(kr, G["l0_xyz"], G["l0_tris"])
    co = level0.canonical_level0(ko, O["xyz"], O["tris"])
    assert np.array_equal(cr[0], co[0])                        # same set of crossing edges
    assert np.array_equal(G["l0_pairs"][np.argsort(kr)], O["pairs"][np.argsort(ko)])   # same low->high orientation
    assert np.array_equal(cr[1], co[1])                        # float64 interpolation, bit for bit
    assert np.array_equal(cr[2], co[2])                        # same triangles incl. quad diagonals
    # canonical-diagonal mode differs only in the diagonals
    O0 = level0.march3d(A, v, diag_mode=0)
    c0 = level0.canonical_level0(level0.edge_keys_from_pairs(O0["pairs"], shape), O0["xyz"], O0["tris"])
    assert np.array_equal(level0.tet_polygons(cr[2], shape), level0.tet_polygons(c0[2], shape))


@pytest.mark.parametrize("name", golden_names())
def test_level1_canonical(name):
    G = load(name)
    A, v = G["A"], float(G["value"])
    corner = np.array(A.shape) - 1
    O = level0.march3d(A, v, diag_mode=1)
    keys = level0.edge_keys_from_pairs(O["pairs"], A.shape)
    smooth = float(G["smooth"]) if "smooth" in G.files else None
    postpass.set_compare_scale(1e8 if smooth else None)
    L1 = postpass.level1_from_level0(keys, O["xyz"], O["tris"], corner, smooth=smooth)
    assert L1["n_after_weld"] == int(G["n_tris_after_weld"])   # invariant through the weld
    band = G["l1_count_band"]
    n_tiny = L1["n_after_weld"] - L1["n_after_tiny"]
    nsites = len(L1["sites"])
    # smoothing spreads a welded group's representative choice over its 1-ring: excuse ~2 voxels around sites
    reach = 2 * int(postpass.expander_for(corner).max()) if smooth else 2
    cmp = postpass.compare_level1(L1, G["l1_grid_points"], G["l1_triangles"], corner, reach=reach)
    # same triangles (as weld-bucket triples) and same winding, except where the reference's own
    # hash order decides (weld representative, tiny-collapse merge point, ambiguous orientation)
    assert not cmp["missing"] and not cmp["extra"], (cmp["missing"][:2], cmp["extra"][:2])
    assert not cmp["winding"], cmp["winding"][:2]
    assert cmp["excused_rows"] <= (200 if smooth else 20) * max(nsites, 1)
    if nsites == 0:
        assert L1["n_after_tiny"] == int(G["n_tris_after_tiny"])
        assert cmp["n_oracle"] == cmp["n_other"] == len(G["l1_triangles"])
        assert cmp["excused_rows"] == 0
    else:
        assert abs(L1["n_after_tiny"] - int(G["n_tris_after_tiny"])) <= max(n_tiny, 2)
        assert abs(len(L1["triangles"]) - len(G["l1_triangles"])) <= max(2 * n_tiny, band[1] - band[0], 4)
    if name in ("sphere32", "inv_sphere20", "shells24", "blobs27", "noise24_v0", "noise32_v0"):
        # well-behaved closed surfaces: nothing about the winding may need excusing
        assert cmp["excused_winding"] == 0


def test_two_dots_reference_golden():
    """the 8 triangles the reference's own unit test expects appear in the dense march (the
    reference finds only these 8 because its BFS never reaches the rest)."""
    G = load("two_dots")
    A, v = G["A"], float(G["value"])
    O = level0.march3d(A, v, diag_mode=1)
    world = O["xyz"] * G["delta"] + G["mins"]
    ipts = np.trunc(world).astype(int)          # int(x) truncation as in the reference test
    got = set(frozenset(tuple(ipts[i]) for i in t) for t in O["tris"])
    expected = set(frozenset(tuple(p) for p in tri) for tri in G["expected_int_triangles"].reshape(-1, 3, 3))
    assert len(expected) == 8
    assert expected <= got


def test_two_dots_seeded_selection_is_the_reference_test_exactly():
    """contourist/test/test_tetrahedral.py:13-37 with its end points: the seeded search (find_initial_voxels +
    expand_voxels, restated in oracle/seeds.py) keeps exactly the 8 triangles the reference's test asserts"""
    from oracle import seeds
    G = load("two_dots")
    A, v = G["A"], float(G["value"])
    O = level0.march3d(A, v, diag_mode=1)
    keys = level0.edge_keys_from_pairs(O["pairs"], A.shape)
    # the test's end points (-8,-8,-8) -> (-8,-8,8) are lattice points (0,0,0) -> (0,0,8) of the reference's grid =
    # (1,1,1) -> (1,1,9) of the golden array, which carries one lattice step of margin (indices -1..9)
    mask, surface = seeds.select(A, v, [[(1, 1, 1), (1, 1, 9)]], keys, O["tris"], lo=(1, 1, 1), hi=(9, 9, 9))
    world = O["xyz"] * G["delta"] + G["mins"]
    ipts = np.trunc(world).astype(int)
    got = set(frozenset(tuple(ipts[i]) for i in t) for t in O["tris"][mask])
    expected = set(frozenset(tuple(p) for p in tri) for tri in G["expected_int_triangles"].reshape(-1, 3, 3))
    assert got == expected and len(got) == 8
    assert surface == {(0, 0, 1), (1, 1, 1)}      # voxel (-1,-1,0) (not range-checked seed) and voxel (0,0,0)


def test_set_order_emulation_matches_this_cpython():
    """tuple hash + 8-slot set order; checked against the running interpreter
    (only meaningful on CPython 3.8+ / 64-bit, which is what produced the goldens)."""
    import ctypes
    import sys
    if sys.implementation.name != "cpython" or sys.maxsize < 2 ** 62:
        pytest.skip("needs 64-bit CPython")
    rng = np.random.RandomState(0)
    L = level0.lib()
    for _ in range(2000):
        n = int(rng.randint(2, 4))
        pts = [tuple(int(x) for x in rng.randint(0, 600, size=3)) for _ in range(n)]
        if len(set(pts)) < n:
            continue
        hs = np.array([hash(p) & (2 ** 64 - 1) for p in pts], dtype=np.uint64)
        for p, h in zip(pts, hs):
            arr = np.array(p, dtype=np.int64)
            assert L.oracle_py_tuplehash(arr.ctypes.data, 3) == int(h)
        slots = np.zeros(n, dtype=np.int32)
        L.oracle_py_set8_slots(hs.ctypes.data, n, slots.ctypes.data)
        s = set()
        for p in pts:
            s.add(p)
        assert [pts[i] for i in np.argsort(slots)] == list(s)


def test_coarse_regime_oracle_vs_reference():
    """corner = 511 (weld buckets of 1/19 voxel): the oracle's post-pass reproduces the reference's weld exactly; its
    canonical tiny collapse lands inside the reference's own order-variation band, its clean-up at most 0.5 % below"""
    import os
    from oracle import level0, postpass
    G = np.load(os.path.join(GOLDEN_DIR, "coarse_sphere_r12_corner511.npz"))
    c = int(G["corner"])
    corner = np.array([c] * 3)
    assert int(postpass.expander_for(corner)[0]) == 19
    keys = level0.edge_keys_from_pairs(G["l0_pairs"], (c + 1,) * 3)
    L1 = postpass.level1_from_level0(keys, G["l0_xyz"], G["l0_tris"], corner)
    sc = G["stage_counts"]
    assert len(G["l0_tris"]) == sc[0, 0] and len(set(sc[:, 0].tolist())) == 1
    assert L1["n_after_weld"] == sc[0, 1] and len(set(sc[:, 1].tolist())) == 1
    assert sc[:, 2].min() <= L1["n_after_tiny"] <= sc[:, 2].max()
    assert sc[:, 3].min() * 0.995 <= len(L1["triangles"]) <= sc[:, 3].max()
    assert len(set(sc[:, 3].tolist())) > 1      # the reference itself does not agree with itself here
