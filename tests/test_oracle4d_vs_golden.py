"""CPU: the 4-D oracle (oracle/march4d_oracle.c) against vectors produced by the real reference
(oracle/make_goldens4d.py).  The reference has no test of its own for the pentatope path, so these
goldens are the only pin."""
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import level0_4d

G4 = os.path.join(ROOT, "tests", "golden4d")


def names():
    # (fixtures named *seeded* were made with explicit end points: the reference reached a subset; own tests below)
    return sorted(f[:-4] for f in os.listdir(G4) if f.endswith(".npz") and "seeded" not in f) if os.path.isdir(G4) else []


@pytest.mark.parametrize("name", names())
def test_level0_4d_exact(name):
    G = np.load(os.path.join(G4, name + ".npz"))
    A, v = G["A"], float(G["value"])
    O = level0_4d.march4d(A, v, diag_mode=1)
    assert O["nborder"] == len(G["surface_voxels"])
    kr = level0_4d.edge_keys4(G["l0_pairs"], A.shape)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    cr = level0_4d.canonical4(kr, G["l0_xyzt"], G["l0_tets"])
    co = level0_4d.canonical4(ko, O["xyzt"], O["tets"])
    assert np.array_equal(cr[0], co[0])                      # crossing edges
    assert np.array_equal(G["l0_pairs"][np.argsort(kr)], O["pairs"][np.argsort(ko)])   # low -> high orientation
    assert np.array_equal(cr[1], co[1])                      # float64 coordinates, bit for bit
    assert np.array_equal(cr[2], co[2])                      # tetrahedra incl. the hash-order 2-3 splits
    O0 = level0_4d.march4d(A, v, diag_mode=0)
    c0 = level0_4d.canonical4(level0_4d.edge_keys4(O0["pairs"], A.shape), O0["xyzt"], O0["tets"])
    assert np.array_equal(level0_4d.pentatope_groups(cr[2], A.shape), level0_4d.pentatope_groups(c0[2], A.shape))


@pytest.mark.parametrize("name", names())
def test_b3_post_steps(name):
    """bin_times / drop_instant_tetrahedra / remove_tiny_simplices against the reference's own results"""
    from oracle import postpass4d
    G = np.load(os.path.join(G4, name + ".npz"))
    A, v = G["A"], float(G["value"])
    corner = np.array(A.shape) - 1
    O = level0_4d.march4d(A, v, diag_mode=1)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    R = postpass4d.find_tetrahedra_post(ko, O["xyzt"], O["tets"], corner)
    kr = level0_4d.edge_keys4(G["l0_pairs"], A.shape)
    assert np.array_equal(R["xyzt_binned"][np.argsort(ko)], G["b3_xyzt_binned"][np.argsort(kr)])   # bit for bit
    assert R["n_after_drop"] == int(G["n_tets_after_drop"])
    assert R["n_after_tiny"] == int(G["n_tets_after_tiny"])


@pytest.mark.parametrize("name", names())
def test_morph_triangles_b4_b5(name):
    """collect_morph_triangles + orient_triangles: segments (with direction), slice polygons and points equal the
    reference's; windings agree on every triangle both produce (the split of 4-segment slices follows the
    reference's dict numbering and is not contractual)."""
    from oracle import postpass4d
    G = np.load(os.path.join(G4, name + ".npz"))
    A, v = G["A"], float(G["value"])
    corner = np.array(A.shape) - 1
    O = level0_4d.march4d(A, v, diag_mode=1)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    W = postpass4d.find_tetrahedra_post(ko, O["xyzt"], O["tets"], corner)
    M = postpass4d.collect_morph_triangles(ko, W["xyzt"], W["tets"])
    rk = level0_4d.edge_keys4(G["mt_point_pairs"], A.shape)
    assert np.array_equal(G["mt_points4d"][np.argsort(rk)], M["points4d"])
    ref_seg = set((int(rk[i]), int(rk[j])) for i, j in G["mt_segments"])
    got_seg = set((int(M["keys"][i]), int(M["keys"][j])) for i, j in M["segments"])
    assert ref_seg == got_seg                                           # same segments, same low-t -> high-t direction
    assert len(G["mt_triangles"]) == len(M["triangles"])
    assert postpass4d.morph_polygons(rk, G["mt_segments"], G["mt_triangles"]) == \
        postpass4d.morph_polygons(M["keys"], M["segments"], M["triangles"])
    ot, label, flags = postpass4d.orient_morph_triangles(M)
    common, agree = postpass4d.winding_agreement(rk, G["mt_segments"], G["mt_triangles"], M["keys"], M["segments"], ot)
    assert common > 0.7 * len(ot) and agree == common


def test_seeded_growth_4d_vs_reference():
    """GridContour4D(corner, f, value, end points) of the reference (its own calling convention, pentatopes.py:528-551):
    the restated search (oracle/seeds.py, 80 neighbours) selects exactly the hyper-voxels and tetrahedra it reached"""
    from oracle import seeds
    G = np.load(os.path.join(G4, "two_blobs_seeded_12x12x12x7.npz"))
    A, v = G["A"], float(G["value"])
    O = level0_4d.march4d(A, v, diag_mode=1)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    keep, surf = seeds.select4d(A, v, G["end_points"], ko, O["tets"])
    assert 0 < keep.sum() < len(keep)                                    # the other blob is left out
    assert sorted(surf) == sorted(tuple(int(x) for x in q) for q in G["surface_voxels"])
    kr = level0_4d.edge_keys4(G["l0_pairs"], A.shape)
    want = np.sort(kr[G["l0_tets"]], axis=1)
    got = np.sort(ko[O["tets"][keep]], axis=1)
    assert np.array_equal(want[np.lexsort(want.T[::-1])], got[np.lexsort(got.T[::-1])])


def test_reference_test0_call():
    """the reference's own 4-D demo call (pentatopes.py:528-551: GridContour4D([8]*4, function, 2.0, endpoints), the
    callable evaluated in float64, two start voxels outside the grid): on fp32 samples of the same field the restated
    search reaches the same hyper-voxels inside the grid and the march gives the same tetrahedra there"""
    from oracle import seeds
    from oracle.make_goldens4d import test0_field
    G = np.load(os.path.join(G4, "reference_test0_seeded.npz"))
    g = np.arange(9, dtype=np.float64)
    X, Y, Z, T = np.meshgrid(g, g, g, g, indexing="ij")
    A = test0_field(X, Y, Z, T).astype(np.float32)
    v = float(G["value"])
    O = level0_4d.march4d(A, v, diag_mode=1)
    ko = level0_4d.edge_keys4(O["pairs"], A.shape)
    keep, surf = seeds.select4d(A, v, G["end_points"], ko, O["tets"])
    ref_inside = set(tuple(int(x) for x in q) for q in G["surface_voxels"] if q.min() >= 0 and q.max() < 8)
    assert len(G["surface_voxels"]) - len(ref_inside) == 2            # (0,0,0,-1) and (3,3,3,8): seeds outside the grid
    assert surf == ref_inside
    P = G["l0_pairs"]
    inside = (P.min(axis=1) >= 0) & (P.max(axis=1) <= 8)
    tets_inside = inside[G["l0_tets"]].all(axis=1)
    kr = np.full(len(P), -1, dtype=np.int64)
    kr[inside] = level0_4d.edge_keys4(P[inside], A.shape)
    want = np.sort(kr[G["l0_tets"][tets_inside]], axis=1)
    got = np.sort(ko[O["tets"][keep]], axis=1)
    a = set(map(tuple, want.tolist()))
    b = set(map(tuple, got.tolist()))
    assert a == b and len(a) == 26004           # every tetrahedron inside the grid; the other 96 sit in the two outside voxels
