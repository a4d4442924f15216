"""GPU: seeded selection of surface components (tetrahedral.py:396-463) -- the reference's own unit test run
verbatim through the mirrored API, and device == oracle on multi-component fields."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def two_dots(x, y, z):
    if x == y == z == -8 or x == y == z == 0:
        return 1
    return -1


def test_reference_unit_test_verbatim():
    """contourist/test/test_tetrahedral.py:13-37, only the import changed"""
    from contourist_amd import tetrahedral
    f = two_dots
    mins = [-8] * 3
    maxes = [8] * 3
    deltas = [2] * 3
    eps = [[(-8, -8, -8), (-8, -8, 8)]]
    S = tetrahedral.TriangulatedIsosurfaces(mins, maxes, deltas, f, 0, eps)
    (points, triangles) = S.get_points_and_triangles()
    points = [tuple(int(i) for i in pt) for pt in points]
    triangle_vertices = set(frozenset(points[i] for i in triangle) for triangle in triangles)
    expected = set([frozenset([(-9, -9, -8), (-9, -8, -8), (-8, -8, -7)]),
                    frozenset([(-7, -8, -8), (-7, -8, -7), (-7, -7, -7)]),
                    frozenset([(-8, -8, -7), (-8, -7, -7), (-7, -7, -7)]),
                    frozenset([(-8, -8, -7), (-7, -8, -7), (-7, -7, -7)]),
                    frozenset([(-9, -9, -8), (-8, -9, -8), (-8, -8, -7)]),
                    frozenset([(-8, -7, -8), (-7, -7, -8), (-7, -7, -7)]),
                    frozenset([(-7, -8, -8), (-7, -7, -8), (-7, -7, -7)]),
                    frozenset([(-8, -7, -8), (-8, -7, -7), (-7, -7, -7)])])
    assert triangle_vertices == expected
    # and the golden written by the real reference for the same call
    G = np.load(os.path.join(GOLDEN_DIR, "two_dots.npz"))
    ref = set(frozenset(tuple(int(x) for x in G["l1_points"][i]) for i in t) for t in G["l1_triangles"])
    assert triangle_vertices == ref


def level0_on_device(A, v):
    from contourist_amd import _ffi
    ctx = _ffi.Context(0)
    ctx.upload_grid(A)
    counts = ctx.extract3d(v, _ffi.CX_DIAG_CPYTHON310)
    xyz, keys, tris = ctx.download_level0(counts)
    return ctx, counts, xyz, keys.astype(np.int64), tris.astype(np.int64)


@pytest.mark.parametrize("name,seed_sets", [("blobs27", 2), ("shells24", 2), ("noise24_v07", 3)])
def test_device_selection_equals_oracle(name, seed_sets):
    """seeds taken from crossing edges of individual components: the device keeps exactly the triangles the
    restated reference search keeps (fields without samples equal to the isovalue)"""
    from oracle import level0, seeds
    G = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    A, v = G["A"], float(G["value"])
    ctx, counts, xyz, keys, tris = level0_on_device(A, v)
    try:
        lin, d = keys >> 3, keys & 7
        n1n2 = A.shape[1] * A.shape[2]
        q = np.stack([lin // n1n2, (lin // A.shape[2]) % A.shape[1], lin % A.shape[2]], axis=1)
        dv = np.stack([(d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1)
        rng = np.random.RandomState(7)
        for trial in range(seed_sets):
            pick = rng.choice(len(keys), size=1 + trial, replace=False)
            eps = [[tuple(int(x) for x in q[p]), tuple(int(x) for x in q[p] + dv[p])] for p in pick]
            O = level0.march3d(A, v, diag_mode=1)
            ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
            want, _ = seeds.select(A, v, eps, ko, O["tris"])
            got = ctx.select_seeded(eps)
            assert got["triangles_kept"] == int(want.sum())
            # Level 1 of the selection == the oracle's Level 1 of the filtered Level-0 mesh (vertices of dropped
            # components do not exist for the weld)
            from oracle import postpass
            corner = np.array(A.shape) - 1
            used = np.zeros(len(ko), dtype=bool)
            used[O["tris"][want].ravel()] = True
            renum = np.cumsum(used) - 1
            L1 = postpass.level1_from_level0(ko[used], O["xyz"][used], renum[O["tris"][want]], corner)
            post = ctx.postprocess3d(0)
            pts, t1 = ctx.download_level1(post)
            assert post["n_after_weld"] == L1["n_after_weld"] and post["n_after_tiny"] == L1["n_after_tiny"]
            assert len(t1) == len(L1["triangles"])
            cmp = postpass.compare_level1(L1, pts, t1, corner, reach=0)
            assert not cmp["missing"] and not cmp["extra"] and not cmp["winding"]
    finally:
        ctx.close()


def test_bad_endpoints_are_rejected():
    from contourist_amd import _ffi
    G = np.load(os.path.join(GOLDEN_DIR, "sphere32.npz"))
    ctx, counts, xyz, keys, tris = level0_on_device(G["A"], float(G["value"]))
    try:
        with pytest.raises(_ffi.CxError):
            ctx.select_seeded([[(0, 0, 0), (0, 0, 1)]])          # both outside the sphere: do not straddle the isovalue
    finally:
        ctx.close()


def test_search_for_endpoints_with_skip():
    """skip > 1: the coarse crossing search seeds the growth; a component that the coarse lattice misses is not
    returned (the reference's sparsity mode, grid_field.py:64-84 + tetrahedral.py:396-463), a large one is."""
    from contourist_amd import tetrahedral
    n = 48
    ax = np.arange(n, dtype=np.float64)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    big = np.sqrt((X - 16.3) ** 2 + (Y - 16.1) ** 2 + (Z - 16.2) ** 2) - 9.0       # sphere of radius 9
    small = np.sqrt((X - 38.5) ** 2 + (Y - 38.5) ** 2 + (Z - 38.5) ** 2) - 1.2     # sphere of radius 1.2 between coarse points
    A = np.minimum(big, small).astype(np.float32)
    full = tetrahedral.TriangulatedIsosurfaces([0] * 3, None, [1] * 3, A, 0.0, [])
    full.search_for_endpoints()
    p_all, t_all = full.get_points_and_triangles()
    coarse = tetrahedral.TriangulatedIsosurfaces([0] * 3, None, [1] * 3, A, 0.0, [])
    coarse.search_for_endpoints(skip=8)
    p_big, t_big = coarse.get_points_and_triangles()
    assert 0 < len(t_big) < len(t_all)
    assert np.all(np.linalg.norm(np.asarray(p_big) - np.array([16.3, 16.1, 16.2]), axis=1) < 10.5)   # only the big sphere
    near_small = np.linalg.norm(np.asarray(p_all) - 38.5, axis=1) < 3
    assert near_small.any()                                                            # the exhaustive search has both
    # closed surface: Euler characteristic 2
    assert len(p_big) - len(t_big) // 2 == 2


def test_many_seeds_take_the_parallel_path():
    """the one-thread-per-pair seed kernel (what more than 65 536 end point pairs get; forced here): same selection
    as the oracle on a field where every candidate voxel of a pair belongs to one component"""
    from oracle import level0, seeds
    G = np.load(os.path.join(GOLDEN_DIR, "blobs27.npz"))
    A, v = G["A"], float(G["value"])
    ctx, counts, xyz, keys, tris = level0_on_device(A, v)
    try:
        O = level0.march3d(A, v, diag_mode=1)
        ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
        lin, d = keys >> 3, keys & 7
        n1n2 = A.shape[1] * A.shape[2]
        q = np.stack([lin // n1n2, (lin // A.shape[2]) % A.shape[1], lin % A.shape[2]], axis=1)
        dv = np.stack([(d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1)
        # all crossing edges of the component that contains vertex 0, repeated to exceed 1024 pairs
        mask0, surf = seeds.select(A, v, [[tuple(q[0]), tuple(q[0] + dv[0])]], ko, O["tris"])
        vox = seeds.triangle_voxels(ko, O["tris"], A.shape)
        comp_vertices = np.unique(np.vectorize({int(k): n for n, k in enumerate(keys)}.get)(ko[np.unique(O["tris"][mask0])]))
        eps = [[tuple(int(x) for x in q[p]), tuple(int(x) for x in q[p] + dv[p])] for p in comp_vertices]
        while len(eps) <= 1024:
            eps = eps + eps
        got = ctx.select_seeded(eps, parallel=True)
        assert got["triangles_kept"] == int(mask0.sum())
    finally:
        ctx.close()
