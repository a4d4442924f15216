"""GPU: several isovalues of one grid in ONE call (cx_extract3d_levels, BASELINE config 5).

Every level must be (i) bit for bit the mesh cx_extract3d gives for that isovalue -- same vertex records, same index triples,
same numbering; (ii) the oracle's Level-0 mesh (edge set, float coordinates within 1e-6, triangles with the CPython-order
diagonals); (iii) at Level 1, the oracle's post-pass of the oracle's Level 0 (`compare_level1`, as for single levels) and,
for the fixture's own isovalue, the real reference's triangle count."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def single(A, v, flags):
    from contourist_amd import _ffi
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        c = ctx.extract3d(v, flags)
        return (c,) + ctx.download_level0(c)
    finally:
        ctx.close()


@pytest.mark.parametrize("name,values", [("noise32_v0", [0.4, -0.5, 0.0]), ("noise24_v07", [0.7, 0.1, -0.3, 0.45]),
                                         ("shells24", None), ("noise_14x20x26", [0.2, -0.2])])
@pytest.mark.parametrize("diag", [1, 0])
def test_levels_equal_single_extractions_and_oracle(name, values, diag):
    from contourist_amd import _ffi
    from oracle import level0, postpass
    G = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    A = G["A"]
    if values is None:
        values = [float(G["value"]), float(np.percentile(A, 30)), float(np.percentile(A, 80))]
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        counts = ctx.extract3d_levels(values, diag)
        assert len(counts) == len(values)
        for n in (1, 0, len(values) - 1, 0):          # any order, any number of times
            ctx.select_level(n)
            xyz, keys, tris = ctx.download_level0(counts[n])
            c1, x1, k1, t1 = single(A, values[n], diag)
            assert counts[n] == c1
            assert np.array_equal(keys, k1) and np.array_equal(tris, t1) and np.array_equal(xyz.view(np.uint32), x1.view(np.uint32))
        for n, v in enumerate(values):
            ctx.select_level(n)
            xyz, keys, tris = ctx.download_level0(counts[n])
            O = level0.march3d(A, v, diag_mode=diag)
            ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
            co = level0.canonical_level0(ko, O["xyz"], O["tris"])
            ch = level0.canonical_level0(keys.astype(np.int64), xyz, tris.astype(np.int64))
            assert np.array_equal(co[0], ch[0]) and np.array_equal(co[2], ch[2])
            assert np.all(np.abs(ch[1] - co[1]) <= 1e-6 * np.abs(co[1]) + 1e-6)
            if diag == 1 and len(ko):
                corner = np.array(A.shape) - 1
                L1 = postpass.level1_from_level0(ko, O["xyz"], O["tris"], corner)
                post = ctx.postprocess3d(0)
                pts, t1 = ctx.download_level1(post)
                assert post["n_after_weld"] == L1["n_after_weld"] and post["n_after_tiny"] == L1["n_after_tiny"]
                cmp = postpass.compare_level1(L1, pts, t1, corner, reach=0)
                assert not cmp["missing"] and not cmp["extra"] and not cmp["winding"]
    finally:
        ctx.close()


def test_multi_level_class_against_reference_golden():
    """MultiLevelIsosurfaces (the mirrored-API entry): ascending order, every level == TriangulatedIsosurfaces of that value,
    and the fixture's own level has the real reference's Level-1 triangle count (fine regime: identical)"""
    from contourist_amd import tetrahedral
    G = np.load(os.path.join(GOLDEN_DIR, "sphere32.npz"))
    A = G["A"]
    v0 = float(G["value"])
    values = [v0 * 1.3, v0, v0 * 0.6]
    M = tetrahedral.MultiLevelIsosurfaces(G["mins"], None, G["delta"], A, values)
    out = list(M.levels())
    assert [o[0] for o in out] == sorted(values)
    for v, pts, tris in out:
        S = tetrahedral.TriangulatedIsosurfaces(G["mins"], None, G["delta"], A, v, [])
        S.search_for_endpoints()
        p2, t2 = S.get_points_and_triangles()
        assert np.array_equal(np.asarray(pts), np.asarray(p2)) and np.array_equal(np.asarray(tris), np.asarray(t2))
        if v == v0:
            assert len(tris) == len(G["l1_triangles"]) and len(pts) == len(G["l1_points"])


def test_levels_then_single_extraction_and_reuse():
    """a single extraction after a multi-level call invalidates the levels; the context stays usable for both"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(2)
    g = np.linspace(-1, 1, 40)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    A = (np.sin(3 * X) * np.cos(2 * Y) + 0.7 * np.sin(4 * Z) + 0.03 * rng.standard_normal(X.shape)).astype(np.float32)
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        c = ctx.extract3d_levels([0.1, 0.3], 1)
        ctx.select_level(1)
        a = ctx.download_level0(c[1])
        c1 = ctx.extract3d(0.3, 1)
        b = ctx.download_level0(c1)
        assert c1 == c[1] and all(np.array_equal(x, y) for x, y in zip(a, b))
        with pytest.raises(_ffi.CxError):
            ctx.select_level(0)
        c = ctx.extract3d_levels([0.3, 0.1, -0.2, 0.5, 0.0], 1)
        ctx.select_level(0)
        a2 = ctx.download_level0(c[0])
        assert all(np.array_equal(x, y) for x, y in zip(a2, b))
    finally:
        ctx.close()


def test_levels_pool_overflow_falls_back_to_full_regions():
    """the levels share one pool of queue entries, a slice of every streaming wave's region each; white noise crosses nearly every
    voxel at every level, so the slices overflow, the flag comes back with the counts and the call repeats itself with full-size
    regions -- the meshes must be the single extractions' bit for bit, and a smooth field afterwards goes through the pool again"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(123)
    A = rng.standard_normal((40, 36, 64)).astype(np.float32)
    values = [-0.4, 0.0, 0.5]
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        counts = ctx.extract3d_levels(values, 1)
        assert sum(c["n_cells"] for c in counts) > 2.0 * A.size        # every level through most cells: more than the pool holds
        for n in (2, 0, 1):
            ctx.select_level(n)
            xyz, keys, tris = ctx.download_level0(counts[n])
            c1, x1, k1, t1 = single(A, values[n], 1)
            assert counts[n] == c1
            assert np.array_equal(keys, k1) and np.array_equal(tris, t1) and np.array_equal(xyz.view(np.uint32), x1.view(np.uint32))
        # another shape on the same context: pooled again
        g = np.linspace(-1, 1, 44)
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        B = (X * X + 0.8 * Y * Y + 1.2 * Z * Z).astype(np.float32)
        ctx.upload_grid(B)
        vb = [0.3, 0.6, 0.9]
        cb = ctx.extract3d_levels(vb, 1)
        for n in range(3):
            ctx.select_level(n)
            xyz, keys, tris = ctx.download_level0(cb[n])
            c1, x1, k1, t1 = single(B, vb[n], 1)
            assert cb[n] == c1 and np.array_equal(keys, k1) and np.array_equal(tris, t1) and np.array_equal(xyz.view(np.uint32), x1.view(np.uint32))
    finally:
        ctx.close()
