"""GPU: the tile emit path (cx_k_tile_emit + cx_k_tile_boundary, CX_KERNEL_TILED: one workgroup per tile of the streaming pass,
the hand-over from the vertex numbering to the triangles in LDS) against the staged kernels and against the oracle.

Same numbering by construction: vertex records and index triples must be IDENTICAL arrays, not just equal sets.
Reference: tetrahedral.py:554-595 (the triangles of every voxel), 471-512 (the interpolated pairs)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_names

pytestmark = pytest.mark.gpu


def field(shape, seed, freq=(3.1, 2.7, 3.9), noise=0.05):
    rng = np.random.RandomState(seed)
    n0, n1, n2 = shape
    g0, g1, g2 = np.meshgrid(np.linspace(-1, 1, n0), np.linspace(-1, 1, n1), np.linspace(-1, 1, n2), indexing="ij")
    A = np.sin(freq[0] * g0 + 0.4) * np.cos(freq[1] * g1) + 0.8 * np.sin(freq[2] * g2 + 1.0) + noise * rng.standard_normal(shape)
    return A.astype(np.float32)


def both_paths(A, v, diag, origin=(0, 0, 0)):
    from contourist_amd import _ffi
    out = []
    for extra in (_ffi.CX_KERNEL_TILED, _ffi.CX_KERNEL_STAGED):
        ctx = _ffi.Context(0)
        try:
            ctx.set_origin(*origin)
            ctx.upload_grid(A)
            c = ctx.extract3d(v, diag | extra)
            path = ctx.level0_path()
            xyz, keys, tris = ctx.download_level0(c)
            out.append((c, path, xyz, keys, tris))
        finally:
            ctx.close()
    return out


def same(a, b):
    (ca, pa, xa, ka, ta), (cb, pb, xb, kb, tb) = a, b
    assert ca == cb
    assert np.array_equal(ka, kb), "vertex numbering differs between the tile path and the staged kernels"
    assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)), "vertex coordinates differ bitwise"
    bad = np.nonzero((ta != tb).any(axis=1))[0]
    assert len(bad) == 0, "index triples differ: %d of %d, first at %s: %s vs %s" % (len(bad), len(ta), bad[:4], ta[bad[:2]], tb[bad[:2]])


def against_oracle(A, v, diag, x, k, t):
    from oracle import level0
    O = level0.march3d(A, v, diag_mode=diag)
    ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
    co = level0.canonical_level0(ko, O["xyz"], O["tris"])
    ch = level0.canonical_level0(k.astype(np.int64), x, t.astype(np.int64))
    assert np.array_equal(co[0], ch[0]) and np.array_equal(co[2], ch[2])
    assert np.all(np.abs(ch[1] - co[1]) <= 1e-6 * np.abs(co[1]) + 1e-6)


# one tile; several tiles in k (> 256 samples), in j (> 16 rows), in i (more planes than a chunk); ragged rows; array edges inside a tile
SHAPES = [(9, 7, 8), (5, 4, 4), (2, 2, 4), (33, 33, 36), (37, 41, 52), (29, 23, 67), (40, 36, 260), (24, 20, 300),
          (70, 19, 515), (130, 64, 64), (64, 70, 256), (16, 130, 512), (96, 96, 96), (12, 17, 257), (9, 16, 256), (9, 32, 512),
          (40, 16, 256), (48, 130, 512)]
# few planes over a field that varies fast along i: a third of all cells has a crossing -- more per tile than the LDS words hold
CROWDED = {(16, 130, 512), (9, 16, 256), (9, 32, 512)}


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("diag", [0, 1])
def test_tiled_equals_staged_and_oracle(shape, diag):
    A = field(shape, 5 + shape[0], noise=0.05 if max(shape) <= 100 else 0.004)   # (long rows: noise well below the step per sample, or whole tiles are surface)
    v = 0.07
    t, s = both_paths(A, v, diag)
    assert s[1] == 1, "CX_KERNEL_STAGED must run the staged kernels"
    assert t[1] == (1 if shape in CROWDED else 3), "CX_KERNEL_TILED must run the tile kernels (no sample within tolerance) unless a tile is beyond the LDS words"
    same(t, s)
    against_oracle(A, v, diag, t[2], t[3], t[4])


@pytest.mark.parametrize("name", golden_names())
def test_fixtures_through_both_paths(name):
    """every reference fixture: tile path == staged path; fixtures with samples inside the np.allclose tolerances (or with more
    surface cells in a tile than the LDS words hold) are sent through the staged kernels automatically (path 1)"""
    G = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    A, v = G["A"], float(G["value"])
    if min(A.shape) < 2 or A.shape[2] < 4:
        pytest.skip("rows shorter than 4 samples take the generic kernel")
    t, s = both_paths(A, v, 1)
    assert s[1] == 1 and t[1] in (1, 3)
    same(t, s)
    if name in ("sphere32", "shells24", "blobs27"):
        assert t[1] == 3


def test_tolerance_path_and_crowded_tiles_fall_back_to_staged():
    """(i) a 3x3x3 block straddling the isovalue by 1e-7 drops tetrahedra and vertices (np.allclose rules): the tile kernels stand down on
    the device and the host re-runs the extraction through the staged kernels; ONE sample equal to the isovalue removes nothing and
    stays on the tile path.  (ii) white noise: more surface cells in a tile than a workgroup's LDS words -- the same fallback."""
    from contourist_amd import _ffi
    v = float(np.float32(0.07))
    for block, want_path in ((1, 3), (3, 1)):
        A = field((20, 24, 28), 3)
        gi, gj, gk = np.meshgrid(np.arange(block), np.arange(block), np.arange(block), indexing="ij")
        A[10:10 + block, 11:11 + block, 12:12 + block] = (0.07 + 1e-7 * (1 - 2 * ((gi + gj + gk) % 2)) * (block > 1)).astype(np.float32)
        t, s = both_paths(A, v, 1)
        assert t[1] == want_path and s[1] == 1
        same(t, s)
        against_oracle(A, v, 1, t[2], t[3], t[4])
        # asynchronous form: the fallback happens when the counts are fetched
        ctx = _ffi.Context(0)
        try:
            cf = t[0]
            ctx.reserve(cf["n_cells"] + 64, cf["n_vertices"] + 64, cf["n_triangles"] + 64)   # the async form does not grow buffers
            ctx.upload_grid(A)
            ctx.extract3d_async(v, 1 | _ffi.CX_KERNEL_TILED)
            c = ctx.counts()
            assert c == cf and ctx.level0_path() == want_path
            x2, k2, t2 = ctx.download_level0(c)
            assert np.array_equal(k2, t[3]) and np.array_equal(t2, t[4])
        finally:
            ctx.close()
    rng = np.random.RandomState(7)
    A = rng.standard_normal((24, 40, 300)).astype(np.float32)
    t, s = both_paths(A, 0.1, 1)
    assert t[1] == 1 and s[1] == 1
    same(t, s)
    against_oracle(A, 0.1, 1, t[2], t[3], t[4])


def test_slab_origin_and_negative_origin():
    """CPython-order diagonals hash global lattice coordinates: both paths with a slab origin and with a rim origin"""
    A = field((21, 26, 40), 9)
    for origin in ((37, 0, 0), (-1, -1, -1)):
        t, s = both_paths(A, 0.07, 1, origin)
        assert t[1] == 3 and s[1] == 1
        same(t, s)


def test_level1_after_tiled_extraction():
    """Level 1 (weld / tiny collapse / clean / orient) on the tile path's mesh equals Level 1 on the staged kernels' mesh"""
    from contourist_amd import _ffi
    A = field((40, 44, 48), 12)
    res = []
    for extra in (_ffi.CX_KERNEL_TILED, _ffi.CX_KERNEL_STAGED):
        ctx = _ffi.Context(0)
        try:
            ctx.upload_grid(A)
            c = ctx.extract3d(0.07, 1 | extra)
            l1 = ctx.postprocess3d()
            res.append((c, l1, ctx.download_level1(l1)))
        finally:
            ctx.close()
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)


def test_seeded_selection_after_tiled_extraction():
    """cell records are produced on demand for the seeded selection (the tile kernels write none): the voxel groups the reference's
    breadth-first search reaches from one crossing edge (tetrahedral.py:396-463), against oracle/seeds.py"""
    from contourist_amd import _ffi
    from oracle import level0, seeds
    G = np.load(os.path.join(GOLDEN_DIR, "blobs27.npz"))
    A, v = G["A"], float(G["value"])
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        c = ctx.extract3d(v, 1 | _ffi.CX_KERNEL_TILED)
        assert ctx.level0_path() == 3
        xyz, keys, tris = ctx.download_level0(c)
        keys = keys.astype(np.int64)
        O = level0.march3d(A, v, diag_mode=1)
        ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
        lin, d = keys >> 3, keys & 7
        n1n2 = A.shape[1] * A.shape[2]
        q = np.array([lin[0] // n1n2, (lin[0] // A.shape[2]) % A.shape[1], lin[0] % A.shape[2]])
        dv = np.array([(d[0] >> 2) & 1, (d[0] >> 1) & 1, d[0] & 1])
        eps = [[tuple(int(x) for x in q), tuple(int(x) for x in q + dv)]]
        want, _ = seeds.select(A, v, eps, ko, O["tris"])
        got = ctx.select_seeded(eps)
        assert got["triangles_kept"] == int(want.sum()) and 0 < got["triangles_kept"] < len(tris)
    finally:
        ctx.close()


def test_tiled_and_staged_alternate_on_one_context_and_many_chunks():
    """one context, paths alternating, a grid tall enough for several chunks of planes per tile column"""
    from contourist_amd import _ffi
    A = field((150, 40, 300), 21, noise=0.02)
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        ref = None
        for flags in (1 | _ffi.CX_KERNEL_STAGED, 1 | _ffi.CX_KERNEL_TILED, 1, 1 | _ffi.CX_KERNEL_TILED, _ffi.CX_KERNEL_TILED, 0):
            c = ctx.extract3d(0.07, flags)
            x, k, t = ctx.download_level0(c)
            if flags & 1:
                if ref is None:
                    ref = (c, x.copy(), k.copy(), t.copy())
                else:
                    assert c == ref[0] and np.array_equal(k, ref[2]) and np.array_equal(t, ref[3]) and np.array_equal(x.view(np.uint32), ref[1].view(np.uint32))
            else:
                if flags & _ffi.CX_KERNEL_TILED:
                    canon = (c, k.copy(), t.copy())
                else:
                    assert c == canon[0] and np.array_equal(k, canon[1]) and np.array_equal(t, canon[2])
    finally:
        ctx.close()
