"""GPU: the fused emit kernel (vertex indices of neighbour cells computed from the stream kernel's sign words and
lane prefixes) against the staged kernels (per-cell table + cell records) and against the oracle.

Same numbering by construction: vertex records and index triples must be IDENTICAL arrays, not just equal sets."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_names

pytestmark = pytest.mark.gpu


def field(shape, seed, freq=(3.1, 2.7, 3.9)):
    rng = np.random.RandomState(seed)
    n0, n1, n2 = shape
    g0, g1, g2 = np.meshgrid(np.linspace(-1, 1, n0), np.linspace(-1, 1, n1), np.linspace(-1, 1, n2), indexing="ij")
    A = np.sin(freq[0] * g0 + 0.4) * np.cos(freq[1] * g1) + 0.8 * np.sin(freq[2] * g2 + 1.0) + 0.05 * rng.standard_normal(shape)
    return A.astype(np.float32)


def both_paths(A, v, diag, origin=(0, 0, 0)):
    from contourist_amd import _ffi
    out = []
    for extra in (_ffi.CX_KERNEL_FUSED, _ffi.CX_KERNEL_STAGED):
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


SHAPES = [(9, 7, 8), (5, 4, 4), (2, 2, 4), (33, 33, 36), (37, 41, 52), (29, 23, 67), (40, 36, 260), (24, 20, 300),
          (70, 19, 515), (130, 64, 64), (64, 70, 256), (16, 130, 512), (96, 96, 96)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("diag", [0, 1])
def test_fused_equals_staged_and_oracle(shape, diag):
    from oracle import level0
    A = field(shape, 5 + shape[0])
    v = 0.07
    (cf, pf, xf, kf, tf), (cs, ps, xs, ks, ts) = both_paths(A, v, diag)
    assert ps == 1, "CX_KERNEL_STAGED must run the staged kernels"
    assert pf == 2, "CX_KERNEL_FUSED must run the fused emit kernel (no sample of this field is within tolerance)"
    assert cf == cs
    assert np.array_equal(kf, ks), "vertex numbering differs between the fused and the staged kernels"
    assert np.array_equal(xf.view(np.uint32), xs.view(np.uint32)), "vertex coordinates differ bitwise"
    assert np.array_equal(tf, ts), "index triples differ"
    O = level0.march3d(A, v, diag_mode=diag)
    ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
    co = level0.canonical_level0(ko, O["xyz"], O["tris"])
    ch = level0.canonical_level0(kf.astype(np.int64), xf, tf.astype(np.int64))
    assert np.array_equal(co[0], ch[0]) and np.array_equal(co[2], ch[2])
    assert np.all(np.abs(ch[1] - co[1]) <= 1e-6 * np.abs(co[1]) + 1e-6)


@pytest.mark.parametrize("name", golden_names())
def test_fixtures_through_both_paths(name):
    """every reference fixture: fused path == staged path; fixtures with samples inside the np.allclose tolerances are
    sent through the staged kernels automatically (path 1), the others take the fused kernel (path 2)"""
    G = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    A, v = G["A"], float(G["value"])
    if min(A.shape) < 2 or A.shape[2] < 4:
        pytest.skip("rows shorter than 4 samples take the generic kernel")
    (cf, pf, xf, kf, tf), (cs, ps, xs, ks, ts) = both_paths(A, v, 1)
    assert ps == 1 and pf in (1, 2)
    assert cf == cs and np.array_equal(kf, ks) and np.array_equal(tf, ts)
    assert np.array_equal(xf.view(np.uint32), xs.view(np.uint32))
    if name in ("sphere32", "noise32_v0", "noise24_v0", "shells24", "blobs27"):
        assert pf == 2


def test_tolerance_path_falls_back_to_staged():
    """samples within the reference's tolerances of the isovalue.  ONE sample equal to it removes nothing (the np.allclose
    rules need a whole tetrahedron within tolerance): the wave recounts exactly, finds the counts the signs gave and stays
    on the common path.  A 3x3x3 block straddling the isovalue by 1e-7 drops tetrahedra and vertices: the fused kernel
    stands down on the device and the host re-runs the extraction through the staged kernels (exact per-cell path)."""
    from contourist_amd import _ffi
    from oracle import level0
    v = float(np.float32(0.07))
    for block, want_path in ((1, 2), (3, 1)):
        A = field((20, 24, 28), 3)
        gi, gj, gk = np.meshgrid(np.arange(block), np.arange(block), np.arange(block), indexing="ij")
        A[10:10 + block, 11:11 + block, 12:12 + block] = (0.07 + 1e-7 * (1 - 2 * ((gi + gj + gk) % 2)) * (block > 1)).astype(np.float32)
        (cf, pf, xf, kf, tf), (cs, ps, xs, ks, ts) = both_paths(A, v, 1)
        assert pf == want_path and ps == 1
        assert cf == cs and np.array_equal(kf, ks) and np.array_equal(tf, ts)
        assert np.array_equal(xf.view(np.uint32), xs.view(np.uint32))
        O = level0.march3d(A, v, diag_mode=1)
        ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
        co = level0.canonical_level0(ko, O["xyz"], O["tris"])
        ch = level0.canonical_level0(kf.astype(np.int64), xf, tf.astype(np.int64))
        assert np.array_equal(co[0], ch[0]) and np.array_equal(co[2], ch[2])
        assert np.all(np.abs(ch[1] - co[1]) <= 1e-6 * np.abs(co[1]) + 1e-6)
        # asynchronous form: the fallback happens when the counts are fetched
        ctx = _ffi.Context(0)
        try:
            ctx.reserve(cf["n_cells"] + 64, cf["n_vertices"] + 64, cf["n_triangles"] + 64)   # the async form does not grow buffers
            ctx.upload_grid(A)
            ctx.extract3d_async(v, 1 | _ffi.CX_KERNEL_FUSED)
            c = ctx.counts()
            assert c == cf and ctx.level0_path() == want_path
            x2, k2, t2 = ctx.download_level0(c)
            assert np.array_equal(k2, kf) and np.array_equal(t2, tf)
        finally:
            ctx.close()


def test_slab_origin_and_negative_origin():
    """CPython-order diagonals hash global lattice coordinates: both paths with a slab origin and with a rim origin"""
    A = field((21, 26, 40), 9)
    for origin in ((37, 0, 0), (-1, -1, -1)):
        (cf, pf, xf, kf, tf), (cs, ps, xs, ks, ts) = both_paths(A, 0.07, 1, origin)
        assert pf == 2 and ps == 1
        assert cf == cs and np.array_equal(kf, ks) and np.array_equal(tf, ts)


def test_seeded_selection_after_fused_extraction():
    """cell records are produced on demand for the seeded selection (the fused kernel writes none)"""
    from contourist_amd import _ffi
    from oracle import level0, seeds
    G = np.load(os.path.join(GOLDEN_DIR, "blobs27.npz"))
    A, v = G["A"], float(G["value"])
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        c = ctx.extract3d(v, 1 | _ffi.CX_KERNEL_FUSED)
        assert ctx.level0_path() == 2
        xyz, keys, tris = ctx.download_level0(c)
        keys = keys.astype(np.int64)
        O = level0.march3d(A, v, diag_mode=1)
        ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
        lin, d = keys >> 3, keys & 7
        n1n2 = A.shape[1] * A.shape[2]
        p = 0
        q = np.array([lin[p] // n1n2, (lin[p] // A.shape[2]) % A.shape[1], lin[p] % A.shape[2]])
        dv = np.array([(d[p] >> 2) & 1, (d[p] >> 1) & 1, d[p] & 1])
        eps = [[tuple(int(x) for x in q), tuple(int(x) for x in q + dv)]]
        want, _ = seeds.select(A, v, eps, ko, O["tris"])
        got = ctx.select_seeded(eps)
        assert got["triangles_kept"] == int(want.sum()) and 0 < got["triangles_kept"] < len(tris)
    finally:
        ctx.close()


def test_fused_and_staged_alternate_on_one_context():
    """both emit paths on ONE context, alternating, on grids of different sizes: each keeps its own side tables (a stray
    hipFree of the staged path's table in the fused path's regrow branch made the next staged extraction fault)"""
    from contourist_amd import _ffi
    ctx = _ffi.Context(0)
    try:
        for shape, seed in (((33, 33, 36), 1), ((70, 48, 132), 2), ((24, 20, 300), 3), ((96, 96, 96), 4)):
            A = field(shape, seed)
            ctx.upload_grid(A)
            ref = None
            for flags in (1, 1 | _ffi.CX_KERNEL_FUSED, 1 | _ffi.CX_KERNEL_STAGED, _ffi.CX_KERNEL_FUSED, 0, 1 | _ffi.CX_KERNEL_FUSED, 1):
                c = ctx.extract3d(0.07, flags)
                xyz, keys, tris = ctx.download_level0(c)
                if flags & 1:
                    if ref is None:
                        ref = (c, keys.copy(), tris.copy(), xyz.copy())
                    else:
                        assert c == ref[0] and np.array_equal(keys, ref[1]) and np.array_equal(tris, ref[2])
                        assert np.array_equal(xyz.view(np.uint32), ref[3].view(np.uint32))
    finally:
        ctx.close()
