"""GPU: one context across grids of different sizes, exact capacities and the public flag check.

Regressions for round-1 review findings: the seeded-selection mask was freed by an unrelated regrow path of the
extraction (use after free / double free at destroy), the ablation bits were honoured through the public `flags`
argument, and no test ran with the output capacities exactly equal to the counts (a read past the last cell record
or a write past the last triangle would go unnoticed otherwise)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _seeds_from_keys(keys, shape, picks):
    lin, d = keys >> 3, keys & 7
    n1n2 = shape[1] * shape[2]
    q = np.stack([lin // n1n2, (lin // shape[2]) % shape[1], lin % shape[2]], axis=1)
    dv = np.stack([(d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1)
    return [[tuple(int(x) for x in q[p]), tuple(int(x) for x in q[p] + dv[p])] for p in picks]


def _selection_matches_oracle(ctx, A, v):
    from contourist_amd import _ffi
    from oracle import level0, seeds
    ctx.upload_grid(A)
    counts = ctx.extract3d(v, _ffi.CX_DIAG_CPYTHON310)
    xyz, keys, tris = ctx.download_level0(counts)
    keys = keys.astype(np.int64)
    O = level0.march3d(A, v, diag_mode=1)
    ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
    assert np.array_equal(np.sort(keys), np.sort(ko))
    eps = _seeds_from_keys(keys, A.shape, [0, len(keys) // 2])
    want, _ = seeds.select(A, v, eps, ko, O["tris"])
    got = ctx.select_seeded(eps)
    assert got["triangles_kept"] == int(want.sum())
    tk, vk = ctx.seeded_masks(counts)
    # the device's kept triangles as key triples == the oracle's
    dev = set(tuple(sorted(int(k) for k in keys[t])) for t in tris[tk])
    ora = set(tuple(sorted(int(k) for k in ko[t])) for t in O["tris"][want])
    assert dev == ora
    post = ctx.postprocess3d(0)
    assert post["n_triangles"] > 0


def test_seeded_selection_small_grid_then_larger_grid_same_context():
    """extract(small) -> select -> extract(larger: every side table regrows) -> select -> destroy"""
    from contourist_amd import _ffi
    small = np.load(os.path.join(GOLDEN_DIR, "rel_tol16.npz"))
    mid = np.load(os.path.join(GOLDEN_DIR, "blobs27.npz"))
    big = np.load(os.path.join(GOLDEN_DIR, "noise32_v0.npz"))
    ctx = _ffi.Context(0)
    try:
        _selection_matches_oracle(ctx, small["A"], float(small["value"]))
        _selection_matches_oracle(ctx, mid["A"], float(mid["value"]))
        # a much larger grid on the same context: queues, batch records and the selection mask all grow
        n = 96
        g = np.linspace(-1.2, 1.2, n)
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        A = (np.sqrt(X * X + Y * Y + Z * Z) - 0.9 + 0.25 * np.sin(5 * X) * np.sin(4 * Y) * np.sin(3 * Z)).astype(np.float32)
        _selection_matches_oracle(ctx, A, 0.0)
        _selection_matches_oracle(ctx, big["A"], float(big["value"]))
        _selection_matches_oracle(ctx, small["A"], float(small["value"]))
    finally:
        ctx.close()       # a double free would abort the process here


def test_unknown_flag_bits_are_rejected():
    from contourist_amd import _ffi
    G = np.load(os.path.join(GOLDEN_DIR, "sphere32.npz"))
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(G["A"])
        for bad in (0x100000, 0x10000, 0x2, 0x80000000):
            with pytest.raises(_ffi.CxError) as e:
                ctx.extract3d(float(G["value"]), _ffi.CX_DIAG_CPYTHON310 | bad)
            assert e.value.code == -1
            with pytest.raises(_ffi.CxError):
                ctx.extract3d_async(float(G["value"]), bad)
        c = ctx.extract3d(float(G["value"]), _ffi.CX_DIAG_CPYTHON310 | _ffi.CX_KERNEL_GENERIC)
        assert c["n_vertices"] == 6386 and c["n_triangles"] == 12768
    finally:
        ctx.close()


@pytest.mark.parametrize("shape,generic", [((37, 41, 52), False), ((37, 41, 52), True), ((29, 23, 67), False), ((64, 64, 64), False)])
def test_exact_capacities(shape, generic):
    """cx_reserve with exactly the counts of the surface (ccap == n_cells, vcap == n_vertices, tcap == n_triangles) on
    grids whose record count is no multiple of any kernel stride: same mesh as with room to spare, as the oracle's."""
    from contourist_amd import _ffi
    from oracle import level0
    rng = np.random.RandomState(11)
    n0, n1, n2 = shape
    g0, g1, g2 = np.meshgrid(np.linspace(-1, 1, n0), np.linspace(-1, 1, n1), np.linspace(-1, 1, n2), indexing="ij")
    A = (np.sin(3.1 * g0 + 0.4) * np.cos(2.7 * g1) + 0.8 * np.sin(3.9 * g2 + 1.0) + 0.05 * rng.standard_normal(shape)).astype(np.float32)
    flags = _ffi.CX_DIAG_CPYTHON310 | (_ffi.CX_KERNEL_GENERIC if generic else 0)
    O = level0.march3d(A, 0.1, diag_mode=1)
    ko = level0.edge_keys_from_pairs(O["pairs"], A.shape)
    co = level0.canonical_level0(ko, O["xyz"], O["tris"])
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        loose = ctx.extract3d(0.1, flags)
        assert loose["n_vertices"] == len(ko) and loose["n_triangles"] == len(O["tris"])
    finally:
        ctx.close()
    ctx = _ffi.Context(0)          # fresh context: buffers of exactly the size needed
    try:
        ctx.reserve(max(loose["n_cells"], 1), loose["n_vertices"], loose["n_triangles"])
        ctx.upload_grid(A)
        c = ctx.extract3d(0.1, flags)
        assert c == loose
        xyz, keys, tris = ctx.download_level0(c)
        ch = level0.canonical_level0(keys.astype(np.int64), xyz, tris.astype(np.int64))
        assert np.array_equal(co[0], ch[0]) and np.array_equal(co[2], ch[2])
        # one vertex / triangle short: reported as a capacity problem, grown and re-run by the synchronous call
        ctx2 = _ffi.Context(0)
        try:
            ctx2.reserve(max(loose["n_cells"] - 1, 1), loose["n_vertices"] - 1, loose["n_triangles"] - 1)
            ctx2.upload_grid(A)
            assert ctx2.extract3d(0.1, flags) == loose
            xyz2, keys2, tris2 = ctx2.download_level0(loose)
            ch2 = level0.canonical_level0(keys2.astype(np.int64), xyz2, tris2.astype(np.int64))
            assert np.array_equal(co[0], ch2[0]) and np.array_equal(co[2], ch2[2])
        finally:
            ctx2.close()
    finally:
        ctx.close()


def test_vertex_records_and_expanded_coordinates_agree():
    """the march's 8-byte vertex records {edge id, fp32 t} against the float4 {x, y, z, id} cx_level0_download expands them to:
    same ids, same order, and q + t d reproduces the coordinates bit for bit (the expansion is that sum in fp32)"""
    from contourist_amd import _ffi
    rng = np.random.RandomState(77)
    A = rng.standard_normal((21, 18, 37)).astype(np.float32)
    ctx = _ffi.Context(0)
    try:
        ctx.upload_grid(A)
        c = ctx.extract3d(0.25, _ffi.CX_DIAG_CPYTHON310)
        xyz, keys, tris = ctx.download_level0(c)
        ids, t, tris2 = ctx.download_level0_records(c)
        assert np.array_equal(ids, keys) and np.array_equal(tris, tris2)
        assert np.all((t >= 0) & (t <= 1))
        lin, d = ids.astype(np.int64) >> 3, ids.astype(np.int64) & 7
        q = np.stack(np.unravel_index(lin, A.shape), axis=1).astype(np.float32)
        step = np.stack([(d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1).astype(bool)
        want = np.where(step, q + t[:, None], q).astype(np.float32)
        assert np.array_equal(want.view(np.uint32), xyz.view(np.uint32))
    finally:
        ctx.close()


def test_slab_step_of_a_single_rank_is_adopt_plus_extract():
    "cx_slab_step with world == 1 (no exchange): the same mesh as adopt + extract, and a communicator is only asked for when world > 1"
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi
    rng = np.random.RandomState(5)
    A = torch.from_numpy(rng.standard_normal((19, 16, 24)).astype(np.float32)).cuda()
    ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
        c0 = ctx.extract3d(0.1, 1)
        a = ctx.download_level0(c0)
        ctx.slab_step(A.data_ptr(), A.shape[0], A.shape[1], A.shape[2], 0, 1, 0.1, 1, keepalive=A)
        c1 = ctx.counts()
        b = ctx.download_level0(c1)
        assert c0 == c1 and all(np.array_equal(x, y) for x, y in zip(a, b))
        with pytest.raises(_ffi.CxError):        # two ranks without a communicator: refused before anything is enqueued
            ctx.slab_step(A.data_ptr(), A.shape[0] - 1, A.shape[1], A.shape[2], 0, 2, 0.1, 1, keepalive=A)
        with pytest.raises(_ffi.CxError):
            ctx.halo_exchange(None, 0, 2, A.data_ptr(), A.shape[0] - 1, A.shape[1] * A.shape[2])
    finally:
        ctx.close()
