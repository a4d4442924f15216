"""GPU: the bench-size and the maximum-size grid, checked through size-independent properties
(no oracle runs at these sizes): closed surfaces have Euler characteristic 2 per component
(V - T/2 = 2 for a triangulated sphere), edge ids are unique and decode to crossing edges,
every triangle is wound from low to high, and an extraction repeats bit-exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sphere_field(shape, centre, torch, dev):
    ax = [torch.arange(n, dtype=torch.float32, device=dev) - c for n, c in zip(shape, centre)]
    return (ax[0][:, None, None] ** 2 + ax[1][None, :, None] ** 2 + ax[2][None, None, :] ** 2).contiguous()


def extract_on_device(ctx, A, value, flags):
    ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
    counts = ctx.extract3d(value, flags)
    xyz, keys, tris = ctx.download_level0(counts)
    return counts, xyz, keys.astype(np.int64), tris.astype(np.int64)


def check_closed_sphere(counts, xyz, keys, tris, shape, centre, radius):
    V, T = counts["n_vertices"], counts["n_triangles"]
    assert V > 0 and T > 0 and V - T // 2 == 2 and T % 2 == 0             # Euler characteristic of a sphere
    assert len(np.unique(keys)) == V                                       # one vertex per crossing edge
    assert tris.min() >= 0 and tris.max() < V
    # every vertex lies on its lattice edge and close to the sphere (linear interpolation of r^2: error < 1/(8 r) voxel... loose bound)
    lin, d = keys >> 3, keys & 7
    n1n2 = shape[1] * shape[2]
    q = np.stack([lin // n1n2, (lin % n1n2) // shape[2], lin % shape[2]], axis=1).astype(np.float64)
    dv = np.stack([(d >> 2) & 1, (d >> 1) & 1, d & 1], axis=1).astype(np.float64)
    t = ((xyz - q) * dv).sum(axis=1) / dv.sum(axis=1)
    assert np.all((t >= 0) & (t <= 1)) and np.allclose(xyz, q + dv * t[:, None], atol=1e-4)
    r = np.linalg.norm(xyz - np.asarray(centre, dtype=np.float64), axis=1)
    assert np.all(np.abs(r - radius) < 0.75)
    # winding: normals point from low (inside) to high (outside)
    p0, p1, p2 = xyz[tris[:, 0]], xyz[tris[:, 1]], xyz[tris[:, 2]]
    n = np.cross(p1 - p0, p2 - p0)
    out = (p0 + p1 + p2) / 3.0 - np.asarray(centre, dtype=np.float64)
    s = np.einsum("ij,ij->i", n, out)
    assert np.all(s[np.linalg.norm(n, axis=1) > 1e-9] > 0)


@pytest.mark.parametrize("shape", [(512, 512, 512), (1024, 1024, 512), (257, 300, 1030)])
def test_sphere_at_full_size(shape):
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi
    dev = torch.device("cuda", 0)
    centre = [(n - 1) / 2.0 + 0.25 for n in shape]
    radius = 0.4 * min(shape)
    A = sphere_field(shape, centre, torch, dev)
    ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        res = extract_on_device(ctx, A, radius * radius, _ffi.CX_DIAG_CPYTHON310)
        check_closed_sphere(*res, shape, centre, radius)
        again = extract_on_device(ctx, A, radius * radius, _ffi.CX_DIAG_CPYTHON310)
        assert res[0] == again[0]
        for a, b in zip(res[1:], again[1:]):
            assert np.array_equal(a, b)                                    # deterministic, bit for bit
    finally:
        ctx.close()
        del A
        torch.cuda.empty_cache()


def test_more_than_2_29_samples_is_refused():
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi
    A = torch.zeros((1025, 1024, 512), dtype=torch.float32, device="cuda:0")
    ctx = _ffi.Context(0)
    try:
        with pytest.raises(_ffi.CxError):
            ctx.adopt_device_grid(A.data_ptr(), tuple(A.shape), keepalive=A)
    finally:
        ctx.close()


def edge_consistency(tris):
    """(manifold edges, manifold edges whose two triangles run along them in the SAME direction).
    A consistently wound surface has none of the latter (surface_geometry.py:110-138)."""
    t = np.asarray(tris, dtype=np.int64)
    a = np.concatenate([t[:, 0], t[:, 1], t[:, 2]])
    b = np.concatenate([t[:, 1], t[:, 2], t[:, 0]])
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    fwd = (a == lo).astype(np.int64)
    key = lo * (t.max() + 1) + hi
    order = np.argsort(key, kind="stable")
    key, fwd = key[order], fwd[order]
    start = np.concatenate([[True], key[1:] != key[:-1]])
    idx = np.nonzero(start)[0]
    count = np.diff(np.concatenate([idx, [len(key)]]))
    two = idx[count == 2]
    same = int(np.sum(fwd[two] == fwd[two + 1]))
    return len(two), same, int(np.sum(count != 2))


def test_level1_orientation_at_full_size():
    """Level 1 on a 512^3 grid (weld buckets of 1/19 voxel): a thick spherical shell, i.e. two concentric spheres whose
    Level-0 windings are opposite (low is between them).  Every manifold edge is run in opposite directions by its two
    triangles, and the reference's rule (normal_x > 0 at the vertex with the largest x,
    surface_geometry.py:79-103) makes BOTH enclose a positive volume."""
    torch = pytest.importorskip("torch")
    from contourist_amd import _ffi
    dev = torch.device("cuda", 0)
    shape = (512, 512, 512)
    r2 = sphere_field(shape, (250.25, 260.5, 255.75), torch, dev)
    A = ((r2 - 200.0 ** 2) * (r2 - 110.0 ** 2) * 1e-4).contiguous()
    del r2
    ctx = _ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        ctx.adopt_device_grid(A.data_ptr(), shape, keepalive=A)
        ctx.extract3d(0.0, _ffi.CX_DIAG_CPYTHON310)
        post = ctx.postprocess3d(0)
        pts, tris = ctx.download_level1(post)
        assert post["n_components"] == 2 and post["n_triangles"] == len(tris) > 500000
        # edges shared by 3+ triangles (the weld pinches sheets together; ~1 000 here) cannot have all of their triangles
        # pairwise opposite; the linking never flips across them (cx_post.hip, cxp_k_edges_link), so every MANIFOLD edge
        # stays consistently wound
        manifold, same, other = edge_consistency(tris)
        assert manifold > 0.99 * 1.5 * len(tris) and other < 0.001 * manifold and same <= 1e-5 * manifold
        # signed volume of each sphere (split by the distance from the centre): outward normals on both
        rr = np.linalg.norm(pts[tris].mean(axis=1) - np.array([250.25, 260.5, 255.75]), axis=1)
        vols = []
        for part in (rr < 155.0, rr >= 155.0):
            p = pts[tris[part]]
            vols.append(np.einsum("ij,ij->i", p[:, 0], np.cross(p[:, 1], p[:, 2])).sum() / 6.0)
        assert abs(vols[0] / (4.0 / 3.0 * np.pi * 110.0 ** 3) - 1.0) < 0.01 and abs(vols[1] / (4.0 / 3.0 * np.pi * 200.0 ** 3) - 1.0) < 0.01
        again = ctx.postprocess3d(0)
        assert again == post
    finally:
        ctx.close()
        del A
        torch.cuda.empty_cache()


def _two_balls(shape, dev=None):
    "f = min distance to two ball centres minus the radii (fp32), two closed components well inside the volume"
    import torch
    ax = [torch.arange(n, dtype=torch.float32, device=dev) for n in shape]
    X, Y, Z = torch.meshgrid(*ax, indexing="ij")
    c1 = (0.30 * shape[0], 0.45 * shape[1], 0.50 * shape[2]); r1 = 0.18 * min(shape)
    c2 = (0.72 * shape[0], 0.55 * shape[1], 0.48 * shape[2]); r2 = 0.13 * min(shape)
    d1 = torch.sqrt((X - c1[0]) ** 2 + (Y - c1[1]) ** 2 + (Z - c1[2]) ** 2) - r1
    del X
    d2 = torch.sqrt((ax[0][:, None, None] - c2[0]) ** 2 + (Y - c2[1]) ** 2 + (Z - c2[2]) ** 2) - r2
    del Y, Z
    return torch.minimum(d1, d2).contiguous(), (c1, r1, c2, r2)


def _canonical_mesh(points, tris):
    "vertices in lexicographic order, triangles renumbered, each rotated to start at its smallest index (winding kept), rows sorted"
    P = np.asarray(points, dtype=np.float64)
    T = np.asarray(tris, dtype=np.int64)
    order = np.lexsort((P[:, 2], P[:, 1], P[:, 0]))
    rank = np.empty(len(P), dtype=np.int64)
    rank[order] = np.arange(len(P))
    T = rank[T]
    r = np.argmin(T, axis=1)
    rows = np.arange(len(T))
    T = np.stack([T[rows, r], T[rows, (r + 1) % 3], T[rows, (r + 2) % 3]], axis=1)
    T = T[np.lexsort((T[:, 2], T[:, 1], T[:, 0]))]
    return P[order], T


def test_volume_in_slabs_equals_the_single_extraction():
    """A volume with more samples than one extraction addresses goes through the device slab by slab (GridContour3d._post_in_slabs):
    with the limit lowered, a 70 x 48 x 52 volume in 5 slabs (and in 2, and with a last slab that absorbs a single plane) gives the
    points and triangles of the single extraction, bit for bit up to the order of the vertices -- host array, device tensor, a smooth noise field and two balls"""
    torch = pytest.importorskip("torch")
    from contourist_amd import tetrahedral, synthetic
    fields = [synthetic.smooth_noise_host((70, 48, 52), 77, passes=30) if hasattr(synthetic, "smooth_noise_host") else None]
    fields = [f for f in fields if f is not None]
    fields.append(_two_balls((70, 48, 52))[0].numpy())
    for A in fields:
        A = np.ascontiguousarray(A, dtype=np.float32)
        corner = tuple(n - 1 for n in A.shape)
        ref = tetrahedral.GridContour3d(corner, A, 0.0)
        p0, t0 = ref.get_points_and_triangles()
        assert len(t0) > 1000
        P0, T0 = _canonical_mesh(p0, t0)
        assert len(np.unique(P0, axis=0)) == len(P0)          # (welded: no two vertices coincide, the canonical order is unique)
        for limit, on_device in ((48 * 52 * 16, False), (48 * 52 * 36, True), (48 * 52 * 24, False)):
            S = torch.from_numpy(A).cuda() if on_device else A
            m = tetrahedral.GridContour3d(corner, S, 0.0)
            m.MAX_SAMPLES_PER_EXTRACTION = limit
            assert m._in_slabs()
            p1, t1 = m.get_points_and_triangles()
            assert m._slab_counts["n_slabs"] >= 2
            P1, T1 = _canonical_mesh(p1, t1)          # (the vertex order differs: ascending edge id here, march order there)
            assert np.array_equal(P1, P0) and np.array_equal(T1, T0)
            with pytest.raises(NotImplementedError):
                m.level0()


def test_volume_beyond_one_extraction():
    """1056 x 720 x 720 fp32 = 547 M samples (> 2^29) resident on the GPU, two balls: get_points_and_triangles marches it in slabs;
    the result is a consistently wound two-component surface on the balls (every point within a quarter voxel of its sphere, no
    manifold edge run in the same direction by its two triangles; where the weld left no pinched edge: Euler characteristic 2 per component)"""
    torch = pytest.importorskip("torch")
    from contourist_amd import tetrahedral
    shape = (1056, 720, 720)
    assert shape[0] * shape[1] * shape[2] > (1 << 29)
    A, (c1, r1, c2, r2) = _two_balls(shape, torch.device("cuda", 0))
    m = tetrahedral.GridContour3d(tuple(n - 1 for n in shape), A, 0.0)
    assert m._in_slabs()
    pts, tris = m.get_points_and_triangles()
    pts, tris = np.asarray(pts), np.asarray(tris)
    assert m._slab_counts["n_slabs"] >= 2 and m._post["n_components"] == 2
    d = np.minimum(np.abs(np.linalg.norm(pts - np.array(c1), axis=1) - r1), np.abs(np.linalg.norm(pts - np.array(c2), axis=1) - r2))
    assert d.max() < 0.25              # linear interpolation on a unit lattice + the weld buckets of this corner (1/9 voxel)
    manifold, same, other = edge_consistency(tris)
    assert len(tris) > 2000000 and same == 0 and other <= 1e-3 * manifold
    if other == 0:
        assert manifold * 2 == len(tris) * 3 and len(pts) - manifold + len(tris) == 4      # V - E + F = 2 per closed component
