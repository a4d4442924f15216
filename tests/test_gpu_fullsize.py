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
