"""GPU: Level 1 sharded over slabs (distributed.level1_slabs_sharded) gives, slab by slab, exactly the mesh
the undivided volume gives: same surviving vertices (by edge id) with bit-identical float64 coordinates, same triangles,
same winding.  The ranks are played one after the other by several contexts on the one GPU of the box: the exchange itself
(distributed.merge_shard_components) is host code; the two-process run over gloo is in tests/test_gpu_distributed.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def fields():
    F = {}
    n = 72
    x, y, z = np.meshgrid(*[np.arange(n, dtype=np.float64)] * 3, indexing="ij")
    # one component through every slab, leaving the volume through its faces (max-x ties on the face x = n-1)
    F["gyroid"] = ((np.sin(x * 0.23) * np.cos(y * 0.23) + np.sin(y * 0.23) * np.cos(z * 0.23) + np.sin(z * 0.23) * np.cos(x * 0.23)).astype(np.float32), 0.1)
    # many closed components, some inside one slab, some across a boundary; centres on lattice points: crossings AT lattice
    # points and tiny triangles next to the slab boundaries
    r = np.full((n, n, n), 1e9)
    rng = np.random.RandomState(5)
    for c in rng.randint(6, n - 6, size=(40, 3)):
        r = np.minimum(r, np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2))
    F["balls"] = (r.astype(np.float32), 4.0)
    # values on a coarse set: equal samples, crossings exactly at lattice points, degenerate triangles everywhere
    F["steps"] = ((np.round(np.sin(x * 0.3) * np.sin(y * 0.31) * np.sin(z * 0.29) * 4) / 4).astype(np.float32), 0.25)
    F["noise"] = (rng.rand(n, n, n).astype(np.float32), 0.5)
    return F


def canon(keys, tris):
    "oriented triangles as rows of vertex edge ids, rotated so that the smallest comes first"
    k = np.asarray(keys, dtype=np.int64)[np.asarray(tris, dtype=np.int64).reshape(-1, 3)]
    s = np.argmin(k, axis=1)
    rows = np.arange(len(k))
    k = np.stack([k[rows, s], k[rows, (s + 1) % 3], k[rows, (s + 2) % 3]], axis=1)
    return k[np.lexsort(k.T[::-1])]


@pytest.mark.parametrize("name", ["gyroid", "balls", "steps", "noise"])
@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_level1_equals_the_undivided_volume(name, world):
    from contourist_amd import _ffi, distributed
    A, value = fields()[name]
    whole = _ffi.Context(0)
    whole.upload_grid(A)
    whole.extract3d(value, _ffi.CX_DIAG_CPYTHON310)
    post = whole.postprocess3d(0)
    wp, wt = whole.download_level1(post)
    wk = whole.download_level1_keys(post).astype(np.int64)
    assert len(np.unique(wk)) == len(wk)
    ctxs = [_ffi.Context(0) for _ in range(world)]
    lists, lays = [], []
    for r in range(world):
        lay = distributed.shard_layout(A.shape[0], world, r)
        lays.append(lay)
        lists.append(distributed.shard_local(ctxs[r], np.ascontiguousarray(A[lay["e0"]:lay["e1"]]), lay, value, A.shape))
    answers, stats = distributed.merge_shard_components(lists)
    assert stats["unmatched"] == 0, stats
    parts = []
    for r in range(world):
        out = distributed.shard_finish(ctxs[r], lists[r], answers[r])
        parts.append((out["keys"], out["points"], out["triangles"]))
    keys, pts, tris = distributed.assemble_level1(parts)
    print(name, world, "vertices", len(wk), "triangles", len(wt), "boundary", [len(L["tri_label"]) for L in lists], stats)
    order = np.argsort(wk)
    assert np.array_equal(keys, wk[order])
    assert np.array_equal(pts, wp[order])                     # bit for bit
    assert sum(len(p[2]) for p in parts) == len(wt)             # every triangle is in exactly one slab
    assert np.array_equal(canon(keys, tris), canon(wk, wt))
    # the work rank 0 does follows the boundary, not the volume
    # (four layers of cells per boundary: at most 4 x the fullest layer of the surface, with room for welded-away triangles)
    per_layer = np.bincount(np.floor(wp[wt].min(axis=1)[:, 0]).astype(np.int64), minlength=A.shape[0]) if len(wt) else np.zeros(1)
    assert sum(len(L["tri_label"]) for L in lists) <= 6 * (world - 1) * int(per_layer.max()) + 64
