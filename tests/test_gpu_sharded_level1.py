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
    # one ball in the first slab, one that just reaches INTO the copies of the next slab's cells, nothing anywhere else: ranks
    # without a single triangle, components that consist of a neighbour's triangles only
    F["lonely"] = (np.minimum(np.sqrt((x - 9.0) ** 2 + (y - 30) ** 2 + (z - 30) ** 2) - 5.0,
                              np.sqrt((x - 36.6) ** 2 + (y - 20) ** 2 + (z - 50) ** 2) - 1.2).astype(np.float32), 0.0)
    return F


def play_ranks(A, value, world, shape=None):
    """the ranks one after the other, a context each: -> (parts [(keys, points, triangles)], lists, stats)"""
    import torch
    from contourist_amd import _ffi, distributed
    dev = torch.device("cuda", 0)
    shape = tuple(A.shape) if shape is None else shape
    ctxs = [_ffi.Context(0) for _ in range(world)]
    lists = []
    for r in range(world):
        lay = distributed.shard_layout(shape[0], world, r)
        local = A[lay["e0"]:lay["e1"]]
        if not torch.is_tensor(local):
            local = np.ascontiguousarray(local)
        lists.append(distributed.shard_local(ctxs[r], local, lay, value, shape, torch_device=dev))
    small = []
    for r in range(world):
        if r == 0:
            pairs, unmatched = np.zeros((0, 2), dtype=np.int64), 0
        else:
            pairs, unmatched = distributed.pair_labels(lists[r]["own1"][0], lists[r]["own1"][1], lists[r - 1]["copy4"][0], lists[r - 1]["copy4"][1])
        small.append(distributed.shard_small(lists[r], pairs, unmatched))
    answers, stats = distributed.merge_shard_components(small)
    parts = []
    for r in range(world):
        out = distributed.shard_finish(ctxs[r], lists[r], answers[r])
        parts.append((out["keys"], out["points"], out["triangles"]))
    return parts, lists, stats


def canon(keys, tris):
    "oriented triangles as rows of vertex edge ids, rotated so that the smallest comes first"
    k = np.asarray(keys, dtype=np.int64)[np.asarray(tris, dtype=np.int64).reshape(-1, 3)]
    s = np.argmin(k, axis=1)
    rows = np.arange(len(k))
    k = np.stack([k[rows, s], k[rows, (s + 1) % 3], k[rows, (s + 2) % 3]], axis=1)
    return k[np.lexsort(k.T[::-1])]


@pytest.mark.parametrize("name", ["gyroid", "balls", "steps", "noise", "lonely"])
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
    parts, lists, stats = play_ranks(A, value, world)
    assert stats["unmatched"] == 0, stats
    keys, pts, tris = distributed.assemble_level1(parts)
    print(name, world, "vertices", len(wk), "triangles", len(wt), "boundary", [L["n_own_lower"] + L["n_upper_copies"] for L in lists], stats)
    order = np.argsort(wk)
    assert np.array_equal(keys, wk[order])
    assert np.array_equal(pts, wp[order])                     # bit for bit
    assert sum(len(p[2]) for p in parts) == len(wt)             # every triangle is in exactly one slab
    assert np.array_equal(canon(keys, tris), canon(wk, wt))
    # the work rank 0 does follows the boundary, not the volume
    # (four layers of cells per boundary: at most 4 x the fullest layer of the surface, with room for welded-away triangles)
    per_layer = np.bincount(np.floor(wp[wt].min(axis=1)[:, 0]).astype(np.int64), minlength=A.shape[0]) if len(wt) else np.zeros(1)
    assert sum(L["n_own_lower"] + L["n_upper_copies"] for L in lists) <= 3 * (world - 1) * int(per_layer.max()) + 64
    # ... and what reaches rank 0 is a list of components, not of triangles
    assert stats["pairs"] <= stats["nodes"] ** 2 and stats["nodes"] <= 2 * sum(len(L["cand_label"]) for L in lists)


def test_one_slab_is_the_undivided_volume_and_clean_can_be_switched_off():
    """world = 1: no neighbours, no lists, the local decisions stand; clean=False (the reference's extract_surface_geometry(clean=False))
    goes through the sharded path as well"""
    from contourist_amd import _ffi, distributed
    A, value = fields()["balls"]
    for clean in (True, False):
        whole = _ffi.Context(0)
        whole.upload_grid(A)
        whole.extract3d(value, _ffi.CX_DIAG_CPYTHON310)
        post = whole.postprocess3d(0 if clean else 1)
        wp, wt = whole.download_level1(post)
        wk = whole.download_level1_keys(post).astype(np.int64)
        for world in (1, 3):
            ctxs = [_ffi.Context(0) for _ in range(world)]
            lists = []
            for r in range(world):
                lay = distributed.shard_layout(A.shape[0], world, r)
                lists.append(distributed.shard_local(ctxs[r], np.ascontiguousarray(A[lay["e0"]:lay["e1"]]), lay, value, A.shape, clean=clean))
            small = []
            for r in range(world):
                if r == 0:
                    pairs, unmatched = np.zeros((0, 2), dtype=np.int64), 0
                else:
                    import torch
                    pairs, unmatched = distributed.pair_labels(*[torch.from_numpy(np.asarray(x)) for x in (
                        lists[r]["own1"][0], lists[r]["own1"][1], lists[r - 1]["copy4"][0], lists[r - 1]["copy4"][1])])
                small.append(distributed.shard_small(lists[r], pairs, unmatched))
            answers, stats = distributed.merge_shard_components(small)
            assert stats["unmatched"] == 0
            parts = []
            for r in range(world):
                out = distributed.shard_finish(ctxs[r], lists[r], answers[r])
                parts.append((out["keys"], out["points"], out["triangles"]))
            keys, pts, tris = distributed.assemble_level1(parts)
            order = np.argsort(wk)
            assert np.array_equal(keys, wk[order]) and np.array_equal(pts, wp[order]), (clean, world)
            assert np.array_equal(canon(keys, tris), canon(wk, wt)), (clean, world)


def test_sharded_level1_at_full_size():
    """BASELINE's 512^3 bench field in 8 slabs of 64 planes (the ranks played one after the other on the one GPU): the union of
    the 8 parts is the undivided volume's Level-1 mesh, bit for bit; what goes through rank 0 is a few per cent of the mesh"""
    torch = pytest.importorskip("torch")
    import time
    from contourist_amd import _ffi, distributed, synthetic
    dev = torch.device("cuda", 0)
    n, world = 512, 8
    A = synthetic.smooth_noise_torch((n, n, n), 1235, 1400, dev)    # the bench's field (bench.py defaults)
    whole = _ffi.Context(0)
    whole.adopt_device_grid(A.data_ptr(), (n, n, n), keepalive=A)
    whole.extract3d(0.0, _ffi.CX_DIAG_CPYTHON310)
    post = whole.postprocess3d(0)
    wp, wt = whole.download_level1(post)
    wk = whole.download_level1_keys(post).astype(np.int64)
    del whole
    t0 = time.perf_counter()
    parts, lists, stats = play_ranks(A, 0.0, world)
    all_ms = (time.perf_counter() - t0) * 1e3
    assert stats["unmatched"] == 0
    boundary = sum(L["n_own_lower"] + L["n_upper_copies"] for L in lists)
    print("512^3 in 8 slabs: %.0f ms for the 8 ranks one after the other (first calls, downloads included); %d boundary triangles of %d stay "
          "between neighbours, rank 0 sees %s" % (all_ms, boundary, len(wt), stats))
    assert sum(len(p[2]) for p in parts) == len(wt) and boundary < 0.04 * len(wt)
    keys, pts, tris = distributed.assemble_level1(parts)
    order = np.argsort(wk)
    assert np.array_equal(keys, wk[order]) and np.array_equal(pts, wp[order])
    assert np.array_equal(canon(keys, tris), canon(wk, wt))
