"""GPU: the slab path with the HIP extractor.  (a) two slabs marched one after the other in one
process; (b) two processes over gloo sharing the one GPU of the test box (RCCL needs one device
per rank; the driver's multi-GPU bench exercises that).  Both must reproduce the undivided
volume exactly, including the CPython-order quad diagonals (hash of GLOBAL lattice coordinates)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def field():
    rng = np.random.RandomState(7)
    A = rng.standard_normal((41, 24, 32))
    for _ in range(2):
        for ax in range(3):
            A = 0.25 * np.roll(A, 1, ax) + 0.5 * A + 0.25 * np.roll(A, -1, ax)
    return (A / A.std()).astype(np.float32)


def whole(A, value):
    from contourist_amd import distributed as cd
    from oracle import level0
    run = cd.hip_extract(0)
    xyz, keys, tris = run(A, value)
    return level0.canonical_level0(keys.astype(np.int64), xyz, tris.astype(np.int64))


def test_two_slabs_in_one_process():
    from contourist_amd import distributed as cd
    from oracle import level0
    A = field()
    ref = whole(A, 0.2)
    run = cd.hip_extract(0)
    parts = []
    world = 3
    for rank in range(world):
        i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
        has_halo = rank + 1 < world
        local = np.ascontiguousarray(A[i0:i1 + (1 if has_halo else 0)])
        xyz, keys, tris = run(local, 0.2, (i0, 0, 0))
        parts.append(cd.local_to_global(xyz, keys, tris, local.shape, i0, i1 - i0, has_halo))
    keys, xyz, tris = cd.assemble(parts)
    out = level0.canonical_level0(keys, xyz, tris)
    assert np.array_equal(ref[0], out[0]) and np.array_equal(ref[2], out[2])
    assert np.allclose(ref[1], out[1], rtol=1e-6, atol=1e-6)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    A = field()
    i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
    res = cd.extract_slabs(A[i0:i1], 0.2, rank, world, cd.hip_extract(0), A.shape, dist=dist)
    if rank == 0:
        np.savez(os.path.join(outdir, "out.npz"), keys=res[0], xyz=res[1], tris=res[2])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_hip(tmp_path):
    import torch.multiprocessing as mp
    from oracle import level0
    A = field()
    ref = whole(A, 0.2)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "out.npz"))
    out = level0.canonical_level0(got["keys"], got["xyz"], got["tris"])
    assert np.array_equal(ref[0], out[0]) and np.array_equal(ref[2], out[2])
    assert np.allclose(ref[1], out[1], rtol=1e-6, atol=1e-6)


LEVELS = (-0.5, 0.2, 0.9)


def _levels_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    A = field()
    i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
    res = cd.extract_slabs_levels(A[i0:i1], LEVELS, rank, world, cd.hip_extract_levels(0), A.shape, dist=dist)
    if rank == 0:
        np.savez(os.path.join(outdir, "levels.npz"), **{"%s%d" % (k, l): a for l, m in enumerate(res) for k, a in zip(("keys", "xyz", "tris"), m)})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_hip_levels(tmp_path):
    """BASELINE config 5 over ranks with the HIP extractor: every rank marches its slab for ALL isovalues in one
    cx_extract3d_levels call; each assembled level == the undivided volume at that isovalue (CPython-order diagonals hash
    global lattice coordinates: cx_set_origin)"""
    import torch.multiprocessing as mp
    from oracle import level0
    A = field()
    mp.spawn(_levels_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "levels.npz"))
    for l, v in enumerate(LEVELS):
        ref = whole(A, v)
        out = level0.canonical_level0(got["keys%d" % l], got["xyz%d" % l], got["tris%d" % l])
        assert len(ref[0]) > 100
        assert np.array_equal(ref[0], out[0]) and np.array_equal(ref[2], out[2])
        assert np.allclose(ref[1], out[1], rtol=1e-6, atol=1e-6)


def same_level1(A, pts0, tris0, pts1, tris1):
    """two Level-1 meshes are the same surface: the same multiset of float64 points, and the same oriented triangles
    written as triples of weld-bucket ids (the vertex numbering differs: march order on a single GPU, edge-id order
    for an assembled mesh)"""
    from oracle import postpass
    corner = np.array(A.shape) - 1

    def rows(P):
        return P[np.lexsort((P[:, 2], P[:, 1], P[:, 0]))]
    assert pts0.shape == pts1.shape and np.array_equal(rows(pts0), rows(pts1))
    a = postpass.canonical_level1(pts0, tris0, corner)
    b = postpass.canonical_level1(pts1, tris1, corner)
    assert a.shape == b.shape and np.array_equal(a, b)


def whole_level1(A, value):
    from contourist_amd import _ffi
    ctx = _ffi.Context(0)
    ctx.upload_grid(A)
    ctx.extract3d(value, 1)
    post = ctx.postprocess3d(0)
    pts, tris = ctx.download_level1(post)
    return pts, tris, post


def test_level1_of_slabs_in_one_process():
    """Level 1 of a mesh assembled from three slabs == Level 1 of the undivided volume, bit for bit: same counts
    after every stage, same float64 points, same oriented triangles"""
    from contourist_amd import distributed as cd
    A = field()
    pts0, tris0, post0 = whole_level1(A, 0.2)
    run = cd.hip_extract(0, float64_points=True)
    parts = []
    world = 3
    for rank in range(world):
        i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
        has_halo = rank + 1 < world
        local = np.ascontiguousarray(A[i0:i1 + (1 if has_halo else 0)])
        xyz, keys, tris = run(local, 0.2, (i0, 0, 0))
        assert xyz.dtype == np.float64
        parts.append(cd.local_to_global(xyz, keys, tris, local.shape, i0, i1 - i0, has_halo, xyz_is_global=True))
    keys, xyz, tris = cd.assemble(parts)
    post = run.context.postprocess3d_mesh(xyz, tris, [n - 1 for n in A.shape])
    pts1, tris1 = run.context.download_level1(post)
    assert post == post0
    same_level1(A, pts0, tris0, pts1, tris1)


def _worker_level1(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    A = field()
    i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
    res = cd.level1_slabs(A[i0:i1], 0.2, rank, world, A.shape, device=0, dist=dist)
    if rank == 0:
        np.savez(os.path.join(outdir, "l1.npz"), pts=res[0], tris=res[1], counts=np.array([res[2][k] for k in sorted(res[2])]))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_level1_two_ranks_gloo_hip(tmp_path):
    import torch.multiprocessing as mp
    A = field()
    pts0, tris0, post0 = whole_level1(A, 0.2)
    mp.spawn(_worker_level1, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "l1.npz"))
    assert got["counts"].tolist() == [post0[k] for k in sorted(post0)]
    same_level1(A, pts0, tris0, got["pts"], got["tris"])


def _worker_level1_sharded(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    A = field()
    i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
    res = cd.level1_slabs_sharded(A[i0:i1], 0.2, rank, world, A.shape, device=0, dist=dist)
    assert (res["stats"] is not None) == (rank == 0)
    np.savez(os.path.join(outdir, "shard%d.npz" % rank), keys=res["keys"], pts=res["points"], tris=res["triangles"],
             boundary=np.array([res["boundary"]["triangles"], res["boundary"]["components"]]),
             unmatched=np.array([res["stats"]["unmatched"] if rank == 0 else 0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_level1_sharded_ranks_gloo_hip(tmp_path, world):
    """Level 1 WITHOUT gathering the mesh (distributed.level1_slabs_sharded): every rank post-processes its slab (+ two layers
    of its neighbours' cells) on the GPU, the ranks exchange boundary labels and start-triangle candidates through rank 0 only;
    the union of the ranks' parts is the undivided volume's Level-1 mesh -- same vertices by edge id with bit-identical
    coordinates, same oriented triangles, every triangle in exactly one part"""
    import torch.multiprocessing as mp
    from contourist_amd import _ffi, distributed as cd
    from test_gpu_sharded_level1 import canon
    A = field()
    ctx = _ffi.Context(0)
    ctx.upload_grid(A)
    ctx.extract3d(0.2, 1)
    post = ctx.postprocess3d(0)
    wp, wt = ctx.download_level1(post)
    wk = ctx.download_level1_keys(post).astype(np.int64)
    mp.spawn(_worker_level1_sharded, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "shard%d.npz" % r)) for r in range(world)]
    assert int(got[0]["unmatched"][0]) == 0
    keys, pts, tris = cd.assemble_level1([(g["keys"], g["pts"], g["tris"]) for g in got])
    order = np.argsort(wk)
    assert len(wt) > 1000 and sum(len(g["tris"]) for g in got) == len(wt)
    assert np.array_equal(keys, wk[order]) and np.array_equal(pts, wp[order])
    assert np.array_equal(canon(keys, tris), canon(wk, wt))
    assert all(int(g["boundary"][0]) < len(wt) // 2 for g in got)


def test_postprocess_mesh_rejects_bad_indices():
    from contourist_amd import _ffi
    ctx = _ffi.Context(0)
    pts = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    with pytest.raises(_ffi.CxError):
        ctx.postprocess3d_mesh(pts, [[0, 1, 3]], [4, 4, 4])
    with pytest.raises(_ffi.CxError):
        ctx.postprocess3d_mesh(pts, [[0, 1, 2]], [4, 0, 4])
    post = ctx.postprocess3d_mesh(pts, [[0, 1, 2]], [4, 4, 4])
    assert post["n_triangles"] == 1 and post["n_vertices"] == 3
    p1, t1 = ctx.download_level1(post)
    assert sorted(t1[0].tolist()) == [0, 1, 2]


def test_halo_exchange_c_abi_single_rank_and_arguments():
    """cx_halo_exchange: one rank has nothing to exchange; bad arguments are refused before RCCL is touched.  (Two ranks need two
    GPUs -- RCCL refuses two ranks on one device -- so the send / receive pair itself is not exercised on this pool; the Python
    host's torch.distributed exchange is what tests/test_distributed_gloo.py and the bench cover.)"""
    import torch
    from contourist_amd import _ffi
    ctx = _ffi.Context(0)
    buf = torch.zeros(3 * 16, dtype=torch.float32, device="cuda:0")
    ctx.halo_exchange(None, 0, 1, buf.data_ptr(), 2, 16)                 # world 1: no-op, no communicator needed
    for args in ((None, 0, 2, buf.data_ptr(), 2, 16),                    # no communicator
                 (None, 2, 2, buf.data_ptr(), 2, 16),                    # rank out of range
                 (None, 0, 0, buf.data_ptr(), 2, 16),                    # world 0
                 (None, 0, 1, buf.data_ptr(), 0, 16)):                   # no owned plane
        with pytest.raises(_ffi.CxError):
            ctx.halo_exchange(*args)
