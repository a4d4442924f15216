"""CPU, 2 and 3 processes over gloo: the slab partition, one-plane halo exchange, ownership rule
and global-id assembly of contourist_amd.distributed reproduce the single-volume result exactly.
(The local march is the oracle here because this host has no GPU; on the GPU box the same
plumbing runs with the HIP extractor, tests/test_gpu_distributed.py.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def field():
    rng = np.random.RandomState(42)
    A = rng.standard_normal((23, 12, 16))
    for _ in range(2):
        for ax in range(3):
            A = 0.25 * np.roll(A, 1, ax) + 0.5 * A + 0.25 * np.roll(A, -1, ax)
    return (A / A.std()).astype(np.float32)


def oracle_extract(local, value, origin=(0, 0, 0)):
    from oracle import level0
    O = level0.march3d(local, value, diag_mode=1, origin=origin)
    keys = level0.edge_keys_from_pairs(O["pairs"], local.shape)
    return O["xyz"], keys, O["tris"]


def worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    A = field()
    i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
    res = cd.extract_slabs(A[i0:i1], 0.1, rank, world, oracle_extract, A.shape, dist=dist)
    if rank == 0:
        keys, xyz, tris = res
        np.savez(os.path.join(outdir, "out.npz"), keys=keys, xyz=xyz, tris=tris)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_reproduce_single_volume(world, tmp_path):
    import torch.multiprocessing as mp
    from oracle import level0
    port = free_port()
    mp.spawn(worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "out.npz"))
    A = field()
    xyz, keys, tris = oracle_extract(A, 0.1)
    ref = level0.canonical_level0(keys, xyz, tris)
    out = level0.canonical_level0(got["keys"], got["xyz"], got["tris"])
    assert np.array_equal(ref[0], out[0])
    assert np.allclose(ref[1], out[1], rtol=0, atol=1e-12)      # (i_local + t) + i0 vs (i_local + i0) + t: last-bit association
    assert np.array_equal(ref[2], out[2])


def test_slab_bounds_cover():
    from contourist_amd import distributed as cd
    for n0 in (5, 64, 513):
        for world in (1, 2, 3, 8):
            b = [cd.slab_bounds(n0, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n0
            assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))


def overlap_worker(rank, world, port, outdir):
    """two volumes in rotation: the halo of volume i+1 is exchanged (HaloExchange.start) while volume i is in use"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n_own, has_upper = 4, rank + 1 < world
    vols = []
    for r in range(2):
        buf = torch.full((n_own + (1 if has_upper else 0), 3, 5), -1.0)
        buf[:n_own] = torch.arange(n_own * 15, dtype=torch.float32).reshape(n_own, 3, 5) + 1000.0 * rank + 100000.0 * r
        vols.append(buf)
    pending = {}
    ok = True
    for i in range(6):
        buf = vols[i % 2]
        if i not in pending:
            pending[i] = cd.HaloExchange(buf, n_own, rank, world, dist)
        pending.pop(i).finish()
        pending[i + 1] = cd.HaloExchange(vols[(i + 1) % 2], n_own, rank, world, dist)
        if has_upper:     # the halo plane is the upper neighbour's first plane of the same volume
            want = torch.arange(15, dtype=torch.float32).reshape(3, 5) + 1000.0 * (rank + 1) + 100000.0 * (i % 2)
            ok = ok and bool(torch.equal(buf[n_own], want))
    for h in pending.values():
        h.finish()
    with open(os.path.join(outdir, "ok%d" % rank), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_halo_exchange(tmp_path):
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(overlap_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert open(os.path.join(str(tmp_path), "ok%d" % rank)).read() == "1"


def oracle_extract_global(local, value, origin=(0, 0, 0)):
    """like oracle_extract, but with the float64 points in the coordinates of the WHOLE volume, interpolated from
    the global lattice points (what cx_level0_points_f64 returns on the device): bit for bit what the undivided
    volume yields"""
    from oracle import level0
    O = level0.march3d(local, value, diag_mode=1, origin=origin)
    keys = level0.edge_keys_from_pairs(O["pairs"], local.shape)
    low, high = O["pairs"][:, :3].astype(np.int64), O["pairs"][:, 3:].astype(np.int64)
    flow = local[tuple(low.T)].astype(np.float64)
    fhigh = local[tuple(high.T)].astype(np.float64)
    den = 1.0 * (fhigh - flow)
    ratio = np.where(np.abs(den) <= 1e-8, 0.5, (value - flow) / np.where(den == 0, 1.0, den))
    org = np.array(origin, dtype=np.float64)
    xyz = (low + org) + ratio[:, None] * ((high + org) - (low + org))
    return xyz, keys, O["tris"]


oracle_extract_global.global_points = True


def worker_level1(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    A = field()
    i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
    res = cd.extract_slabs(A[i0:i1], 0.1, rank, world, oracle_extract_global, A.shape, dist=dist)
    if rank == 0:
        keys, xyz, tris = res
        np.savez(os.path.join(outdir, "l1in.npz"), keys=keys, xyz=xyz, tris=tris)
    dist.barrier()
    dist.destroy_process_group()


def test_level1_input_of_slabs_is_the_undivided_volume(tmp_path):
    """world_size 2 over gloo: the mesh rank 0 hands to the Level-1 post-pass (distributed.level1_slabs) -- vertices
    in ascending global edge-id order, float64 points, triangles -- is bit for bit the Level-0 mesh of the undivided
    volume, so the post-pass (here the oracle's) gives the same Level-1 mesh"""
    import torch.multiprocessing as mp
    from oracle import level0, postpass
    mp.spawn(worker_level1, args=(2, free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "l1in.npz"))
    A = field()
    xyz, keys, tris = oracle_extract_global(A, 0.1)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(got["keys"], keys[order]) and np.all(np.diff(got["keys"]) > 0)
    assert np.array_equal(got["xyz"], xyz[order])                       # exactly, not just close
    ref = level0.canonical_level0(keys, xyz, tris)
    out = level0.canonical_level0(got["keys"], got["xyz"], got["tris"])
    assert np.array_equal(ref[2], out[2])
    corner = np.array(A.shape) - 1
    a = postpass.level1_from_level0(keys, xyz, tris, corner)
    b = postpass.level1_from_level0(got["keys"], got["xyz"], got["tris"], corner)
    assert a["n_after_weld"] == b["n_after_weld"] and a["n_after_tiny"] == b["n_after_tiny"]
    assert np.array_equal(postpass.canonical_level1(a["grid_points"], a["triangles"], corner),
                          postpass.canonical_level1(b["grid_points"], b["triangles"], corner))


LEVELS = (-0.6, 0.1, 0.45, 1.1)


def oracle_extract_levels(local, values, origin=(0, 0, 0)):
    return [oracle_extract(local, v, origin) for v in values]


def levels_worker(rank, world, port, outdir):
    """BASELINE config 5 over ranks: each rank its slab x ALL isovalues, ONE halo exchange for all of them"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    A = field()
    i0, i1 = cd.slab_bounds(A.shape[0], world, rank)
    res = cd.extract_slabs_levels(A[i0:i1], LEVELS, rank, world, oracle_extract_levels, A.shape, dist=dist)
    if rank == 0:
        assert len(res) == len(LEVELS)
        np.savez(os.path.join(outdir, "levels.npz"), **{"%s%d" % (k, l): a for l, m in enumerate(res) for k, a in zip(("keys", "xyz", "tris"), m)})
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_levels_reproduce_single_volume(world, tmp_path):
    """every level of the slab-partitioned multi-level extraction == the single-volume extraction of that isovalue"""
    import torch.multiprocessing as mp
    from oracle import level0
    port = free_port()
    mp.spawn(levels_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "levels.npz"))
    A = field()
    for l, v in enumerate(LEVELS):
        xyz, keys, tris = oracle_extract(A, v)
        ref = level0.canonical_level0(keys, xyz, tris)
        out = level0.canonical_level0(got["keys%d" % l], got["xyz%d" % l], got["tris%d" % l])
        assert len(ref[0]) > 50
        assert np.array_equal(ref[0], out[0])
        assert np.allclose(ref[1], out[1], rtol=0, atol=1e-12)
        assert np.array_equal(ref[2], out[2])
