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


# ---- pre-flight of the N > 1 paths that cannot run on a one-GPU box (round 4) -------------------------------------------------
class _FakeCtx(object):
    """stands in for _ffi.Context in own_communicators: records what was called; can be told to fail in one place"""
    handle = 1

    def __init__(self, fail=None):
        self.fail, self.calls, self.comm = fail, [], None

    def rccl_available(self):
        self.calls.append("available")
        return self.fail != "available"

    def rccl_unique_id(self):
        self.calls.append("unique_id")
        if self.fail == "unique_id":
            raise RuntimeError("no id")
        return np.arange(128, dtype=np.uint8)

    def rccl_comm_init(self, uid, rank, world):
        self.calls.append("init")
        assert np.array_equal(np.asarray(uid), np.arange(128, dtype=np.uint8))     # the id rank 0 made reached this rank
        if self.fail == "init":
            raise RuntimeError("ncclCommInitRank failed")
        self.comm = "own"

    def rccl_comm_share(self, owner):
        self.calls.append("share")
        self.comm = "shared"

    def rccl_comm_destroy(self):
        self.calls.append("destroy")
        self.comm = None


def own_comm_worker(rank, world, port, outdir, fail_rank, fail_where):
    sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ctxs = [_FakeCtx(fail_where if rank == fail_rank else None) for _ in range(3)]
    ok = cd.own_communicators(ctxs, rank, world, dist)
    json.dump({"ok": bool(ok), "calls": [c.calls for c in ctxs], "comm": [c.comm for c in ctxs]}, open(os.path.join(outdir, "r%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank,fail_where", [(-1, None), (1, "available"), (0, "unique_id"), (1, "init"), (0, "init")])
def test_own_communicators_all_ranks_fall_back_together(fail_rank, fail_where, tmp_path):
    """ADVICE round 3 / VERDICT item 6: whatever fails on ONE rank (no RCCL in the process, no id, ncclCommInitRank), every rank
    learns of it and returns False -- and a rank that cannot enter the collective init says so BEFORE anybody enters it."""
    import json
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(own_comm_worker, args=(world, free_port(), str(tmp_path), fail_rank, fail_where), nprocs=world, join=True)
    R = [json.load(open(os.path.join(str(tmp_path), "r%d.json" % r))) for r in range(world)]
    want = fail_rank < 0
    assert all(r["ok"] == want for r in R), R
    for r in R:
        if want:
            assert r["comm"] == ["own", "shared", "shared"]          # ONE communicator per rank, shared by its contexts
            assert r["calls"][0].count("init") == 1 and "init" not in r["calls"][1]
        else:
            assert r["comm"] == [None, None, None]                   # nobody keeps half a setup
        if fail_where in ("available", "unique_id"):
            assert all("init" not in c for c in r["calls"]), "a rank entered the collective init although another had said it could not"


def shard_error_worker(rank, world, port, outdir, mode):
    """level1_slabs_sharded with the GPU parts replaced: what happens to the other ranks when ONE rank's pairing raises, when rank 0's
    merge raises, or when boundary triangles do not match"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from contourist_amd import distributed as cd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n0 = 24
    lay = cd.shard_layout(n0, world, rank)

    class Ctx(object):
        device = 0

    def fake_local(ctx, local, layout, value, global_shape, clean=True, torch_device=None):
        e = np.zeros(0)
        return dict(cand_label=e.astype(np.uint32), cand_x=e, cand_vertex_key=e.astype(np.int64), cand_nx=e, cand_negative=e.astype(np.uint8),
                    cand_has=e.astype(np.uint8), own1=(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32)),
                    copy4=(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32)), n_own_lower=0, n_upper_copies=0,
                    counts={}, key_offset=0)
    cd.shard_local = fake_local
    cd.exchange_boundary_lists = lambda L, r, w, d, dev: None if r == 0 else (torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32))
    if mode == "pair_raises":
        def pl(*a):
            raise ValueError("pairing broke on rank 1")
        cd.pair_labels = pl if rank == 1 else (lambda *a: (np.zeros((0, 2), np.int64), 0))
    elif mode == "unmatched":
        cd.pair_labels = lambda *a: (np.zeros((0, 2), np.int64), 7 if rank == 1 else 0)
    else:
        cd.pair_labels = lambda *a: (np.zeros((0, 2), np.int64), 0)
    if mode == "merge_raises":
        def mg(lists):
            raise KeyError("merge broke")
        cd.merge_shard_components = mg
    cd.shard_finish = lambda ctx, L, answer, download=True: dict(counts={})
    import torch as _t
    _t.device = (lambda *a, **k: "cpu") if False else _t.device
    own = torch.zeros((lay["i1"] - lay["i0"], 4, 4), dtype=torch.float32)
    msg = "ok"
    try:
        cd.level1_slabs_sharded(own, 0.0, rank, world, (n0, 4, 4), dist=dist, context=Ctx(), download=False)
    except Exception as e:      # noqa: BLE001
        msg = "%s: %s" % (type(e).__name__, e)
    open(os.path.join(outdir, "r%d.txt" % rank), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["fine", "pair_raises", "unmatched", "merge_raises"])
def test_sharded_level1_errors_reach_every_rank(mode, tmp_path):
    """an exception on one rank between the agreed flag and the scatter, rank 0's merge raising, or `unmatched`
    boundary triangles must end the call on EVERY rank with an error -- nobody waits in a collective, nobody returns a mesh whose
    winding may disagree across slabs."""
    import torch.multiprocessing as mp
    world = 3
    mp.spawn(shard_error_worker, args=(world, free_port(), str(tmp_path), mode), nprocs=world, join=True)
    msgs = [open(os.path.join(str(tmp_path), "r%d.txt" % r)).read() for r in range(world)]
    if mode == "fine":
        assert msgs == ["ok"] * world, msgs
    else:
        assert all(m.startswith("RuntimeError: sharded Level 1 failed") for m in msgs), msgs
        key = {"pair_raises": "pairing broke on rank 1", "unmatched": "7 boundary triangles", "merge_raises": "merge broke"}[mode]
        assert all(key in m for m in msgs), msgs


def test_sharded_level1_thin_slab_raises_everywhere_before_any_exchange():
    """a slab thinner than the layers it must hand over: every rank raises from the same arithmetic, no communication needed"""
    import torch
    from contourist_amd import distributed as cd
    for rank in range(8):
        i0, i1 = cd.slab_bounds(16, 8, rank)        # 2 planes per rank < SHARD_LAYERS + 1
        with pytest.raises(ValueError, match="fewer than"):
            cd.level1_slabs_sharded(torch.zeros((i1 - i0, 4, 4)), 0.0, rank, 8, (16, 4, 4), dist=object())
