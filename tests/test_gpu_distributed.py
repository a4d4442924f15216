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
