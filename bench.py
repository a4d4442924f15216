#!/usr/bin/env python3
"""bench.py -- Mvoxels/s of Level-0 isosurface extraction (marching tetrahedra) on MI355X.

One "step" = one pass of the hot path over one volume resident in HBM: (N>1: one-plane halo
exchange over RCCL) -> classify/interpolate kernel -> triangle emit kernel, leaving the indexed
mesh (vertex records + index triples) in HBM.  Workload (BASELINE.json configs[2] shape at N=1):
512^3 fp32 smooth-noise field, single isovalue 0; with N GPUs every rank owns one such 512^3
slab of a (N*512) x 512 x 512 volume partitioned along array axis 0 ("z-slab"), weak scaling.

Prints ONE JSON line (rank 0).  `value` = samples of all ranks / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="samples per axis of one rank's slab")
    ap.add_argument("--passes", type=int, default=1400, help="[1,2,1]/4 smoothing passes of the noise field")
    ap.add_argument("--value", type=float, default=0.0)
    ap.add_argument("--rotate", type=int, default=0, help="number of distinct grids cycled (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the shape-agnostic classify kernel")
    ap.add_argument("--strong", action="store_true",
                    help="ONE size^3 volume split into N slabs (SURVEY config 3) instead of one size^3 slab per GPU")
    return ap.parse_args()


def cpu_baseline(field_host, value, budget_s=20.0):
    """time the C oracle (1 thread) on a bounded sub-volume of the same field"""
    import numpy as np
    from oracle import level0
    n0 = field_host.shape[0]
    # calibrate on 16 planes, then size the sample for ~budget_s
    t0 = time.time()
    level0.march3d(np.ascontiguousarray(field_host[:16]), value, diag_mode=1)
    per_plane = (time.time() - t0) / 15.0
    planes = int(max(16, min(n0, budget_s / max(per_plane, 1e-9))))
    sub = np.ascontiguousarray(field_host[:planes])
    t0 = time.time()
    O = level0.march3d(sub, value, diag_mode=1)
    dt = time.time() - t0
    return {"value": sub.size / dt / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "sample": "planes 0:%d of the %s field (%d samples, %d triangles) in %.1f s, oracle/march_oracle.c single thread"
                      % (planes, "x".join(str(n) for n in field_host.shape), sub.size, len(O["tris"]), dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from contourist_amd import _ffi, synthetic
    from contourist_amd import distributed as cxdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = max(torch.cuda.device_count(), 1)
    device_index = local_rank % ndev            # one rank per GPU; ranks only share a GPU in the 1-GPU rehearsal
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if distributed:
        backend = os.environ.get("BENCH_BACKEND", "nccl")     # "gloo" only to rehearse the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n = args.size
    nrot = args.rotate or (1 if 4 * n ** 3 > 300e6 else max(5, int(1.5e9 // (4 * n ** 3))))
    overlap_halo = distributed and os.environ.get("BENCH_SYNC_HALO", "0") != "1"
    if overlap_halo:
        nrot = max(nrot, 2)      # the halo of the next volume is exchanged while the current one is extracted

    # weak scaling (default): every rank owns `n` planes of its own field; strong scaling: the ranks own the
    # axis-0 slabs of ONE n^3 field.  Either way +1 halo plane from the upper neighbour, except on the last rank.
    has_upper = distributed and rank + 1 < world
    i0, i1 = cxdist.slab_bounds(n, world, rank) if args.strong else (0, n)
    n_own = i1 - i0
    slabs = []
    for r in range(nrot):
        if args.strong:
            own = synthetic.smooth_noise_torch((n, n, n), 1235 + r, args.passes, dev)[i0:i1]
        else:
            own = synthetic.smooth_noise_torch((n, n, n), 1235 + 97 * rank + r, args.passes, dev)
        buf = torch.empty((n_own + (1 if has_upper else 0), n, n), dtype=torch.float32, device=dev)
        buf[:n_own] = own
        del own
        slabs.append(buf)
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream()
    ctx = _ffi.Context(device_index, stream=stream.cuda_stream)
    ctx.set_origin(i0 if args.strong else rank * n, 0, 0)
    flags = _ffi.CX_DIAG_CPYTHON310 | (_ffi.CX_KERNEL_GENERIC if args.generic else 0)

    def halo_exchange(buf):
        """lower plane of rank r+1 -> halo plane of rank r (RCCL send/recv over xGMI)"""
        cxdist.exchange_halo(buf, n_own, rank, world, dist)

    pending = {}

    def step(i):
        buf = slabs[i % nrot]
        if overlap_halo:
            # one process per GPU; the only exchange of the path is the 1-plane halo.  The exchange for volume i+1
            # (another buffer) is posted before volume i is extracted and runs on RCCL's stream meanwhile.
            if i not in pending:
                pending[i] = cxdist.HaloExchange(buf, n_own, rank, world, dist)
            pending.pop(i).finish()
            pending[i + 1] = cxdist.HaloExchange(slabs[(i + 1) % nrot], n_own, rank, world, dist)
        elif distributed:
            halo_exchange(buf)
        ctx.adopt_device_grid(buf.data_ptr(), tuple(buf.shape), keepalive=buf)
        ctx.extract3d_async(args.value, flags)

    # size the output buffers once (synchronous extract grows them as needed)
    counts = None
    for r in range(nrot):
        if distributed:
            halo_exchange(slabs[r])
        ctx.adopt_device_grid(slabs[r].data_ptr(), tuple(slabs[r].shape), keepalive=slabs[r])
        c = ctx.extract3d(args.value, flags)
        counts = c if counts is None else {k: max(counts[k], c[k]) for k in c}
    ctx.reserve(int(counts["n_cells"] * 1.05) + 1024, int(counts["n_vertices"] * 1.05) + 1024,
                int(counts["n_triangles"] * 1.05) + 1024)

    for i in range(args.warmup):
        step(i)
    ctx.timing_enable(True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    for h in pending.values():        # the exchange posted for the volume after the last one
        h.finish()
    pending.clear()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = ctx.timing_read()
    final = ctx.counts()           # also verifies that the last extract fitted its buffers
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_rank = n_own * n * n
    total_samples = n ** 3 if args.strong else world * n ** 3
    if rank == 0:
        nt = max(timing["n"], 1)
        k1_ms = timing["classify_ms"] / nt
        k2_ms = timing["emit_ms"] / nt
        alg_bytes = 4.0 * slabs[0].numel()             # 4 B per input sample, read once
        # per-kernel durations (HIP events on the extraction stream, inside the timed region) and the
        # algorithmic bytes of each: stream = 4 B per sample read; vertex stage = 16 B per vertex record +
        # 16 B per cell record written; triangle stage = 12 B per triangle written
        kernels = [
            {"name": "cx_k_stream", "ms": timing["stream_ms"] / nt, "alg_bytes": alg_bytes},
            {"name": "cx_k_scan_waves+cx_k_list_batches", "ms": timing["scan_ms"] / nt, "alg_bytes": 0.0},
            {"name": "cx_k_emit_vertices", "ms": timing["cells_ms"] / nt,
             "alg_bytes": 16.0 * final["n_vertices"] + 16.0 * final["n_cells"]},
            {"name": "cx_k_emit_triangles", "ms": k2_ms, "alg_bytes": 12.0 * final["n_triangles"]},
        ]
        if args.generic:
            kernels = [{"name": "cx_k_classify_generic", "ms": k1_ms, "alg_bytes": alg_bytes}, kernels[3]]
        for kk in kernels:
            kk["GBps"] = kk["alg_bytes"] / (kk["ms"] * 1e-3) / 1e9 if kk["ms"] > 0 else 0.0
            kk["frac"] = kk["GBps"] / HBM_PEAK_GBS
        dom = max(kernels, key=lambda kk: kk["ms"])
        achieved = dom["GBps"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("%s_%d" % (dom["name"], n))
            except Exception:
                traffic = None
        out = {
            "metric": "Mvoxels/s isosurface extraction on 512^3 fp32 grid" if n == 512 else "Mvoxels/s isosurface extraction on %d^3 fp32 grid" % n,
            "value": total_samples * args.steps / elapsed / 1e6,
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%dx%d fp32 smooth-noise %s ([1,2,1]/4 x %d passes, closed interior), isovalue %g, marching tetrahedra Level-0 (classify+interpolate+emit indexed mesh)"
                            % (n, n, n, "volume split over the GPUs" if args.strong else "slab per GPU", args.passes, args.value),
                "partition": ("one volume in axis-0 slabs, 1-plane halo over RCCL" if args.strong else
                              "one slab per GPU (axis 0), 1-plane halo over RCCL") if distributed else "single GPU",
                "halo_exchange": ("overlapped with the previous volume's extraction" if overlap_halo else "in line") if distributed else None,
                "active_voxel_fraction": final["n_border_voxels"] / float((n - 1) ** 3),
                "vertices": final["n_vertices"], "triangles": final["n_triangles"],
                "grids_rotated": nrot,
                "classify_kernel": "generic" if args.generic else "auto",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": dom["name"] + " (longest of the Level-0 kernels)", "kernel_ms": dom["ms"],
                "kernels": kernels,
                "level0_ms": k1_ms + k2_ms,
                "level0_frac": alg_bytes / ((k1_ms + k2_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS if k1_ms + k2_ms > 0 else 0.0,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            host = slabs[0].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(host, args.value)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
