#!/usr/bin/env python3
"""bench.py -- Mvoxels/s of Level-0 isosurface extraction (marching tetrahedra) on MI355X.

One "step" = one pass of the hot path over one volume resident in HBM: (N>1: one-plane halo
exchange over RCCL) -> stream/classify -> scan -> vertex + triangle emit, leaving the indexed mesh
(vertex records + index triples) in HBM.

Workload: BASELINE.json's metric configuration -- ONE 512^3 fp32 smooth-noise volume, single isovalue 0.
With N GPUs the volume is split into N slabs along array axis 0 ("z-slab", SURVEY config 3), one process
per GPU, one-plane halo over RCCL: STRONG scaling (total work fixed).  `--weak` gives every rank its own
512^3 slab instead; an N>1 strong run also reports the weak figure as a second field.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run`, as a CHILD process, before anything in this process touches a GPU).

Prints ONE JSON line (rank 0).  `value` = samples of the whole job / max-over-ranks wall time.
"""
import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--size", type=int, default=512, help="samples per axis of the volume")
    ap.add_argument("--passes", type=int, default=1400, help="[1,2,1]/4 smoothing passes of the noise field")
    ap.add_argument("--value", type=float, default=0.0)
    ap.add_argument("--rotate", type=int, default=0, help="number of distinct grids cycled (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the Level-1 / API timing after the timed region (N > 1: the sharded Level 1)")
    ap.add_argument("--generic", action="store_true", help="force the shape-agnostic classify kernel")
    ap.add_argument("--weak", action="store_true", help="one size^3 slab PER GPU instead of one volume split over the GPUs")
    ap.add_argument("--strong", action="store_true", help="(default) ONE size^3 volume split into N slabs")
    ap.add_argument("--no-single-stream", action="store_true", help="skip the extra pass on one stream (unoverlapped kernel durations)")
    ap.add_argument("--levels", type=int, default=8, help="BASELINE config 5 after the timed region: this many isovalues (20..90th percentiles) of the same volume in ONE call per rank (0 = skip)")
    ap.add_argument("--no-config4", action="store_true", help="skip BASELINE config 4 after the timed region (N = 1 only: 128^3 x 64 4-D field, pentatope march, "
                    "morph triangles, the per-t isosurface stream of 64 times in one call)")
    ap.add_argument("--streams", type=int, default=0, help="extractions in flight: consecutive steps alternate between this many contexts / HIP streams "
                                                           "(0 = auto: 2; 3 from 4 ranks on, where a rank's slab is thin and every step waits for a halo plane)")
    return ap.parse_args()


def launch_ranks(args):
    """start `args.gpus` ranks as a child torch.distributed.run; this process never touches a GPU"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(field_host, value, budget_s=20.0):
    """time the C oracle (1 thread) on a bounded sub-volume of the same field; quote the real reference beside it"""
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
    out = {"value": sub.size / dt / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": "port",
           "sample": "planes 0:%d of the %s field (%d samples, %d triangles) in %.1f s, oracle/march_oracle.c single thread"
                     % (planes, "x".join(str(n) for n in field_host.shape), sub.size, len(O["tris"]), dt),
           "sample_planes": planes,
           "oracle_counts": {"n_vertices": int(len(O["pairs"])), "n_triangles": int(len(O["tris"])), "n_border_voxels": int(O["nborder_mixed"])}}
    tpath = os.path.join(ROOT, "tests", "golden", "reference_timings.json")
    if os.path.exists(tpath):
        # the reference's pure Python never travels to the GPU box: its rate was measured where the fixtures were
        # made (oracle/make_reference_timings.py); scaled to this host by the port's rate on both machines
        R = json.load(open(tpath))
        ratio = R["port_over_reference_median"]
        out["reference_python"] = {
            "value": R["reference_Mvoxels_s_median"], "unit": "Mvoxels/s", "cores": 1,
            "measured": R["where"] + "; median over %d fixtures of 16^3..33^3 samples" % len(R["fixtures"]),
            "port_over_reference": ratio,
            "reference_equivalent_here": out["value"] / ratio,
            "note": "single-threaded pure Python (Level 0 + Level 1); the port is Level 0 only",
        }
    return out


class Job(object):
    """one rank's share of a volume (or, weak: its own volume), resident in HBM"""

    def __init__(self, args, torch, dist, cxdist, synthetic, dev, rank, world, strong, nrot):
        n = args.size
        self.n, self.rank, self.world, self.dist, self.cxdist = n, rank, world, dist, cxdist
        self.has_upper = world > 1 and rank + 1 < world
        self.i0, i1 = cxdist.slab_bounds(n, world, rank) if strong else (0, n)
        self.n_own = i1 - self.i0
        self.origin0 = self.i0 if strong else rank * n
        self.slabs = []
        for r in range(nrot):
            if strong:
                own = synthetic.smooth_noise_torch((n, n, n), 1235 + r, args.passes, dev)[self.i0:i1]
            else:
                own = synthetic.smooth_noise_torch((n, n, n), 1235 + 97 * rank + r, args.passes, dev)
            buf = torch.empty((self.n_own + (1 if self.has_upper else 0), n, n), dtype=torch.float32, device=dev)
            buf[:self.n_own] = own
            del own
            self.slabs.append(buf)
        torch.cuda.synchronize()
        self.total_samples = n ** 3 if strong else world * n ** 3


def run_job(args, torch, dist, ctxs, streams, job, flags, overlap_halo, timing=True, c_halo=False):
    """size buffers, warm up, time args.steps steps; -> (elapsed max over ranks, timing dict, counts).
    ctxs / streams: consecutive steps alternate between these contexts (one HIP stream each): independent volumes, so
    the extraction of step i+1 starts while step i is still in its emit stages -- each kernel alone leaves part of the
    chip idle in its ramp and tail (thin slabs most of all)."""
    cxdist, rank, world = job.cxdist, job.rank, job.world
    distributed = world > 1
    nrot = len(job.slabs)
    ns = len(ctxs)
    for c in ctxs:
        c.set_origin(job.origin0, 0, 0)
    pending = {}

    def step(i):
        buf = job.slabs[i % nrot]
        ctx, st = ctxs[i % ns], streams[i % ns]
        if c_halo:
            # the whole step of this rank as ONE C call: adopt, halo exchange on the extraction stream with the context's own
            # communicator (context k of every rank shares communicator k; volume i goes to context i % ns on every rank), extract
            ctx.slab_step(buf.data_ptr(), job.n_own, job.n, job.n, rank, world, args.value, flags, keepalive=buf)
            return
        if overlap_halo:
            # the only exchange of the path is the 1-plane halo.  The exchange for volume i+1 (another buffer) is
            # posted before volume i is extracted and runs on RCCL's stream meanwhile.
            if i not in pending:
                with torch.cuda.stream(st):
                    pending[i] = cxdist.HaloExchange(buf, job.n_own, rank, world, dist)
            with torch.cuda.stream(st):          # the stream that extracts volume i waits for its halo
                pending.pop(i).finish()
            # (posted on the stream that last read that buffer, volume i + 1 - nrot: RCCL orders itself behind the work queued there)
            last_user = i + 1 - nrot
            with torch.cuda.stream(streams[last_user % ns] if last_user >= 0 else streams[0]):
                pending[i + 1] = cxdist.HaloExchange(job.slabs[(i + 1) % nrot], job.n_own, rank, world, dist)
        elif distributed:
            with torch.cuda.stream(st):
                cxdist.exchange_halo(buf, job.n_own, rank, world, dist)
        ctx.adopt_device_grid(buf.data_ptr(), tuple(buf.shape), keepalive=buf)
        ctx.extract3d_async(args.value, flags)

    # size the output buffers once (synchronous extract grows them as needed)
    counts = None
    for r in range(nrot):
        if distributed:
            cxdist.exchange_halo(job.slabs[r], job.n_own, rank, world, dist)
        torch.cuda.synchronize()
        for ctx in ctxs[:1]:
            ctx.adopt_device_grid(job.slabs[r].data_ptr(), tuple(job.slabs[r].shape), keepalive=job.slabs[r])
            c = ctx.extract3d(args.value, flags)
            counts = c if counts is None else {k: max(counts[k], c[k]) for k in c}
    for ctx in ctxs:
        ctx.reserve(int(counts["n_cells"] * 1.05) + 1024, int(counts["n_vertices"] * 1.05) + 1024,
                    int(counts["n_triangles"] * 1.05) + 1024)
    for i in range(args.warmup):
        step(i)
    for ctx in ctxs:
        ctx.timing_enable(timing)
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
    tm = None
    for ctx in ctxs:
        t = ctx.timing_read()
        ctx.timing_enable(False)
        tm = t if tm is None else {k: tm[k] + t[k] for k in tm}
    final = None
    for k in range(min(ns, args.steps + args.warmup)):
        final = ctxs[k].counts()           # also verifies that the last extracts fitted their buffers
    if distributed:
        dev = job.slabs[0].device
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, tm, final


def run_levels(args, torch, dist, _ffi, job, stream, device_index, flags, distributed):
    """BASELINE config 5: `args.levels` isovalues (20..90th percentiles of the field) of the resident volume, every rank its slab
    x ALL levels in one cx_extract3d_levels call (the slab is streamed once for all levels; the halo plane exchanged for the
    single-level steps serves every level).  -> dict for the line (rank 0), timed like the main region (barrier, max over ranks)."""
    import numpy as np
    L = int(args.levels)
    buf = job.slabs[0]
    dev = buf.device
    if job.rank == 0:
        sample = buf[:job.n_own].flatten()[:: max(1, buf[:job.n_own].numel() // (1 << 22))].float()
        q = torch.tensor([float(x) / 100.0 for x in np.linspace(20.0, 90.0, L)], dtype=torch.float32, device=dev)
        vals = torch.quantile(sample, q).double()
    else:
        vals = torch.zeros(L, dtype=torch.float64, device=dev)
    if distributed:
        if dist.get_backend() == "nccl":
            dist.broadcast(vals, src=0)
        else:
            v = vals.cpu()
            dist.broadcast(v, src=0)
            vals = v
    values = [float(x) for x in vals.cpu()]
    if distributed:
        job.cxdist.exchange_halo(buf, job.n_own, job.rank, job.world, dist)
    ctx = _ffi.Context(device_index, stream=stream.cuda_stream)
    try:
        ctx.set_origin(job.origin0, 0, 0)
        ctx.adopt_device_grid(buf.data_ptr(), tuple(buf.shape), keepalive=buf)
        with torch.cuda.stream(stream):
            counts = ctx.extract3d_levels(values, flags & 1)        # sizes every level's buffers
            ctx.extract3d_levels(values, flags & 1)
            reps = max(1, min(5, args.steps))
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.extract3d_levels(values, flags & 1)
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / reps
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
    finally:
        ctx.close()
    torch.cuda.empty_cache()
    return {"levels": values, "ms_all_levels": el * 1e3,
            "Mvoxel_levels_per_s": L * job.total_samples / el / 1e6,
            "Mvoxels_per_s_grid_once": job.total_samples / el / 1e6,
            "triangles_per_level_rank0": [int(c["n_triangles"]) for c in counts],
            "calls_timed": reps,
            "note": "BASELINE config 5: every rank marches its slab for ALL isovalues in one cx_extract3d_levels call (one pass over the "
                    "samples for all levels, then vertex + triangle stages per level); synchronous per call (counts come back to size the buffers); "
                    "whole-job figures, max over ranks"}


def run_config4(torch, _ffi, synthetic, dev, device_index):
    """BASELINE config 4 on this GPU (N = 1 only): 128 x 128 x 128 x 64 fp32 4-D field (two moving blobs + noise, seed 1236), the pentatope
    march (cx_extract4d), find_tetrahedra's post steps, the morph triangles, and the per-t isosurface stream: the surfaces at 64 times
    from the morph triangles in ONE call (cx_morph_eval_many).  Every figure is a warm call (buffers in place), synchronous."""
    import numpy as np
    shape = (128, 128, 128, 64)
    A = synthetic.moving_blobs_torch(shape, 1236, dev)
    ctx = _ffi.Context(device_index)
    try:
        ctx.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            return r, (time.perf_counter() - t0) / reps
        counts, t_l0 = timed(lambda: ctx.extract4d(0.5, 1), 5)
        # two extractions in flight on two contexts (cx_extract4d_async / cx_counts4d_get), as the 3-D headline has them
        ctx2 = _ffi.Context(device_index)
        try:
            ctx2.adopt_device_grid4d(A.data_ptr(), shape, keepalive=A)
            ctx2.extract4d(0.5, 1)
            pair = [ctx, ctx2]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 12
            for i in range(reps):          # `reps` extractions start and end inside the timed region
                if i >= 2:
                    pair[i & 1].counts4d()
                pair[i & 1].extract4d_async(0.5, 1)
            last = [c_.counts4d() for c_ in pair]
            torch.cuda.synchronize()
            t_l0_two = (time.perf_counter() - t0) / reps
            assert last[0] == counts and last[1] == counts
        finally:
            ctx2.close()
        post, t_post = timed(lambda: ctx.postprocess4d(100), 2)
        out = np.zeros(8, dtype=np.int64)
        _, t_morph = timed(lambda: ctx._check(ctx.lib.cx_morph_triangles(ctx.handle, out.ctypes.data)), 2)
        import struct

        def from_orderable(o):
            o = int(o) & 0xFFFFFFFFFFFFFFFF
            b = (o & 0x7FFFFFFFFFFFFFFF) if (o >> 63) else (~o & 0xFFFFFFFFFFFFFFFF)
            return struct.unpack("<d", struct.pack("<Q", b))[0]
        tmin, tmax = from_orderable(out[5]), from_orderable(out[6])
        ts = np.linspace(tmin, tmax, shape[3])
        cm, t_stream = timed(lambda: ctx.morph_eval_many(ts, download=False), 5)
        n = A.numel()
        return {"workload": "128x128x128x64 fp32, two moving blobs + noise (seed 1236), v = 0.5", "counts": counts, "post": post,
                "level0_ms": t_l0 * 1e3, "Mhypervoxels_per_s": n / t_l0 / 1e6, "hbm_frac_input_bytes": 4.0 * n / t_l0 / (HBM_PEAK_GBS * 1e9),
                "level0_two_in_flight_ms": t_l0_two * 1e3, "hbm_frac_input_bytes_two_in_flight": 4.0 * n / t_l0_two / (HBM_PEAK_GBS * 1e9),
                "postprocess_ms": t_post * 1e3, "morph_triangles_ms": t_morph * 1e3, "morph_segments": int(out[1]), "morph_triangles": int(out[2]),
                "per_t_stream": {"times": int(len(ts)), "ms": t_stream * 1e3, "triangles": int(cm[:, 1].sum()), "points": int(cm[:, 0].sum()),
                                 "Mtriangles_per_s": int(cm[:, 1].sum()) / t_stream / 1e6,
                                 "note": "cx_morph_eval_many: the surfaces at 64 equally spaced times in one call, meshes left on the device"}}
    finally:
        ctx.close()


def run_sharded_level1(args, torch, dist, cxdist, job, dev, rank, world, strong, ctx):
    """Level 1 (weld, tiny collapse, clean-up, orientation) of the bench volume WITHOUT gathering the mesh: every rank post-processes
    its own slab (distributed.level1_slabs_sharded); the figure is the slowest rank's time from its resident slab to its part of
    the oriented mesh on the device (second call: buffers exist).  A failure is reported in the line, never raised: the Level-0
    figure above does not depend on it."""
    n = args.size
    shape = (n, n, n) if strong else (world * n, n, n)
    own = job.slabs[0][:job.n_own]
    res, err = None, None
    # the two small object collectives go over a gloo group with a timeout: a rank that dies there raises on the others
    # instead of leaving them waiting (RCCL would wait for ever); the per-triangle lists travel device to device over RCCL
    obj_group = None
    if dist.get_backend() == "nccl":
        try:
            import datetime
            obj_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
        except Exception:        # noqa: BLE001 -- collective: fails or succeeds on every rank alike
            obj_group = None
    call_ms = []
    try:
        # Two untimed calls, then ONE timed call (round 3 took the faster of two to hide an outlier of 70-120 ms; round 4 traced it,
        # HIP loads a kernel's code object at its FIRST launch -- 20-110 ms each, tools/shard_time.py under
        # rocprofv3 --hip-trace -- and torch's first sort / unique costs 0.6 s; both belong to the first call of a process, and in
        # 1 280 later calls none took more than 7 ms).  Every rank makes the same three calls.
        for k in range(3):
            t_call = time.perf_counter()
            res = cxdist.level1_slabs_sharded(own, args.value, rank, world, shape, dist=dist, context=ctx, download=False, object_group=obj_group)
            call_ms.append((time.perf_counter() - t_call) * 1e3)
    except Exception as e:       # noqa: BLE001
        res, err = None, "%s: %s" % (type(e).__name__, e)
    ms = res["ms"] if res is not None else None
    vals = [(ms["halo"] + ms["local"] + ms["exchange"] + ms["finish"]) if ms else -1.0, ms["halo"] if ms else -1.0, ms["local"] if ms else -1.0,
            ms["exchange"] if ms else -1.0, ms["finish"] if ms else -1.0, float(res["boundary"]["triangles"]) if res else -1.0,
            float(res["counts"]["n_triangles"]) if res else -1.0]
    rdev = dev if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(vals, dtype=torch.float64, device=rdev)
    bad = torch.tensor([0 if res is not None else 1], dtype=torch.int32, device=rdev)
    tsum = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if int(bad.item()):
        return {"ms": None, "error": err or "failed on another rank"}
    t, tsum = t.tolist(), tsum.tolist()
    out = {"ms": t[0], "halo_ms": t[1], "local_ms": t[2], "exchange_ms": t[3], "finish_ms": t[4],
           "boundary_triangles_max": int(t[5]), "triangles_all_ranks": int(tsum[6]),
           "note": "max over ranks; per rank: 2+3 planes from the neighbours, march of own + 2 layers of cells each side, local weld / tiny / clean / "
                   "components on the GPU, (hash, label) of the boundary triangles to the upper neighbour (device to device), label pairs and "
                   "start-triangle candidates to rank 0 and the flips back (gather_object / scatter_object_list), winding + own part compacted "
                   "on the device; no mesh leaves its rank"}
    if rank == 0 and res.get("stats"):
        out["merge"] = res["stats"]
    out["calls_ms_rank0"] = call_ms        # first (allocations, code objects), second, and the timed third call as rank 0 saw them
    out["timed"] = "the third of three calls, max over ranks (no best-of)"
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist
    from contourist_amd import _ffi, synthetic
    from contourist_amd import distributed as cxdist

    # stdout carries ONE line, the JSON line of rank 0: everything else that lands on file descriptor 1 -- gloo announces its
    # connections there from C++ ("[Gloo] Rank 0 is connected to ...") -- goes to stderr; the line itself is written to the saved
    # descriptor at the end
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    def emit_line(text):
        sys.stdout.flush()
        os.write(line_fd, (text + "\n").encode())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("BENCH_DRY") == "1":
        # plumbing rehearsal without a GPU (tests/test_bench_launcher.py): rendezvous, one collective, the line's keys
        if distributed:
            dist.init_process_group("gloo")
            world, rank = dist.get_world_size(), dist.get_rank()
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert int(t.item()) == world
        lo, hi = cxdist.slab_bounds(args.size, world, rank) if not args.weak else (0, args.size)
        if rank == 0:
            emit_line(json.dumps({"metric": "Mvoxels/s isosurface extraction on %d^3 fp32 grid" % args.size, "value": None, "unit": "Mvoxels/s",
                                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak" if args.weak else "strong",
                                  "dry_run": True, "planes_rank0": [lo, hi]}))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
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
        world = dist.get_world_size()
        rank = dist.get_rank()
    n = args.size
    strong = not args.weak
    if args.streams <= 0:
        args.streams = 3 if world >= 4 else 2
    # distinct volumes cycled through: at least four (a 512^3 grid is 537 MB = twice the 256 MB Infinity Cache, so with four a
    # volume comes back after 1.6 GB of other samples); small grids need enough to exceed 256 MB several times
    nrot = args.rotate or max(4, int(1.5e9 // (4 * n ** 3)))
    overlap_halo = distributed and os.environ.get("BENCH_SYNC_HALO", "0") != "1"
    if overlap_halo:
        # the halo of the next volume is exchanged while the current one is extracted, and with S extractions in flight the buffer
        # it lands in must not be one that is still being read: S + 1 buffers
        nrot = max(nrot, max(1, args.streams) + 1)

    nstreams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    ctxs = [_ffi.Context(device_index, stream=st.cuda_stream) for st in streams]
    ctx = ctxs[0]
    flags = _ffi.CX_DIAG_CPYTHON310 | (_ffi.CX_KERNEL_GENERIC if args.generic else 0)
    if os.environ.get("CX_DEBUG") == "1" and os.environ.get("BENCH_EXTRA_FLAGS"):      # A/B of debug flag bits (tools)
        flags |= int(os.environ["BENCH_EXTRA_FLAGS"], 0)

    # A rank that hangs in the set-up of the exchange (a collective its peers never enter) must not hang the job: a watchdog ends
    # THIS process with a non-zero code if set-up and the first steps take longer than BENCH_SETUP_TIMEOUT seconds (no re-exec; the
    # launcher then tears the other ranks down).  Disarmed once the warm-up steps of the timed job have run.
    watchdog = None
    if distributed:
        import threading

        def _give_up():
            print("# rank %d: set-up of the %d-rank job did not finish within the time limit -- exiting" % (rank, world), file=sys.stderr, flush=True)
            os._exit(4)
        watchdog = threading.Timer(float(os.environ.get("BENCH_SETUP_TIMEOUT", "600")), _give_up)
        watchdog.daemon = True
        watchdog.start()
    c_halo = False
    if distributed and os.environ.get("BENCH_TORCH_HALO", "0") != "1" and dist.get_backend() == "nccl":
        # the halo exchange inside the C call of a step (own RCCL communicators, one per context); a buffer is always used by the
        # same context: the number of rotating buffers becomes a multiple of the number of contexts
        c_halo = cxdist.own_communicators(ctxs, rank, world, dist)
        if c_halo:
            nrot = ((nrot + nstreams - 1) // nstreams) * nstreams
    job = Job(args, torch, dist, cxdist, synthetic, dev, rank, world, strong, nrot)
    if c_halo:
        # self-check before anything is timed: the plane the C call receives is the plane torch.distributed delivers
        probe = job.slabs[0].clone()
        if job.has_upper:
            probe[job.n_own].fill_(float("nan"))
        torch.cuda.synchronize()
        ctxs[0].halo_exchange(None, rank, world, probe.data_ptr(), job.n_own, n * n)
        ctxs[0].synchronize()
        want = job.slabs[0].clone()
        cxdist.exchange_halo(want, job.n_own, rank, world, dist)
        torch.cuda.synchronize()
        same = torch.tensor([1 if torch.equal(probe, want) else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        c_halo = bool(int(same.item()))
        del probe, want
        if not c_halo and rank == 0:
            print("# C-side halo exchange disagrees with torch.distributed: staying on torch.distributed", file=sys.stderr, flush=True)
    elapsed, timing, final = run_job(args, torch, dist, ctxs, streams, job, flags, overlap_halo, c_halo=c_halo)
    if watchdog is not None:
        watchdog.cancel()
    if os.environ.get("BENCH_NO_EVENTS") == "1":      # measurement of what the per-kernel events cost: the same region without them
        el2, _, _ = run_job(args, torch, dist, ctxs, streams, job, flags, overlap_halo, timing=False, c_halo=c_halo)
        if rank == 0:
            print("# without per-kernel events: %.4f ms/step (with: %.4f)" % (el2 / args.steps * 1e3, elapsed / args.steps * 1e3), file=sys.stderr, flush=True)
    single = None
    if nstreams > 1 and rank == 0 and not distributed and not args.no_single_stream:
        # the same steps on ONE stream: per-kernel durations that do not overlap with another extraction's kernels
        s_el, s_tm, _ = run_job(args, torch, dist, ctxs[:1], streams[:1], job, flags, overlap_halo, c_halo=c_halo)
        single = (s_el, s_tm)

    weak_line = None
    if distributed and strong:
        # second field: the weak-scaling figure (one size^3 slab per GPU) of the same build
        host0 = None
        del job.slabs[:]
        torch.cuda.empty_cache()
        wjob = Job(args, torch, dist, cxdist, synthetic, dev, rank, world, False, nrot)
        wel, _, _ = run_job(args, torch, dist, ctxs, streams, wjob, flags, overlap_halo, timing=False, c_halo=c_halo)
        weak_line = {"value": wjob.total_samples * args.steps / wel / 1e6, "unit": "Mvoxels/s", "ms_per_step": wel / args.steps * 1e3,
                     "workload": "one %d^3 slab per GPU (%d x %d x %d volume)" % (n, world * n, n, n)}
        del wjob.slabs[:]
        torch.cuda.empty_cache()
        job = Job(args, torch, dist, cxdist, synthetic, dev, rank, world, strong, 1)

    projection = None
    if not distributed and n >= 256 and not args.no_single_stream:
        # what one rank of an N-way strong run does, measured on THIS GPU: Level 0 of a slab of n / N planes (+ its halo plane) of
        # the same field, extractions in flight as in the timed region.  A projection -- no halo exchange, no second GPU.
        projection = {"note": "PROJECTION from one GPU: ms per extraction of a slab of n/N planes (+1 halo plane) of the bench field, %d in flight, no halo "
                              "exchange; speedup bound = ms(full) / ms(slab)" % nstreams, "ms": {}}
        for parts in (1, 2, 4, 8):
            planes = n // parts
            lo = (n - planes) // 2
            subs = [job.slabs[k % len(job.slabs)][lo:lo + planes + (1 if lo + planes < n else 0)] for k in range(nstreams)]
            for c, sv in zip(ctxs, subs):
                c.set_origin(lo, 0, 0)
                c.adopt_device_grid(sv.data_ptr(), tuple(sv.shape), keepalive=sv)
                c.extract3d(args.value, flags)
            torch.cuda.synchronize()
            best = None
            for rnd in range(3):
                t0 = time.perf_counter()
                for i in range(40):
                    ctxs[i % nstreams].extract3d_async(args.value, flags)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 40 * 1e3
                best = dt if best is None else min(best, dt)
            projection["ms"][str(planes)] = best
        full = projection["ms"][str(n)]
        projection["speedup_bound"] = {str(p): full / projection["ms"][str(n // p)] for p in (2, 4, 8)}
        for c in ctxs:
            c.set_origin(job.origin0, 0, 0)
    multi = None
    if args.levels > 0:
        multi = run_levels(args, torch, dist, _ffi, job, streams[0], device_index, flags, distributed)

    sharded = None
    if distributed and not args.no_api:
        sharded = run_sharded_level1(args, torch, dist, cxdist, job, dev, rank, world, strong, ctxs[0])
        for c in ctxs:
            c.set_origin(job.origin0, 0, 0)

    if rank == 0:
        local_samples = job.slabs[0].numel()
        alg_bytes = 4.0 * local_samples             # 4 B per input sample, read once
        transport = ("RCCL over xGMI" if dist.get_backend() == "nccl" else "gloo, host-staged (a rehearsal, not RCCL)") if distributed else None
        # sum of the fp32 bit patterns of rank 0's first volume (its slab + halo plane): the field is generated on the host and is the
        # same bits in every environment (under rocprofv3 too); tests/test_gpu_bench_fields.py holds the whole-volume values
        field_checksum = synthetic.field_checksum(job.slabs[0][:job.n_own])
        # the ceiling as measured HERE, beside the 8 TB/s of the data sheet: a plain streaming read of the rotated volumes (16-byte
        # loads; best launch) and a device-to-device copy of one of them (hipMemcpyDtoD through torch: bytes read + bytes written)
        # (on two scratch buffers of 512 MiB each, read / copied in turn: twice the Infinity Cache, whatever the volume's size)
        scratch = [torch.empty(1 << 27, dtype=torch.float32, device=dev).fill_(1.0) for _ in range(2)]
        measured_read = 0.0
        for _rnd in range(2):
            for buf in scratch:
                measured_read = max(measured_read, ctx.measure_read_bandwidth(buf.data_ptr(), buf.numel() * 4, 1))
        dst = torch.empty_like(scratch[0])
        dst.copy_(scratch[0])
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for k in range(4):
            dst.copy_(scratch[k & 1])
        ev1.record()
        torch.cuda.synchronize()
        measured_copy = 2.0 * 4.0 * dst.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
        del scratch
        del dst

        def kernel_table(tmg):
            """per-kernel durations (HIP events on the extraction streams, inside a timed region) with the algorithmic bytes of
            each: stream = 4 B per sample read; emit = 16 B per vertex record + 12 B per triangle written"""
            nt_ = max(tmg["n"], 1)
            rows = []
            for key, name in ctx.kernel_names():
                ms = tmg[key] / nt_
                if key == "stream_ms":
                    ab = alg_bytes
                elif key == "scan_ms":
                    ab = 0.0
                elif key == "cells_ms":
                    ab = ctx.vertex_stage_bytes(final)
                else:
                    ab = ctx.triangle_stage_bytes(final)
                rows.append({"name": name, "ms": ms, "alg_bytes": ab})
            for kk in rows:
                kk["GBps"] = kk["alg_bytes"] / (kk["ms"] * 1e-3) / 1e9 if kk["ms"] > 0 else 0.0
                kk["frac"] = kk["GBps"] / HBM_PEAK_GBS
            rows = [kk for kk in rows if kk["ms"] > 0]
            return rows, (tmg["classify_ms"] + tmg["emit_ms"]) / nt_
        kernels, kernel_sum_ms = kernel_table(timing)
        dom = max(kernels, key=lambda kk: kk["ms"])
        step_ms = elapsed / args.steps * 1e3
        # 4 B x samples of one extraction / time of one extraction.  With ONE extraction in flight that time is the
        # sum of its Level-0 kernel durations; with several in flight the kernels of different extractions overlap in time
        # (their durations add up to MORE than the time an extraction takes), so the time is the step time of the timed region
        level0_ms = kernel_sum_ms if nstreams == 1 else step_ms
        achieved = alg_bytes / (level0_ms * 1e-3) / 1e9 if level0_ms > 0 else 0.0
        bytes_necessary = alg_bytes + 8.0 * final["n_vertices"] + 12.0 * final["n_triangles"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("level0_%d" % n)
            except Exception:
                traffic = None
        out = {
            "metric": "Mvoxels/s isosurface extraction on 512^3 fp32 grid" if n == 512 else "Mvoxels/s isosurface extraction on %d^3 fp32 grid" % n,
            "value": job.total_samples * args.steps / elapsed / 1e6,
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%dx%d fp32 smooth-noise %s ([1,2,1]/4 x %d passes, closed interior), isovalue %g, marching tetrahedra Level-0 (classify+interpolate+emit indexed mesh)"
                            % (n, n, n, ("volume split over %d GPUs" % world if distributed else "volume") if strong else "slab per GPU", args.passes, args.value),
                "partition": (("one volume in axis-0 slabs, 1-plane halo over %s" if strong else
                               "one slab per GPU (axis 0), 1-plane halo over %s") % transport) if distributed else "single GPU",
                "halo_exchange": ("one C call per step: RCCL send / receive on the extraction stream (ONE communicator per rank, shared by its contexts), then the extraction" if c_halo else
                                  ("torch.distributed (%s), overlapped with the previous volume's extraction" % transport if overlap_halo
                                   else "torch.distributed (%s), in line" % transport)) if distributed else None,
                "field_checksum": field_checksum,
                "active_voxel_fraction": final["n_border_voxels"] / float(max((job.n_own - (0 if job.has_upper else 1)) * (n - 1) ** 2, 1)),
                "vertices_rank0": final["n_vertices"], "triangles_rank0": final["n_triangles"],
                "grids_rotated": nrot,
                "extractions_in_flight": nstreams,
                "classify_kernel": "generic" if args.generic else "auto",
            },
            # achieved = 4 B x samples of one extraction / time of one extraction (see above); per-kernel figures
            # (HIP events on the extraction streams, rank 0, inside the timed region) under `kernels`
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "all Level-0 kernels of one extraction (%s)%s" % (" + ".join(kk["name"] for kk in kernels),
                          "" if nstreams == 1 else "; %d extractions in flight, time = step time" % nstreams),
                "kernel_ms": level0_ms,
                "dominant_kernel": dom["name"], "dominant_kernel_ms": dom["ms"],
                "kernels": kernels,
                "kernel_sum_ms": kernel_sum_ms,
                "level0_ms": level0_ms,
                "level0_frac": achieved / HBM_PEAK_GBS,
                # "also report measured stream-read / DtoD bandwidth on the box and quote both fractions"
                "measured_peak_GBps": measured_read,
                "frac_of_measured": achieved / measured_read if measured_read > 0 else None,
                "measured_copy_GBps": measured_copy,
                "measured_peak_note": "stream read: cx_measure_read_bandwidth over the rotated volumes (16-byte loads, best launch); copy: "
                                      "hipMemcpyDtoD of one volume, bytes read + written",
                # what an extraction cannot avoid moving: the samples once + the mesh it leaves (8-byte vertex records, 12-byte triangles)
                "bytes_necessary": bytes_necessary,
                "necessary_GBps": bytes_necessary / (level0_ms * 1e-3) / 1e9 if level0_ms > 0 else 0.0,
                "frac_necessary": bytes_necessary / (level0_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if level0_ms > 0 else 0.0,
                # what the pipeline actually moves, and how close that is to what this device streams: the extraction is bound by its
                # bytes -- each of its three big kernels runs at 4.5-5 TB/s of fabric traffic
                "traffic_GBps": (traffic / (level0_ms * 1e-3) / 1e9) if (traffic and level0_ms > 0) else None,
                "traffic_over_measured_peak": (traffic / (level0_ms * 1e-3) / 1e9 / measured_read) if (traffic and level0_ms > 0 and measured_read > 0) else None,
                "traffic_note": "HBM-side bytes of one extraction = the L2's fabric requests, each at its size (TCC_EA0_RDREQ_32B/_64B/_128B, "
                                "TCC_EA0_WRREQ_64B / TCC_EA0_WRREQ), from separate rocprofv3 --pmc passes over tools/prof_step.py on the same field "
                                "(profiles/traffic.json, tools/traffic.py); counters cannot be read inside this process.  Rounds 1-3 derived it from "
                                "FETCH_SIZE, which tallies the emit stages' 128-byte gather requests at 64 bytes: 1.40 GB was reported where 1.66 GB moved",
            },
        }
        if single is not None:
            # one extraction in flight: kernels that do not overlap with another extraction's, their sum = the extraction
            s_kernels, s_sum = kernel_table(single[1])
            s_ach = alg_bytes / (s_sum * 1e-3) / 1e9 if s_sum > 0 else 0.0
            out["roofline"]["single_stream"] = {
                "ms_per_step": single[0] / args.steps * 1e3, "value": job.total_samples * args.steps / single[0] / 1e6,
                "level0_ms": s_sum, "achieved": s_ach, "frac": s_ach / HBM_PEAK_GBS, "kernels": s_kernels,
                "note": "the same steps on one stream: 4N / sum of the Level-0 kernel durations of one extraction"}
        if weak_line is not None:
            out["weak"] = weak_line
        if projection is not None:
            out["slab_projection"] = projection
        if world == 1 and not args.no_config4:
            try:
                out["config4"] = run_config4(torch, _ffi, synthetic, dev, device_index)
            except Exception as e:      # an extra: the headline stands without it
                out["config4"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if multi is not None:
            out["multi_level"] = multi
            out["levels"] = len(multi["levels"])
            out["ms_all_levels"] = multi["ms_all_levels"]
            out["Mvoxel_levels_per_s"] = multi["Mvoxel_levels_per_s"]
        if sharded is not None:
            out["level1_sharded"] = sharded
            if sharded.get("ms") is not None:
                out["level1_ms"] = sharded["ms"]
        if world == 1 and not args.no_api:
            # what the reference's API returns: Level 0 + Level 1 (weld, tiny collapse, clean, orient) + download
            buf = job.slabs[0]
            ctx.adopt_device_grid(buf.data_ptr(), tuple(buf.shape), keepalive=buf)
            # one untimed pass first: the Level-1 state is allocated, the post-pass kernels and the pinned staging buffers of the
            # download exist (a fresh process pays ~0.1 s for those once; a caller streaming volumes pays it once, too)
            ctx.extract3d(args.value, flags)
            warm = ctx.postprocess3d()
            _w = ctx.download_level1(warm)
            del _w
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c0 = ctx.extract3d(args.value, flags)
            t1 = time.perf_counter()
            post = ctx.postprocess3d()
            t2 = time.perf_counter()
            pts, tris = ctx.download_level1(post)
            t3 = time.perf_counter()
            out["level1_ms"] = (t2 - t1) * 1e3
            out["api_ms"] = (t3 - t0) * 1e3
            out["api"] = {"extract_sync_ms": (t1 - t0) * 1e3, "level1_ms": (t2 - t1) * 1e3, "download_ms": (t3 - t2) * 1e3,
                          "level1_vertices": int(post["n_vertices"]), "level1_triangles": int(post["n_triangles"]),
                          "Mvoxels_per_s_through_api": n ** 3 / (t3 - t0) / 1e6,
                          "note": "get_points_and_triangles() equivalent: extract + post-pass + download of float64 points / int32 triangles (second call on the context: allocations exist)"}
            del pts, tris
        parity_error = None
        if world == 1 and not args.no_cpu_baseline:
            host = job.slabs[0].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(host, args.value)
            # the HIP path on the very sample the CPU leg marched (planes 0:P of rank 0's first volume, as a grid of its own): the
            # counts must be equal -- a cheap end-to-end parity check inside every bench run (the exact comparison of the whole mesh
            # is tests/test_gpu_bench_fields.py)
            P_ = out["cpu_baseline"]["sample_planes"]
            sub = job.slabs[0][:P_]
            ctx.set_origin(0, 0, 0)
            ctx.adopt_device_grid(sub.data_ptr(), tuple(sub.shape), keepalive=sub)
            hc = ctx.extract3d(args.value, flags)
            hip_counts = {k: int(hc[k]) for k in ("n_vertices", "n_triangles", "n_border_voxels")}
            out["cpu_baseline"]["hip_counts_same_sample"] = hip_counts
            out["cpu_baseline"]["counts_equal"] = hip_counts == out["cpu_baseline"]["oracle_counts"]
            if not out["cpu_baseline"]["counts_equal"]:
                parity_error = "HIP counts %r != oracle counts %r on the CPU leg's sample" % (hip_counts, out["cpu_baseline"]["oracle_counts"])
                out["parity_error"] = parity_error
        emit_line(json.dumps(out))
        if parity_error:
            print("# PARITY ERROR: " + parity_error, file=sys.stderr, flush=True)
            sys.exit(3)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
