"""bench.py --gpus N must start its N ranks itself (the driver runs `python bench.py --gpus N` for N > 1 as well as
under torch.distributed.run) and default to STRONG scaling on one volume (SURVEY config 3).  CPU part: the launcher,
the rendezvous and the line's keys with BENCH_DRY=1 (no GPU work); GPU part: the real path, two ranks sharing the
one GPU over gloo on a 64^3 volume."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env, *argv):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    return json.loads(lines[0])


def test_launcher_starts_the_ranks_dry():
    line = run_bench({"BENCH_DRY": "1"}, "--gpus", "2", "--size", "64", "--steps", "2", "--warmup", "1")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["planes_rank0"] == [0, 32]
    line = run_bench({"BENCH_DRY": "1"}, "--gpus", "3", "--size", "64", "--weak")
    assert line["n_gpus"] == 3 and line["scaling"] == "weak"
    line = run_bench({"BENCH_DRY": "1"}, "--size", "64")
    assert line["n_gpus"] == 1 and line["scaling"] == "strong"


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_strong_scaling():
    line = run_bench({"BENCH_BACKEND": "gloo"}, "--gpus", "2", "--size", "64", "--steps", "3", "--warmup", "1", "--passes", "40")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    assert line["weak"]["value"] > 0
    assert line["levels"] == 8 and line["ms_all_levels"] > 0 and line["Mvoxel_levels_per_s"] > 0      # config 5: every rank its slab x 8 levels
    # Level 1 sharded over the ranks: nobody gathers the mesh; the slowest rank's time is the line's level1_ms
    sh = line["level1_sharded"]
    assert sh.get("error") is None and sh["ms"] > 0 and line["level1_ms"] == sh["ms"]
    assert sh["merge"]["unmatched"] == 0 and 0 < sh["boundary_triangles_max"] < sh["triangles_all_ranks"]
    one = run_bench({}, "--gpus", "1", "--size", "64", "--steps", "3", "--warmup", "1", "--passes", "40", "--no-cpu-baseline")
    assert one["n_gpus"] == 1 and one["api_ms"] > 0 and one["level1_ms"] > 0
    assert one["levels"] == 8 and len(one["multi_level"]["triangles_per_level_rank0"]) == 8 and one["ms_all_levels"] > 0
