#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests/test_gpu_level0.py tests/test_gpu_fused_emit.py tests/test_gpu_context_reuse.py tests/test_gpu_bench_fields.py -x -q -m gpu > gpurun_out/r4/t5_tests.txt 2>&1
tail -8 gpurun_out/r4/t5_tests.txt | cut -c1-300
export CX_DEBUG=1
TAG=tq timeout -k 10 200 python3 tools/time_modes.py 512 staged 2>&1 | grep -E "staged  |Error|error"
CX_NO_TQ=1 TAG=gather timeout -k 10 200 python3 tools/time_modes.py 512 staged 2>&1 | grep -E "staged  |Error|error"
TAG=tq timeout -k 10 200 python3 tools/stream_ab.py 512 2>&1 | grep -E "stream:|Error|error"
unset CX_DEBUG
python3 bench.py --steps 20 --warmup 5 --no-api --levels 0 --no-cpu-baseline > gpurun_out/r4/bench_t5.json 2> gpurun_out/r4/bench_t5.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_t5.json"))
r = d["roofline"]
print("ms/step", d["ms_per_step"], "frac", r["frac"], "single", r["single_stream"]["ms_per_step"], [(k["name"], round(k["ms"], 4)) for k in r["single_stream"]["kernels"]])
PY
