#!/bin/bash
# quick A/B of a Level-0 kernel change: parity tests, the bench line (no CPU leg), one-stream kernel durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout -k 10 300 python3 -m pytest tests/test_gpu_level0.py tests/test_gpu_bench_fields.py -x -q > gpurun_out/ab/tests.log 2>&1; tail -2 gpurun_out/ab/tests.log
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api --levels 0 2> gpurun_out/ab/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], {k:v for k,v in d.items() if 'single' in k or 'one_stream' in k})"; done
rm -rf gpurun_out/ab/kt1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/kt1 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api --no-single-stream --levels 0 --streams 1 > gpurun_out/ab/b1.json 2> gpurun_out/ab/kt1.err
f=$(find gpurun_out/ab/kt1 -name "*kernel_stats.csv" | head -1)
grep -E "cx_k" "$f" | sed 's/(.*)"/"/' | cut -d, -f1-4
