#!/bin/bash
# what the round's measurements come from: the default bench line, the rocprofv3 kernel-trace summary of the SAME command,
# and the FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel trace only) -> gpurun_out/prof/, summaries under profiles/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r02}
mkdir -p gpurun_out/prof gpurun_out/pmc profiles
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof/bench_line.json 2> gpurun_out/prof/bench.err
cp gpurun_out/prof/bench_line.json profiles/${R}_bench_line_512.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/kt.err
cp gpurun_out/prof/bench_under_rocprof.json profiles/${R}_bench_line_512_under_rocprof.json
f=$(find gpurun_out/prof/kt -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|cx_k" "$f" > profiles/${R}_kernel_stats_512.csv
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z | sed 's/_size//')
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/$n -- python3 tools/prof_step.py 512 > gpurun_out/pmc/$n.log 2>&1
done
python3 tools/traffic.py gpurun_out/pmc > gpurun_out/prof/traffic.log
cat profiles/${R}_kernel_stats_512.csv
cat profiles/traffic.json
