#!/bin/bash
# kernel-trace summary of the Level-1 post-pass at 512^3: gpurun_out/prof/kernel_stats_level1_512.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/ktl1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ktl1 -- python3 tools/bench_level1.py 512 > gpurun_out/prof/bench_level1_under_rocprof.json 2> gpurun_out/prof/ktl1.err
f=$(find gpurun_out/prof/ktl1 -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|cxp_k|cx_k" "$f" > gpurun_out/prof/kernel_stats_level1_512.csv
timeout -k 10 200 python3 tools/bench_level1.py 512
