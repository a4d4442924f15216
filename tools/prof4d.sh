#!/bin/bash
# kernel-trace summary of the 4-D bench (config 4): gpurun_out/prof/kernel_stats_4d.csv + bench4d line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/kt4
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt4 -- python3 tools/bench4d.py > gpurun_out/prof/bench4d_under_rocprof.json 2> gpurun_out/prof/kt4.err
f=$(find gpurun_out/prof/kt4 -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|cx_k|cxp_k|cxs4_k" "$f" > gpurun_out/prof/kernel_stats_4d.csv
grep "cx_k_" gpurun_out/prof/kernel_stats_4d.csv | cut -d, -f1-4
