#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/kts4
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kts4 -- python3 tools/bench_seeded4d.py > gpurun_out/prof/bench_seeded4d.txt 2> gpurun_out/prof/kts4.err
f=$(ls -t gpurun_out/prof/kts4/*/*kernel_stats.csv | head -1)
grep -E "cxs4_k" "$f" | sed 's/(.*)"/"/' | cut -d, -f1-4
tail -1 gpurun_out/prof/bench_seeded4d.txt
