#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/ktsec
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ktsec -- python3 tools/bench_secondary.py > gpurun_out/prof/bench_secondary.txt 2> gpurun_out/prof/ktsec.err
f=$(ls -t gpurun_out/prof/ktsec/*/*kernel_stats.csv | head -1)
head -25 "$f" | sed 's/(.*)"/"/' | cut -d, -f1-4
