#!/bin/bash
# kernel trace of tools/bench_seeded.py: where the seeded selection's time goes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/kts
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kts -- python3 tools/bench_seeded.py > gpurun_out/prof/bench_seeded.txt 2> gpurun_out/prof/kts.err
f=$(find gpurun_out/prof/kts -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|cxs_k|fill|Memset|memset|copyBuffer" "$f" | sed 's/(.*)"/"/' | cut -d, -f1-4
cat gpurun_out/prof/bench_seeded.txt | tail -3
