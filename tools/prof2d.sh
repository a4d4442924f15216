#!/bin/bash
# kernel trace of tools/bench2d.py (8192^2 x 8 levels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/kt2d
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt2d -- python3 tools/bench2d.py > gpurun_out/prof/bench2d.json 2> gpurun_out/prof/kt2d.err
f=$(ls -t gpurun_out/prof/kt2d/*/*kernel_stats.csv | head -1)
grep -E "c2_k|cxp_k|rocclr" "$f" | sed 's/(.*)"/"/' | cut -d, -f1-4 | head -30
tail -c 600 gpurun_out/prof/bench2d.json
