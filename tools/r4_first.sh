#!/bin/bash
# round 4, first GPU call: (1) field checksums plain vs under rocprofv3, (2) SQ counter table of the current build, (3) bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python3 tools/field_check.py 512 x > gpurun_out/r4/field_plain.json 2> gpurun_out/r4/field_plain.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/fc_kt -- python3 tools/field_check.py 512 x > gpurun_out/r4/field_rocprof.json 2> gpurun_out/r4/field_rocprof.err
rm -rf gpurun_out/r4/fc_kt
cat gpurun_out/r4/field_plain.json gpurun_out/r4/field_rocprof.json
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench0.json 2> gpurun_out/r4/bench0.err
bash tools/pmc_sq.sh 1 > gpurun_out/r4/pmc_sq0.txt 2>&1
cat gpurun_out/r4/pmc_sq0.txt
head -c 3000 gpurun_out/r4/bench0.json
