#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
rocprofv3 -L > gpurun_out/r4/counters_available.txt 2>&1
export CX_DEBUG=1
for v in base pad27 pad48; do
  if [ $v = base ]; then unset CX_LIB_PATH; else export CX_LIB_PATH=$GRAFT_REPO_ROOT/contourist_amd/lib/variants/lib_$v.so; fi
  TAG=$v timeout -k 10 300 python3 tools/time_modes.py 512 ablate_staged 2>&1 | grep -E "staged|canon|Error|error"
done > gpurun_out/r4/occ_ablate.txt 2>&1
cat gpurun_out/r4/occ_ablate.txt
grep -c . gpurun_out/r4/counters_available.txt
