#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export CX_DEBUG=1
for v in base qad qadd3; do
  if [ $v = base ]; then unset CX_LIB_PATH; else export CX_LIB_PATH=$GRAFT_REPO_ROOT/contourist_amd/lib/variants/lib_$v.so; fi
  TAG=$v timeout -k 10 200 python3 tools/time_modes.py 512 staged 2>&1 | grep -E "staged  |Error|error"
  TAG=$v timeout -k 10 200 python3 tools/stream_ab.py 512 2>&1 | grep -E "stream:|Error|error"
done > gpurun_out/r4/s1_qad.txt 2>&1
cat gpurun_out/r4/s1_qad.txt
unset CX_LIB_PATH; unset CX_DEBUG
BENCH_BACKEND=gloo timeout -k 10 900 python3 bench.py --gpus 4 --steps 6 --warmup 2 > gpurun_out/r4/bench_4rank_gloo.json 2> gpurun_out/r4/bench_4rank_gloo.err
grep -v "amdgpu.ids\|Gloo\|socket.cpp" gpurun_out/r4/bench_4rank_gloo.err | tail -5 | cut -c1-300
python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4/bench_4rank_gloo.json"))
    print("4 ranks gloo on one GPU:", d["n_gpus"], round(d["value"]), d["ms_per_step"], d["config"]["halo_exchange"], d["config"]["extractions_in_flight"], d["config"]["grids_rotated"])
    print(d.get("level1_sharded"))
    print(d.get("ms_all_levels"), d.get("weak"))
except Exception as e:
    print("no line", e)
PY
