#!/bin/bash
# round 4, third GPU call: time of the emit stages against occupancy (LDS padding variants); more rounds of the sharded Level 1 under the HIP trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export CX_DEBUG=1
for v in base pad8 pad20 pad27 pad48; do
  if [ $v = base ]; then unset CX_LIB_PATH; else export CX_LIB_PATH=$GRAFT_REPO_ROOT/contourist_amd/lib/variants/lib_$v.so; fi
  TAG=$v timeout -k 10 200 python3 tools/time_modes.py 512 staged 2>&1 | grep -E "staged|Error|error" 
done > gpurun_out/r4/occ_pad.txt 2>&1
cat gpurun_out/r4/occ_pad.txt
unset CX_LIB_PATH
rm -rf /tmp/shard_trace
timeout -k 10 400 rocprofv3 --hip-trace --output-format csv -d /tmp/shard_trace -- python3 tools/shard_time.py 512 8 40 > gpurun_out/r4/shard_time_traced40.txt 2>&1
python3 - <<'PY' > gpurun_out/r4/shard_slow_calls40.txt 2>&1
import csv, glob
for f in glob.glob("/tmp/shard_trace/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), "calls")
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if d > 3_000_000 and int(r["Start_Timestamp"]) - t0 > 3_000_000_000:
            print("%10.3f ms  +%9.3f ms  %s  start %s end %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d / 1e6, r["Function"], r["Start_Timestamp"], r["End_Timestamp"]))
PY
cat gpurun_out/r4/shard_slow_calls40.txt | tail -20
grep -E "SLOW|per rank" gpurun_out/r4/shard_time_traced40.txt | cut -c1-300
timeout -k 10 300 python3 tools/shard_time.py 512 8 40 > gpurun_out/r4/shard_time_plain40.txt 2>&1
grep -E "SLOW|per rank" gpurun_out/r4/shard_time_plain40.txt | cut -c1-300
