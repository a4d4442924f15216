#!/bin/bash
# round 4, second GPU call: instruction rates, whole-volume parity of the host-generated bench field, HIP API trace of the sharded Level 1 (the 70-120 ms stall)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
bash tools/micro/run_valu_rate.sh > gpurun_out/r4/valu_rate2.txt 2>&1
timeout -k 10 900 python3 -m pytest tests/test_gpu_bench_fields.py -x -q -m gpu -k whole_volume > gpurun_out/r4/whole_volume.txt 2>&1
tail -5 gpurun_out/r4/whole_volume.txt
python3 tools/field_check.py 512 > gpurun_out/r4/field_plain2.json 2> gpurun_out/r4/field_plain2.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/fc_kt -- python3 tools/field_check.py 512 > gpurun_out/r4/field_rocprof2.json 2> gpurun_out/r4/field_rocprof2.err
rm -rf gpurun_out/r4/fc_kt
cat gpurun_out/r4/field_plain2.json gpurun_out/r4/field_rocprof2.json
rm -rf /tmp/shard_trace
timeout -k 10 400 rocprofv3 --hip-trace --output-format csv -d /tmp/shard_trace -- python3 tools/shard_time.py 512 8 > gpurun_out/r4/shard_time_traced.txt 2>&1
python3 - <<'PY' > gpurun_out/r4/shard_slow_calls.txt 2>&1
import csv, glob
for f in glob.glob("/tmp/shard_trace/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), "calls")
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    slow = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 3_000_000]
    for r in slow:
        print("%10.3f ms  +%9.3f ms  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Function"]))
PY
tail -30 gpurun_out/r4/shard_slow_calls.txt
cat gpurun_out/r4/shard_time_traced.txt | tail -8
cat gpurun_out/r4/valu_rate2.txt
