#!/bin/bash
# rocprofv3 TA / TCP (vector memory pipeline) counter passes over tools/prof_step.py; argument: flags of the extraction (default 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FL=${1:-1}
OUT=gpurun_out/pmc_ta_$FL
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/prof_step.py 512 $FL > $OUT/$name.log 2>&1; }
run ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE &&
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum &&
run tcp2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum &&
run tcp3 TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
python3 - $OUT <<'PY'
import csv, glob, collections, sys
for d in sorted(glob.glob(sys.argv[1] + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:28]
            if "cx_k" not in k or "hash" in k or "scan" in k or "list" in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            print(d.split("/")[-2], k, {c: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for c, v in cs.items()})
PY
