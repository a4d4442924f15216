#!/bin/bash
# rocprofv3 SQ counter passes over tools/prof_step.py; arguments: extraction flags (default 1), tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FL=${1:-1}
OUT=gpurun_out/pmc_sq_$FL
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/prof_step.py 512 $FL > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
python3 - $OUT <<'PY'
import csv, glob, collections, sys
for d in sorted(glob.glob(sys.argv[1] + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:30]
            if "cx_k" not in k or "hash" in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            print(d.split("/")[-2], k, {c: round(sum(v[1:]) / max(len(v) - 1, 1) / 1e6, 2) for c, v in cs.items()})
PY
