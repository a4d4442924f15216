#!/bin/bash
# rocprofv3 SQ counter passes over tools/prof_step.py (fused emit path); results under gpurun_out/pmc_em/<name>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_em
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_em/$name -- python3 tools/prof_step.py 512 > gpurun_out/pmc_em/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_em/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            if "cx_k" not in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            print(d.split("/")[-2], k, {c: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for c, v in cs.items()})
PY
