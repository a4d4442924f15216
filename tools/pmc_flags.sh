#!/bin/bash
# SQ instruction counters of the classify kernel under ablation flags
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcf
for fl in 0x1 0x810001 0x820001 0x9c0001; do
  timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmcf/f$fl -- python3 tools/prof_step.py 512 $fl > gpurun_out/pmcf/f$fl.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcf/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:30]
            if "cx_k_classify" not in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            print(d.split("/")[-2], {c: round(sum(v[2:]) / max(len(v) - 2, 1) / 1e6, 2) for c, v in cs.items()})
PY
