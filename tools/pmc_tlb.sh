#!/bin/bash
# one small rocprofv3 counter pass (address translation in the vector L1) over tools/prof_step.py; argument: extraction flags
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FL=${1:-1}
OUT=gpurun_out/pmc_tlb_$FL
mkdir -p $OUT
timeout -k 10 120 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/t -- python3 tools/prof_step.py 512 $FL > $OUT/t.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:28]
        if "cx_k" not in k or "hash" in k: continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for c, v in cs.items()})
PY
