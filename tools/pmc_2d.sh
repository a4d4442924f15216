#!/bin/bash
# PMC passes over tools/bench2d.py (2-D contour kernels); results under gpurun_out/pmc2d/<name>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc2d
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc2d/$name -- python3 tools/bench2d.py > gpurun_out/pmc2d/$name.log 2>&1; }
run fetch FETCH_SIZE &&
run write WRITE_SIZE
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc2d/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if not (k.startswith("c2_k_") or k.startswith("cxp_k_scan")): continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                tot[k][c] = (round(sum(v) / max(len(v), 1) * 1024 / 1e6, 2), len(v))
for k, cs in sorted(tot.items()):
    print(k, {c: "%.2f MB/launch x %d launches" % v for c, v in cs.items()})
PY
