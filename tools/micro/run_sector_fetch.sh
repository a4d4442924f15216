#!/bin/bash
# builds and runs the fetch-granularity microbenchmark on the GPU box, plain and under rocprofv3 (request sizes per kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o /tmp/sector_fetch tools/micro/sector_fetch.hip || exit 1
timeout -k 10 120 /tmp/sector_fetch > gpurun_out/micro/sector_fetch.log 2>&1
rm -rf /tmp/sf_pmc
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/sf_pmc -- /tmp/sector_fetch > /dev/null 2>&1
python3 - <<'PY' >> gpurun_out/micro/sector_fetch.log
import csv, glob, collections
for f in glob.glob("/tmp/sf_pmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(k[:60], {c: v[-1] for c, v in sorted(acc[k].items())})
PY
cat gpurun_out/micro/sector_fetch.log
