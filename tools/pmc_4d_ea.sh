#!/bin/bash
# fabric requests by size + SQ basics of the 4-D Level-0 kernels (tools/bench4d.py); output gpurun_out/pmc4d_ea/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc4d_ea
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/bench4d.py > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/failed.txt; }
run ea_rd TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum
run ea_wr TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, collections, sys
tab = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
        if not k.startswith("cx_k_") or "hash" in k: continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            tab[c][k] = sum(v[1:]) / max(len(v) - 1, 1)
ks = sorted({k for c in tab for k in tab[c]})
print("%-34s" % "counter (mean per launch)" + "".join("%22s" % k[5:] for k in ks))
for c in sorted(tab):
    print("%-34s" % c + "".join("%22.4g" % tab[c].get(k, float("nan")) for k in ks))
rd = {k: 32 * tab["TCC_EA0_RDREQ_32B_sum"].get(k, 0) + 64 * tab["TCC_EA0_RDREQ_64B_sum"].get(k, 0) + 128 * tab["TCC_EA0_RDREQ_128B_sum"].get(k, 0) for k in ks}
wr = {k: 64 * tab["TCC_EA0_WRREQ_64B_sum"].get(k, 0) + 32 * max(tab["TCC_EA0_WRREQ_sum"].get(k, 0) - tab["TCC_EA0_WRREQ_64B_sum"].get(k, 0), 0) for k in ks}
print("%-34s" % "MB read" + "".join("%22.1f" % (rd[k] / 1e6) for k in ks))
print("%-34s" % "MB written" + "".join("%22.1f" % (wr[k] / 1e6) for k in ks))
PY
cat $OUT/summary.txt; cat $OUT/failed.txt 2>/dev/null
