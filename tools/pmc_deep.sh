#!/bin/bash
# rocprofv3 counter passes (one --pmc group per run, kernel trace only) over tools/prof_step.py: LDS, vector-memory issue, L1 (TCP), L2 (TCC) and
# fabric (EA) counters of the Level-0 kernels.  usage: tools/pmc_deep.sh [tag] ; output gpurun_out/pmc_deep_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-base}
OUT=gpurun_out/pmc_deep_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/prof_step.py 512 1 > $OUT/$name.log 2>&1 || echo "pass $name failed" >> $OUT/failed.txt; }
run sq_lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH
run sq_vmem SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES
run sq_base SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp2 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum
run tcc3 TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run tcc4 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum
run tcc5 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, collections, sys
tab = collections.defaultdict(dict)
for d in sorted(glob.glob(sys.argv[1] + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "cx_k" not in k or "hash" in k: continue
            k = k.replace("void ", "").split("(")[0].split("<")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                tab[c][k] = sum(v[1:]) / max(len(v) - 1, 1)
ks = ["cx_k_stream", "cx_k_scan_list", "cx_k_emit_vertices", "cx_k_emit_triangles_q"]
print("%-42s" % "counter (mean per launch, launches 2..N)" + "".join("%22s" % k[5:] for k in ks))
for c in sorted(tab):
    print("%-42s" % c + "".join("%22.4g" % tab[c].get(k, float("nan")) for k in ks))
PY
cat $OUT/summary.txt; cat $OUT/failed.txt 2>/dev/null
