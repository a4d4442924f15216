#!/bin/bash
# what the round's measurements come from: the default bench line, the rocprofv3 kernel-trace summary of the SAME command,
# and the FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel trace only).  Everything goes to gpurun_out/prof/
# (the only directory that travels back from the GPU box); tools/collect_profiles.sh then copies the summaries to profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof gpurun_out/pmc
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof/bench_line.json 2> gpurun_out/prof/bench.err
# the kernel trace of the SAME command (two extractions in flight: kernels of consecutive extractions overlap, their durations are
# longer than alone) and of the same steps on one stream (--streams 1: durations that add up to the extraction)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api --no-single-stream --levels 0 > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/kt.err
f=$(find gpurun_out/prof/kt -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|cx_k" "$f" > gpurun_out/prof/kernel_stats_512.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt1 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api --no-single-stream --levels 0 --streams 1 > gpurun_out/prof/bench_one_stream_under_rocprof.json 2> gpurun_out/prof/kt1.err
f=$(find gpurun_out/prof/kt1 -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|cx_k" "$f" > gpurun_out/prof/kernel_stats_512_one_stream.csv
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z | sed 's/_size//')
  rm -rf gpurun_out/pmc/$n
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/$n -- python3 tools/prof_step.py 512 > gpurun_out/pmc/$n.log 2>&1
done
# the fabric requests of the L2 by size: what the traffic figure is computed from (tools/traffic.py)
rm -rf gpurun_out/pmc/ea_rd gpurun_out/pmc/ea_wr
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc/ea_rd -- python3 tools/prof_step.py 512 > gpurun_out/pmc/ea_rd.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc/ea_wr -- python3 tools/prof_step.py 512 > gpurun_out/pmc/ea_wr.log 2>&1
bash tools/pmc_sq.sh 1 > gpurun_out/prof/sq_counters.txt 2>&1
python3 tools/traffic.py gpurun_out/pmc gpurun_out/prof > gpurun_out/prof/traffic.log
timeout -k 10 300 python3 tools/bench_levels.py 512 > gpurun_out/prof/bench_levels_512.json 2> gpurun_out/prof/levels.err
# Level 1 of the bench mesh: kernel trace of three post-passes (tools/l1_stats.py prints the per-pass table), and what one rank of an
# 8-way sharded Level 1 does (one GPU playing the ranks)
rm -rf gpurun_out/prof/ktl1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ktl1 -- python3 tools/bench_level1.py 512 > gpurun_out/prof/bench_level1_512.json 2> gpurun_out/prof/ktl1.err
f=$(find gpurun_out/prof/ktl1 -name "*kernel_stats.csv" | head -1)
grep -E "^\"Name\"|cxp_k|cx_k" "$f" > gpurun_out/prof/kernel_stats_level1_512.csv
timeout -k 10 300 python3 tools/shard_time.py 512 8 > gpurun_out/prof/shard_time_512x8.txt 2> gpurun_out/prof/shard.err
# config 4 (4-D): bench line without the profiler, then the kernel trace of the same command
timeout -k 10 300 python3 tools/bench4d.py > gpurun_out/prof/bench4d.json 2> gpurun_out/prof/bench4d.err
bash tools/prof4d.sh > /dev/null
cat gpurun_out/prof/kernel_stats_512.csv
cat gpurun_out/prof/traffic.json
