#!/bin/bash
# after `gpurun -- bash tools/profile_round.sh`: copy the summaries that came back under gpurun_out/prof/ into profiles/ (tracked)
R=${1:-r04}
cp gpurun_out/prof/bench_line.json profiles/${R}_bench_line_512.json
cp gpurun_out/prof/bench_under_rocprof.json profiles/${R}_bench_line_512_under_rocprof.json
cp gpurun_out/prof/kernel_stats_512.csv profiles/${R}_kernel_stats_512.csv
cp gpurun_out/prof/traffic.json profiles/traffic.json
cp gpurun_out/prof/traffic_detail.json profiles/traffic_detail.json
cp gpurun_out/prof/bench_levels_512.json profiles/${R}_bench_levels_512.json
cp gpurun_out/prof/bench4d.json profiles/${R}_bench4d_128x128x128x64.json
cp gpurun_out/prof/kernel_stats_4d.csv profiles/${R}_kernel_stats_4d.csv
cp gpurun_out/prof/kernel_stats_level1_512.csv profiles/${R}_kernel_stats_level1_512.csv
cp gpurun_out/prof/kernel_stats_512_one_stream.csv profiles/${R}_kernel_stats_512_one_stream.csv
cp gpurun_out/prof/bench_one_stream_under_rocprof.json profiles/${R}_bench_line_512_one_stream_under_rocprof.json
cp gpurun_out/prof/bench_level1_512.json profiles/${R}_bench_level1_512.json
grep -v amdgpu.ids gpurun_out/prof/shard_time_512x8.txt > profiles/${R}_shard_time_512x8.txt
cp gpurun_out/prof/sq_counters.txt profiles/${R}_sq_counters.txt
