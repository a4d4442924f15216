#!/bin/bash
set -e
mkdir -p gpurun_out/micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/mix_rate tools/micro/mix_rate.hip
timeout -k 10 120 /tmp/mix_rate > gpurun_out/micro/mix_rate.log 2>&1
cat gpurun_out/micro/mix_rate.log
