#!/bin/bash
# builds and runs the instruction-rate microbenchmark on the GPU box (tools only; nothing of the product links it)
set -e
mkdir -p gpurun_out/micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o /tmp/valu_rate tools/micro/valu_rate.hip
timeout -k 10 300 /tmp/valu_rate > gpurun_out/micro/valu_rate.log 2>&1
cat gpurun_out/micro/valu_rate.log
