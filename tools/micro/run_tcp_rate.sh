#!/bin/bash
# builds and runs the gather-rate microbenchmark on the GPU box (tools only; nothing of the product links it)
set -e
mkdir -p gpurun_out/micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/tcp_rate tools/micro/tcp_rate.hip
timeout -k 10 120 /tmp/tcp_rate > gpurun_out/micro/tcp_rate.log 2>&1
cat gpurun_out/micro/tcp_rate.log
