// tcp_rate.hip -- microbenchmark (tools only): what a wave-wide gather costs on gfx950 as a function of the number of distinct
// 128-byte lines its 64 lanes touch, for L1-resident and L2-resident working sets.  Build + run: tools/micro/run_tcp_rate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// every lane loads `BYTES` bytes at base + (lane / lanes_per_line) * 128 + (lane % lanes_per_line) * BYTES, and the window moves
// by `step` bytes per iteration inside a working set of `ws` bytes (per workgroup)
template <int BYTES>
__global__ __launch_bounds__(256) void k_gather(const char* __restrict__ buf, size_t ws, uint32_t lanes_per_line, uint32_t iters, uint32_t step, uint32_t* out, int shared) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const char* base = buf + (shared ? 0 : (size_t)blockIdx.x * ws);
    uint32_t off = ((lane / lanes_per_line) * 128u + (lane % lanes_per_line) * BYTES + wave * 2048u) % (uint32_t)(ws - 8192u);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (BYTES == 4) acc += *reinterpret_cast<const uint32_t*>(base + off);
            if (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2*>(base + off); acc += v.x ^ v.y; }
            if (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4*>(base + off); acc += v.x ^ v.y ^ v.z ^ v.w; }
            off += step;
            if (off >= (uint32_t)ws - 8192u) off -= (uint32_t)ws - 8192u;
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

int main() {
    const int nblk = 256 * 6;   // 6 workgroups of 4 waves per CU
    const size_t maxws = 1u << 20;
    char* buf; uint32_t* out;
    hipMalloc(&buf, (size_t)nblk * maxws + (1 << 20));
    hipMemset(buf, 1, (size_t)nblk * maxws + (1 << 20));
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t iters = 4096;
    printf("bytes lanes/line lines/instr  ws/WG   ms     cycles/instr/CU(2.4GHz)  lines/clk/CU  GB/s(requested)\n");
    for (int bytes : {4, 8, 16})
        for (int shared : {1, 0}) {
            const size_t ws = 16384;
            for (uint32_t lpl : {1u, 2u, 4u, 8u, 16u, 32u}) {
                if (lpl * bytes > 128) continue;
                const uint32_t lines = 64u / lpl;
                const uint32_t step = 1024u + 128u;   // a different set of lines every iteration
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (bytes == 4) hipLaunchKernelGGL(k_gather<4>, dim3(nblk), dim3(256), 0, 0, buf, ws, lpl, iters, step, out, shared);
                    if (bytes == 8) hipLaunchKernelGGL(k_gather<8>, dim3(nblk), dim3(256), 0, 0, buf, ws, lpl, iters, step, out, shared);
                    if (bytes == 16) hipLaunchKernelGGL(k_gather<16>, dim3(nblk), dim3(256), 0, 0, buf, ws, lpl, iters, step, out, shared);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep == 0) continue;
                    const double instr_per_cu = (double)nblk / 256.0 * 4.0 * iters;   // wave-instructions per CU
                    const double cyc = ms * 1e-3 * 2.4e9 / instr_per_cu;
                    printf("%5d %9u %11u %7s %7.3f %12.1f %22.2f %12.0f\n", bytes, lpl, lines, shared ? "L1" : "L2", ms, cyc, lines / cyc,
                           (double)nblk * 256 * iters * bytes / (ms * 1e-3) / 1e9);
                }
            }
        }
    return 0;
}
