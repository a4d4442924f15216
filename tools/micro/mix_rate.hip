// mix_rate.hip -- microbenchmark (tools only): do scattered 8-byte gathers and coalesced 8-byte stores of one wave get in each
// other's way on gfx950?  Per iteration a wave issues NG gathers (64 distinct 128-byte lines each, L2-resident working set)
// and NS coalesced stores (nontemporal or plain), in three arrangements.  Build + run: tools/micro/run_mix_rate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
template <int NG, int NS, bool NT, bool WAIT>
__global__ __launch_bounds__(256) void k_mix(const char* __restrict__ src, char* __restrict__ dst, uint32_t iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * 65536u;
    char* obase = dst + ((size_t)blockIdx.x * 4u + wave) * (size_t)iters * NS * 512u;
    uint32_t off = lane * 128u + wave * 16384u, acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint2 v[NG > 0 ? NG : 1];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            v[g] = *reinterpret_cast<const uint2*>(base + ((off + g * 8192u + it * 136u) & 65535u & ~7u));
        }
        if (WAIT) {   // use the loaded data before the stores (the wave waits for its gathers, then stores)
#pragma unroll
            for (int g = 0; g < NG; g++) acc += v[g].x ^ v[g].y;
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
            v2u* p = reinterpret_cast<v2u*>(obase + ((size_t)it * NS + s) * 512u + lane * 8u);
            const v2u val = {acc + (uint32_t)s, lane};
            if (NT) __builtin_nontemporal_store(val, p); else *p = val;
        }
        if (!WAIT) {
#pragma unroll
            for (int g = 0; g < NG; g++) acc += v[g].x ^ v[g].y;
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
template <int NG, int NS, bool NT, bool WAIT>
static void run(const char* name, const char* src, char* dst, uint32_t* out) {
    const int nblk = 256 * 6; const uint32_t iters = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mix<NG, NS, NT, WAIT>), dim3(nblk), dim3(256), 0, 0, src, dst, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double per_cu_iter = ms * 1e-3 * 2.4e9 / (nblk / 256.0 * 4.0 * iters);
    printf("%-44s %8.3f ms  %8.1f cycles per wave-iteration per CU   stores %.0f GB/s\n", name, ms, per_cu_iter, (double)nblk * 4 * iters * NS * 512 / (ms * 1e-3) / 1e9);
}
int main() {
    char *src, *dst; uint32_t* out;
    hipMalloc(&src, (size_t)1536 * 65536 + 65536); hipMemset(src, 1, (size_t)1536 * 65536 + 65536);
    hipMalloc(&dst, (size_t)1536 * 4 * 256 * 6 * 512 + 65536);
    hipMalloc(&out, 4096);
    run<4, 0, true, true>("4 gathers", src, dst, out);
    run<0, 6, true, true>("6 nt stores", src, dst, out);
    run<0, 6, false, true>("6 plain stores", src, dst, out);
    run<4, 6, true, true>("4 gathers, wait, 6 nt stores", src, dst, out);
    run<4, 6, false, true>("4 gathers, wait, 6 plain stores", src, dst, out);
    run<4, 6, true, false>("4 gathers, 6 nt stores, then wait", src, dst, out);
    run<4, 6, false, false>("4 gathers, 6 plain stores, then wait", src, dst, out);
    run<4, 3, true, true>("4 gathers, wait, 3 nt stores", src, dst, out);
    return 0;
}
