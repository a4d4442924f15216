// sector_fetch.hip -- microbenchmark (tools only): how many bytes does the L2 ask the fabric for when a wave gathers 8 bytes from a
// line it has never seen?  One kernel per load flavour (plain, nontemporal, agent-scope relaxed atomic load = sc1, system-scope =
// sc0 sc1); every lane reads 8 bytes from its own 128-byte line of a 1 GiB buffer.  Run under
//   rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --kernel-trace
// and compare the request sizes per kernel (tools/micro/run_sector_fetch.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t v2u __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const uint64_t* __restrict__ buf, uint32_t nlines, uint32_t iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        // a line per lane and iteration, never the same twice (odd multiplier: a permutation of the lines)
        const uint32_t line = (tid * iters + it) * 2654435761u % nlines;
        const uint64_t* p = buf + (size_t)line * 16u + (tid & 7u);      // 8 bytes somewhere in the first half of the line
        uint64_t v;
        if (MODE == 0) v = *p;
        else if (MODE == 1) { const v2u t = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(p)); v = (uint64_t)t.x | ((uint64_t)t.y << 32); }
        else if (MODE == 2) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 3) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else { uint32_t w = *reinterpret_cast<const uint32_t*>(p); v = w; }                          // 4-byte plain load
        acc += (uint32_t)v ^ (uint32_t)(v >> 32);
    }
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    uint64_t* buf; uint32_t* out;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&out, 4096);
    const uint32_t nlines = (uint32_t)(bytes / 128u), iters = 16, blocks = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"plain 8B", "nontemporal 8B", "agent-scope (sc1) 8B", "system-scope (sc0 sc1) 8B", "plain 4B"};
    for (int rep = 0; rep < 2; rep++)
        for (int m = 0; m < 5; m++) {
            hipEventRecord(e0);
            if (m == 0) k_gather<0><<<blocks, 256>>>(buf, nlines, iters, out);
            if (m == 1) k_gather<1><<<blocks, 256>>>(buf, nlines, iters, out);
            if (m == 2) k_gather<2><<<blocks, 256>>>(buf, nlines, iters, out);
            if (m == 3) k_gather<3><<<blocks, 256>>>(buf, nlines, iters, out);
            if (m == 4) k_gather<4><<<blocks, 256>>>(buf, nlines, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)blocks * 256 * iters;
            if (rep) printf("%-28s %8.3f ms  %6.2f G lines/s  (%.0f lines)\n", names[m], ms, n / (ms * 1e-3) / 1e9, n);
        }
    return 0;
}
