// valu_rate.hip -- microbenchmark (tools only): issue cost of the instructions the march kernels are made of, on gfx950, as
// SIMD cycles per wave64 instruction with W waves per SIMD (W = 1, 2, 4, 8), independent operands (8 registers in rotation)
// and as a dependent chain.  Cycles come from s_memtime inside the kernel (shader clock), not from wall time.
// Build + run: tools/micro/run_valu_rate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// every op: r[i] = f(r[i], s)  on register i of 8 (independent) or always register 0 (dependent)
#define DEF_KERNEL(NAME, ASM_INDEP, ASM_DEP, EXTRA_DECL, ...)                                                          \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, unsigned long long* cyc, uint32_t iters, uint32_t seed, int dep) { \
        uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3u + 1u, r2 = r0 * 5u + 2u, r3 = r0 * 7u + 3u, r4 = r0 * 11u + 4u,               \
                 r5 = r0 * 13u + 5u, r6 = r0 * 17u + 6u, r7 = r0 * 19u + 7u;                                                           \
        uint32_t s = seed | 1u;                                                                                                        \
        EXTRA_DECL                                                                                                                     \
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(s));                   \
        __syncthreads();                                                                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                    \
        if (!dep) {                                                                                                                    \
            for (uint32_t it = 0; it < iters; it++) {                                                                                  \
                asm volatile(ASM_INDEP : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s) : __VA_ARGS__); \
            }                                                                                                                          \
        } else {                                                                                                                       \
            for (uint32_t it = 0; it < iters; it++) {                                                                                  \
                asm volatile(ASM_DEP : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s) : __VA_ARGS__); \
            }                                                                                                                          \
        }                                                                                                                              \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                                    \
        if ((threadIdx.x & 63u) == 0u) cyc[blockIdx.x * 4u + (threadIdx.x >> 6)] = t1 - t0;                                            \
        const uint32_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                                                      \
        if (x == 0x12345u) out[threadIdx.x] = x;                                                                                       \
    }

// 64 instructions per asm block: 8 registers x 8
#define I8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define I64(OP) I8(OP) I8(OP) I8(OP) I8(OP) I8(OP) I8(OP) I8(OP) I8(OP)
#define D64(OP) OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") \
                OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") \
                OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") \
                OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0") OP("%0")

#define OP_ADD(R) "v_add_u32 " R ", " R ", %8\n"
#define OP_AND(R) "v_and_b32 " R ", " R ", %8\n"
#define OP_XOR(R) "v_xor_b32 " R ", " R ", %8\n"
#define OP_LSHL(R) "v_lshlrev_b32 " R ", 1, " R "\n"
#define OP_ALIGNBIT(R) "v_alignbit_b32 " R ", " R ", %8, 31\n"
#define OP_BFE(R) "v_bfe_u32 " R ", " R ", 1, 31\n"
#define OP_LSHLOR(R) "v_lshl_or_b32 " R ", " R ", 1, %8\n"
#define OP_ANDOR(R) "v_and_or_b32 " R ", " R ", %8, %8\n"
#define OP_OR3(R) "v_or3_b32 " R ", " R ", %8, %8\n"
#define OP_ADD3(R) "v_add3_u32 " R ", " R ", %8, %8\n"
#define OP_BCNT(R) "v_bcnt_u32_b32 " R ", " R ", %8\n"
#define OP_FFBL(R) "v_ffbl_b32 " R ", " R "\n"
#define OP_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define OP_CMPF(R) "v_cmp_lt_f32 vcc, " R ", %8\n"
#define OP_CMPF_S(R) "v_cmp_lt_f32 s[20:21], " R ", %8\n"
#define OP_CMPU(R) "v_cmp_lt_u32 vcc, " R ", %8\n"
#define OP_MULLO(R) "v_mul_lo_u32 " R ", " R ", %8\n"
#define OP_MUL24(R) "v_mul_u32_u24 " R ", " R ", %8\n"
#define OP_MAD24(R) "v_mad_u32_u24 " R ", " R ", %8, %8\n"
#define OP_DPPSHR(R) "v_add_u32_dpp " R ", " R ", " R " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_DPPMOV(R) "v_mov_b32_dpp " R ", " R " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_DPPBC(R) "v_mov_b32_dpp " R ", " R " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define OP_READLANE(R) "v_readlane_b32 s20, " R ", 3\n"
#define OP_READFIRST(R) "v_readfirstlane_b32 s20, " R "\n"
#define OP_SUBF(R) "v_sub_f32 " R ", " R ", %8\n"
#define OP_MINF(R) "v_min_f32 " R ", |" R "|, %8\n"
#define OP_MIN3F(R) "v_min3_f32 " R ", |" R "|, |%8|, %8\n"
#define OP_FMA(R) "v_fma_f32 " R ", " R ", %8, %8\n"
#define OP_MBCNT(R) "v_mbcnt_lo_u32_b32 " R ", %8, " R "\n"
#define OP_PERM(R) "v_perm_b32 " R ", " R ", %8, %8\n"
#define OP_MOV(R) "v_mov_b32 " R ", %8\n"
#define OP_RCP(R) "v_rcp_f32 " R ", " R "\n"
#define OP_SAND(R) "s_and_b64 s[20:21], s[20:21], s[22:23]\n"
#define OP_SBCNT(R) "s_bcnt1_i32_b64 s20, s[22:23]\n"
#define OP_SADD(R) "s_add_u32 s20, s20, s22\n"
#define OP_SLSHL(R) "s_lshl_b64 s[20:21], s[20:21], 1\n"
#define OP_SFF1(R) "s_ff1_i32_b64 s20, s[22:23]\n"
#define OP_BPERM(R) "ds_bpermute_b32 " R ", %8, " R "\n"
#define OP_SWZ(R) "ds_swizzle_b32 " R ", " R " offset:0x041F\n"
#define OP_PERMLANE(R) "v_permlane32_swap " R ", " R "\n"

#define OP_CMPCND(R) "v_cmp_lt_u32 vcc, " R ", %8\nv_cndmask_b32 " R ", " R ", %8, vcc\n"
#define OP_CMPCND_S(R) "v_cmp_lt_u32 s[20:21], " R ", %8\nv_cndmask_b32_e64 " R ", " R ", %8, s[20:21]\n"
#define OP_CND_S(R) "v_cndmask_b32_e64 " R ", " R ", %8, s[20:21]\n"
#define OP_CND_IMM(R) "v_cndmask_b32_e64 " R ", 0, 1, s[20:21]\n"
#define OP_BFI(R) "v_bfi_b32 " R ", %8, " R ", %8\n"
#define OP_BITOP3(R) "v_bitop3_b32 " R ", " R ", %8, %8 bitop3:0xe0\n"
#define OP_MAXU(R) "v_max_u32 " R ", " R ", %8\n"
#define OP_ADDCO(R) "v_add_co_u32 " R ", vcc, " R ", %8\n"
#define OP_MAD64(R) "v_mad_u64_u32 v[20:21], s[20:21], " R ", %8, v[20:21]\n"
#define OP_LSHLADD64(R) "v_lshl_add_u64 v[20:21], v[20:21], 2, v[22:23]\n"
#define OP_WRITELANE(R) "v_writelane_b32 " R ", s20, 5\n"
#define OP_LSHRREV(R) "v_lshrrev_b32 " R ", 3, " R "\n"
#define OP_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 2, %8\n"
#define OP_SUBREV(R) "v_subrev_u32 " R ", %8, " R "\n"
#define OP_OR(R) "v_or_b32 " R ", " R ", %8\n"
#define OP_CVT(R) "v_cvt_f32_u32 " R ", " R "\n"
#define OP_MULF(R) "v_mul_f32 " R ", " R ", %8\n"
#define OP_DPPWSHL(R) "v_mov_b32_dpp " R ", " R " wave_shl:1 row_mask:0xf bank_mask:0xf\n"
#define NOCLOB "memory"
DEF_KERNEL(add, I64(OP_ADD), D64(OP_ADD), , NOCLOB)
DEF_KERNEL(and, I64(OP_AND), D64(OP_AND), , NOCLOB)
DEF_KERNEL(xor, I64(OP_XOR), D64(OP_XOR), , NOCLOB)
DEF_KERNEL(lshl, I64(OP_LSHL), D64(OP_LSHL), , NOCLOB)
DEF_KERNEL(alignbit, I64(OP_ALIGNBIT), D64(OP_ALIGNBIT), , NOCLOB)
DEF_KERNEL(bfe, I64(OP_BFE), D64(OP_BFE), , NOCLOB)
DEF_KERNEL(lshl_or, I64(OP_LSHLOR), D64(OP_LSHLOR), , NOCLOB)
DEF_KERNEL(and_or, I64(OP_ANDOR), D64(OP_ANDOR), , NOCLOB)
DEF_KERNEL(or3, I64(OP_OR3), D64(OP_OR3), , NOCLOB)
DEF_KERNEL(add3, I64(OP_ADD3), D64(OP_ADD3), , NOCLOB)
DEF_KERNEL(bcnt, I64(OP_BCNT), D64(OP_BCNT), , NOCLOB)
DEF_KERNEL(ffbl, I64(OP_FFBL), D64(OP_FFBL), , NOCLOB)
DEF_KERNEL(cndmask, I64(OP_CNDMASK), D64(OP_CNDMASK), , "vcc", NOCLOB)
DEF_KERNEL(cmp_f32_vcc, I64(OP_CMPF), D64(OP_CMPF), , "vcc", NOCLOB)
DEF_KERNEL(cmp_f32_sgpr, I64(OP_CMPF_S), D64(OP_CMPF_S), , "s20", "s21", NOCLOB)
DEF_KERNEL(cmp_u32_vcc, I64(OP_CMPU), D64(OP_CMPU), , "vcc", NOCLOB)
DEF_KERNEL(mul_lo, I64(OP_MULLO), D64(OP_MULLO), , NOCLOB)
DEF_KERNEL(mul_u24, I64(OP_MUL24), D64(OP_MUL24), , NOCLOB)
DEF_KERNEL(mad_u24, I64(OP_MAD24), D64(OP_MAD24), , NOCLOB)
DEF_KERNEL(dpp_add_shr, I64(OP_DPPSHR), D64(OP_DPPSHR), , NOCLOB)
DEF_KERNEL(dpp_mov_shr, I64(OP_DPPMOV), D64(OP_DPPMOV), , NOCLOB)
DEF_KERNEL(dpp_mov_bcast, I64(OP_DPPBC), D64(OP_DPPBC), , NOCLOB)
DEF_KERNEL(readlane, I64(OP_READLANE), D64(OP_READLANE), , "s20", NOCLOB)
DEF_KERNEL(readfirstlane, I64(OP_READFIRST), D64(OP_READFIRST), , "s20", NOCLOB)
DEF_KERNEL(sub_f32, I64(OP_SUBF), D64(OP_SUBF), , NOCLOB)
DEF_KERNEL(min_f32_abs, I64(OP_MINF), D64(OP_MINF), , NOCLOB)
DEF_KERNEL(min3_f32_abs, I64(OP_MIN3F), D64(OP_MIN3F), , NOCLOB)
DEF_KERNEL(fma_f32, I64(OP_FMA), D64(OP_FMA), , NOCLOB)
DEF_KERNEL(mbcnt, I64(OP_MBCNT), D64(OP_MBCNT), , NOCLOB)
DEF_KERNEL(perm_b32, I64(OP_PERM), D64(OP_PERM), , NOCLOB)
DEF_KERNEL(mov, I64(OP_MOV), D64(OP_MOV), , NOCLOB)
DEF_KERNEL(rcp_f32, I64(OP_RCP), D64(OP_RCP), , NOCLOB)
DEF_KERNEL(s_and_b64, I64(OP_SAND), D64(OP_SAND), , "s20", "s21", "s22", "s23", NOCLOB)
DEF_KERNEL(s_bcnt1, I64(OP_SBCNT), D64(OP_SBCNT), , "s20", "s22", "s23", "scc", NOCLOB)
DEF_KERNEL(s_add, I64(OP_SADD), D64(OP_SADD), , "s20", "s22", "scc", NOCLOB)
DEF_KERNEL(s_lshl_b64, I64(OP_SLSHL), D64(OP_SLSHL), , "s20", "s21", "scc", NOCLOB)
DEF_KERNEL(s_ff1_b64, I64(OP_SFF1), D64(OP_SFF1), , "s20", "s22", "s23", NOCLOB)
DEF_KERNEL(cmp_cndmask_vcc_pair, I64(OP_CMPCND), D64(OP_CMPCND), , "vcc", NOCLOB)
DEF_KERNEL(cmp_cndmask_sgpr_pair, I64(OP_CMPCND_S), D64(OP_CMPCND_S), , "s20", "s21", NOCLOB)
DEF_KERNEL(cndmask_sgpr, I64(OP_CND_S), D64(OP_CND_S), , "s20", "s21", NOCLOB)
DEF_KERNEL(cndmask_imm, I64(OP_CND_IMM), D64(OP_CND_IMM), , "s20", "s21", NOCLOB)
DEF_KERNEL(bfi, I64(OP_BFI), D64(OP_BFI), , NOCLOB)
DEF_KERNEL(bitop3, I64(OP_BITOP3), D64(OP_BITOP3), , NOCLOB)
DEF_KERNEL(max_u32, I64(OP_MAXU), D64(OP_MAXU), , NOCLOB)
DEF_KERNEL(add_co, I64(OP_ADDCO), D64(OP_ADDCO), , "vcc", NOCLOB)
DEF_KERNEL(mad_u64_u32, I64(OP_MAD64), D64(OP_MAD64), , "v20", "v21", "s20", "s21", NOCLOB)
DEF_KERNEL(lshl_add_u64, I64(OP_LSHLADD64), D64(OP_LSHLADD64), , "v20", "v21", "v22", "v23", NOCLOB)
DEF_KERNEL(writelane, I64(OP_WRITELANE), D64(OP_WRITELANE), , "s20", NOCLOB)
DEF_KERNEL(lshrrev, I64(OP_LSHRREV), D64(OP_LSHRREV), , NOCLOB)
DEF_KERNEL(lshl_add_u32, I64(OP_LSHLADD), D64(OP_LSHLADD), , NOCLOB)
DEF_KERNEL(subrev_u32, I64(OP_SUBREV), D64(OP_SUBREV), , NOCLOB)
DEF_KERNEL(or, I64(OP_OR), D64(OP_OR), , NOCLOB)
DEF_KERNEL(cvt_f32_u32, I64(OP_CVT), D64(OP_CVT), , NOCLOB)
DEF_KERNEL(mul_f32, I64(OP_MULF), D64(OP_MULF), , NOCLOB)
DEF_KERNEL(dpp_wave_shl, I64(OP_DPPWSHL), D64(OP_DPPWSHL), , NOCLOB)
DEF_KERNEL(ds_bpermute, I64(OP_BPERM), D64(OP_BPERM) , , NOCLOB)
DEF_KERNEL(ds_swizzle, I64(OP_SWZ), D64(OP_SWZ), , NOCLOB)

// LDS: ds_write_b32 / ds_read_b32 / ds_read_b64 with lane-linear addresses
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, unsigned long long* cyc, uint32_t iters, uint32_t seed, int mode) {
    __shared__ uint32_t sm[4][64 * 9];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* p = &sm[wave][0];
    for (uint32_t i = lane; i < 64 * 9; i += 64) p[i] = i + seed;
    __syncthreads();
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (mode == 0) { p[lane + 64 * (u & 7)] = acc + u; }
            else if (mode == 1) { acc += p[lane + 64 * (u & 7)]; }
            else if (mode == 2) { const uint2 v = *reinterpret_cast<const uint2*>(&p[2 * lane + 128 * (u & 3)]); acc += v.x ^ v.y; }
            else if (mode == 3) { acc += p[(lane * 9 + u) % (64 * 9)]; }                 // stride 9: conflict-free gather
            else if (mode == 4) { acc = p[(acc + lane) & 511u]; }                        // dependent read chain: latency
        }
        if (mode == 5) {   // 16 independent reads, ONE wait: throughput
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = p[lane + 64 * (u & 7)];
#pragma unroll
            for (int u = 0; u < 16; u++) acc ^= v[u];
        }
        if (mode == 6) {   // 16 independent writes
#pragma unroll
            for (int u = 0; u < 16; u++) p[lane + 64 * (u & 7)] = acc + u;
        }
        if (mode != 4) asm volatile("" : "+v"(acc) :: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0u) cyc[blockIdx.x * 4u + wave] = t1 - t0;
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

typedef void (*kfn)(uint32_t*, unsigned long long*, uint32_t, uint32_t, int);
struct entry { const char* name; kfn f; int per_iter; };

int main() {
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, 256 * 8 * 4 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(256 * 8 * 4);
#define E(N) {#N, k_##N, (strstr(#N, "_pair") ? 128 : 64)}
    std::vector<entry> es = {E(add), E(and), E(xor), E(lshl), E(alignbit), E(bfe), E(lshl_or), E(and_or), E(or3), E(add3), E(bcnt), E(ffbl), E(cndmask),
                             E(cmp_f32_vcc), E(cmp_f32_sgpr), E(cmp_u32_vcc), E(mul_lo), E(mul_u24), E(mad_u24), E(dpp_add_shr), E(dpp_mov_shr), E(dpp_mov_bcast),
                             E(readlane), E(readfirstlane), E(sub_f32), E(min_f32_abs), E(min3_f32_abs), E(fma_f32), E(mbcnt), E(perm_b32), E(mov), E(rcp_f32),
                             E(cmp_cndmask_vcc_pair), E(cmp_cndmask_sgpr_pair), E(cndmask_sgpr), E(cndmask_imm), E(bfi), E(bitop3), E(max_u32), E(add_co), E(mad_u64_u32), E(lshl_add_u64), E(writelane), E(lshrrev), E(lshl_add_u32), E(subrev_u32), E(or), E(cvt_f32_u32), E(mul_f32), E(dpp_wave_shl),
                             E(s_and_b64), E(s_bcnt1), E(s_add), E(s_lshl_b64), E(s_ff1_b64), E(ds_bpermute), E(ds_swizzle)};
    const uint32_t iters = 512;
    printf("%-18s | SIMD cycles per wave64 instruction, independent operands, W waves/SIMD = 1 2 4 8 | dependent chain, W = 1 4\n", "op");
    auto run = [&](kfn f, int blocks_per_cu, int dep, int per_iter) {
        const int nblk = 256 * blocks_per_cu;
        f<<<nblk, 256>>>(out, cyc, iters, 12345u, dep);   // warm
        f<<<nblk, 256>>>(out, cyc, iters, 12345u, dep);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, (size_t)nblk * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<unsigned long long> v(h.begin(), h.begin() + nblk * 4);
        std::sort(v.begin(), v.end());
        const double med = (double)v[v.size() / 2];
        // a wave's cycles per instruction, divided by the waves that share its SIMD = SIMD cycles per instruction
        return med / ((double)iters * per_iter) / blocks_per_cu;
    };
    for (auto& e : es) {
        printf("%-18s |", e.name);
        for (int w : {1, 2, 4, 8}) printf(" %6.2f", run(e.f, w, 0, e.per_iter));
        printf(" |");
        for (int w : {1, 4}) printf(" %6.2f", run(e.f, w, 1, e.per_iter));
        printf("\n");
        fflush(stdout);
    }
    const char* ln[] = {"ds_write_b32", "ds_read_b32", "ds_read_b64", "ds_read_b32 s9", "ds_read dep chain", "ds_read_b32 x16 1 wait", "ds_write_b32 x16"};
    for (int mode = 0; mode < 7; mode++) {
        printf("%-18s |", ln[mode]);
        for (int w : {1, 2, 4, 8}) printf(" %6.2f", run((kfn)k_lds, w, mode, 16));
        printf("   (CU-wide LDS: per SIMD; x1/4 = per CU)\n");
    }
    return 0;
}
