// cx_march4d.hip -- Level-0 kernels of the 4-D marching-pentatopes hyper-voxel march (gfx950, wave64).
//
// Reference semantics restated (paths relative to the reference checkout, contourist/...):
//   PENTATOPES / HYPERCUBE                       pentatopes.py:15-30
//   enumerate_pentatope_tetrahedra                pentatopes.py:223-291
//   border_voxel with box = HYPERCUBE             tetrahedral.py:383-394, pentatopes.py:94
//   contour_pair_interpolation                    tetrahedral.py:471-487
//
// Same plan as the 3-D march (cx_march3d.hip): a lattice cell q = (i,j,k,l) owns the 15 edges q -> q+d,
// d = 8di+4dj+2dk+dl in 1..15, and the 24 pentatopes of its hyper-voxel; vertex id = (lin(q) << 4) | d.
//   K0  cx_k_signbits4    one streaming pass over the samples: sign bitmap (1 bit per sample).
//   K1a cx_k_queue4       active cells (sign change among the 16 corners) from the bitmap, 32 per lane -> one flat queue.
//   K1b cx_k_cells4       flat over the queue: exact classification, ONE reservation per workgroup, vertex records
//                         (x,y,z,t), edge ids, the per-cell lookup word and one record per active cell.
//   K2  cx_k_emit_tets    one lane per record, then one lane per tetrahedron: 24 pentatopes -> 4 vertex indices each.
#include <cstdlib>

#include "cx_cell.h"
#include "cx_state4.h"
#include "cx_tables4d.h"

__device__ constexpr uint8_t CX_PC[24][5] = CX_PENT_CORNERS_INIT;
__device__ const uint8_t cx_d_pent_corners[24][5] = CX_PENT_CORNERS_INIT;   // the same for run-time indices
// inclusive prefix sum over the wave (DPP row shifts / broadcasts)
__device__ __forceinline__ uint32_t cx_wave_incl_scan4(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return x;
}
#define CX_PENT_MASK(n) (uint32_t)((1u << CX_PC[n][0]) | (1u << CX_PC[n][1]) | (1u << CX_PC[n][2]) | (1u << CX_PC[n][3]) | (1u << CX_PC[n][4]))

__device__ __forceinline__ void cx_unravel4(const cx_params4& P, uint32_t lin, uint32_t q[4]) {
    q[0] = cx_div(lin, P.div3);
    uint32_t r = lin - q[0] * (P.n1 * P.n2 * P.n3);
    q[1] = cx_div(r, P.div2);
    r -= q[1] * (P.n2 * P.n3);
    q[2] = cx_div(r, P.div1);
    q[3] = r - q[2] * P.n3;
}

// 16 corners (clamped onto the array); vm = validity mask (bit c: corner c inside the array)
__device__ __forceinline__ uint32_t cx_load_corners4(const cx_params4& P, uint32_t lin, const uint32_t q[4], float f[16]) {
    const float* __restrict__ A = P.grid;
    const bool v0 = q[0] + 1 < P.n0, v1 = q[1] + 1 < P.n1, v2 = q[2] + 1 < P.n2, v3 = q[3] + 1 < P.n3;
    const uint32_t o0 = v0 ? P.n1 * P.n2 * P.n3 : 0u, o1 = v1 ? P.n2 * P.n3 : 0u, o2 = v2 ? P.n3 : 0u;
    const uint32_t base = v3 ? lin : lin - 1u;   // at the array edge in l read (l-1, l) and repeat l
    cx_f2 raw[8];      // the eight pairs are requested before any is looked at (a select right behind a load made the compiler wait there)
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t ofs = ((c & 4u) ? o0 : 0u) + ((c & 2u) ? o1 : 0u) + ((c & 1u) ? o2 : 0u);
        raw[c] = *reinterpret_cast<const cx_f2*>(A + base + ofs);
    }
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        f[2 * c] = v3 ? raw[c].x : raw[c].y;
        f[2 * c + 1] = raw[c].y;
    }
    uint32_t vm = 0;
#pragma unroll
    for (uint32_t c = 0; c < 16; c++) {
        const bool ok = (!(c & 8u) || v0) && (!(c & 4u) || v1) && (!(c & 2u) || v2) && (!(c & 1u) || v3);
        vm |= ok ? (1u << c) : 0u;
    }
    return vm;
}

__device__ __forceinline__ uint32_t cx_pent_pattern(uint32_t sm, int n) {
    return ((sm >> CX_PC[n][0]) & 1u) | (((sm >> CX_PC[n][1]) & 1u) << 1) | (((sm >> CX_PC[n][2]) & 1u) << 2) |
           (((sm >> CX_PC[n][3]) & 1u) << 3) | (((sm >> CX_PC[n][4]) & 1u) << 4);
}
// tetrahedra a pentatope with `nlow` low vertices emits (pentatopes.py:246-291)
__device__ __forceinline__ uint32_t cx_pent_ntets(uint32_t nlow) {
    return (nlow == 1u || nlow == 4u) ? 1u : ((nlow == 2u || nlow == 3u) ? 3u : 0u);
}

// exact tolerance handling of one active cell: which pentatopes the reference skips, which owned
// crossings survive (used by some emitted tetrahedron)
struct cx_cell4 {
    uint32_t emask, ntets, pskip, border;
};

static __device__ __forceinline__ bool cx_edge_used_slow4(const cx_params4& P, const uint32_t q[4], uint32_t d) {
    const float* A = P.grid;
    for (uint32_t o = 0; o < 16; o++) {
        if (o & d) continue;
        const uint32_t oq[4] = {(o >> 3) & 1u, (o >> 2) & 1u, (o >> 1) & 1u, o & 1u};
        if (q[0] < oq[0] || q[1] < oq[1] || q[2] < oq[2] || q[3] < oq[3]) continue;
        const uint32_t p[4] = {q[0] - oq[0], q[1] - oq[1], q[2] - oq[2], q[3] - oq[3]};
        if (p[0] + 1 >= P.n0 || p[1] + 1 >= P.n1 || p[2] + 1 >= P.n2 || p[3] + 1 >= P.n3) continue;
        uint32_t near_a = 0, all_b = 1;
        for (uint32_t c = 0; c < 16; c++) {
            const size_t idx = (((size_t)(p[0] + ((c >> 3) & 1u)) * P.n1 + (p[1] + ((c >> 2) & 1u))) * P.n2 + (p[2] + ((c >> 1) & 1u))) * P.n3 +
                               (p[3] + (c & 1u));
            const double fv = (double)A[idx];
            if (fabs(fv - P.value) <= P.tol_value) near_a |= 1u << c;
            if (!cx_near_b(fv, P.value)) all_b = 0;
        }
        if (all_b) continue;
        const uint32_t c1 = o, c2 = o | d;
        for (int n = 0; n < 24; n++) {
            const uint32_t pm = CX_PENT_MASK(n);
            if (((pm >> c1) & 1u) && ((pm >> c2) & 1u) && (near_a & pm) != pm) return true;
        }
    }
    return false;
}

__device__ __forceinline__ cx_cell4 cx_classify_cell4(const cx_params4& P, const float f[16], uint32_t vm, uint32_t sm,
                                                      const uint32_t q[4]) {
    cx_cell4 R;
    R.pskip = 0; R.ntets = 0; R.border = 0;
    const uint32_t s0 = (sm & 1u) ? 0xFFFFu : 0u;
    R.emask = ((sm ^ s0) & vm) & 0xFFFEu;
    float dmin = fabsf(f[0] - P.vcmp);
#pragma unroll
    for (int c = 1; c < 16; c++) dmin = fminf(dmin, fabsf(f[c] - P.vcmp));
    const bool real_voxel = (vm == 0xFFFFu);
    if (dmin > P.near_abs) {
        if (real_voxel) {
            R.border = 1;
#pragma unroll
            for (int n = 0; n < 24; n++) R.ntets += cx_pent_ntets(__popc(cx_pent_pattern(sm, n)));
        } else {
            R.pskip = 0xFFFFFFu;
        }
        return R;
    }
    uint32_t near_a = 0, nb = 0;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const double fc = (double)f[c];
        near_a |= (fabs(fc - P.value) <= P.tol_value) ? (1u << c) : 0u;
        nb |= cx_near_b(fc, P.value) ? (1u << c) : 0u;
    }
    if (real_voxel && nb != 0xFFFFu) {
        R.border = 1;
        for (int n = 0; n < 24; n++) {
            const uint32_t pm = CX_PENT_MASK(n);
            if ((near_a & pm) == pm) R.pskip |= 1u << n;
            else R.ntets += cx_pent_ntets(__popc(cx_pent_pattern(sm, n)));
        }
    } else {
        R.pskip = 0xFFFFFFu;
    }
    if (R.emask && ((near_a & 1u) || (nb & 1u))) {
        for (uint32_t d = 1; d < 16; d++) {
            if (!((R.emask >> d) & 1u)) continue;
            const bool suspicious = (((near_a >> d) & near_a & 1u) | ((nb >> d) & nb & 1u)) != 0u;
            if (suspicious && !cx_edge_used_slow4(P, q, d)) R.emask &= ~(1u << d);
        }
    }
    return R;
}

#define CX4_QCAP 2048u   // one step of the queue kernel adds at most 64 x 32 cells
#define CX4_WCAP 128u    // ... and at most 64 bitmap words with an active cell

// ---- sign bitmap: one pass over the samples at streaming speed.  A wave takes 8 consecutive chunks of
// 64 samples of a row (8 loads in flight); a chunk's 64 comparison results are one ballot = 2 words.
__global__ __launch_bounds__(256) void cx_k_signbits4(const cx_params4 P, const uint32_t nchunk, const cx_fdiv div_chunk,
                                                      const uint32_t nchunks_total) {
    const uint32_t lane = cx_lane_id();
    const uint32_t wave0 = blockIdx.x * 4u + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* __restrict__ A = P.grid;
    // grid-stride over groups of 8 chunks (one wave per group and 262 k waves made the launch rate the limit)
    for (uint32_t c0 = wave0 * 8u; c0 < nchunks_total; c0 += gridDim.x * 32u) {
        float f[8];
        bool ok[8];
        uint32_t row[8], ch[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t c = min(c0 + u, nchunks_total - 1u);
            row[u] = (nchunk == 1u) ? c : cx_div(c, div_chunk);   // no 8 integer divisions per wave (they were 2/3 of its instructions)
            ch[u] = c - row[u] * nchunk;
            const uint32_t l = ch[u] * 64u + lane;
            ok[u] = l < P.n3;
            f[u] = A[(size_t)row[u] * P.n3 + min(l, P.n3 - 1u)];   // unconditional: all 8 loads in flight
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint64_t m = __ballot(ok[u] && f[u] < P.vcmp);
            if (c0 + u < nchunks_total && lane < 2u && 2u * ch[u] + lane < P.nw3)
                P.signbits[(size_t)row[u] * P.nw3 + 2u * ch[u] + lane] = lane ? (uint32_t)(m >> 32) : (uint32_t)m;
        }
    }
}

// The same for rows that are a whole number of bitmap words (n3 % 32 == 0) on a 16-byte aligned grid: the bitmap is then the
// flat array's, bit for sample.  A lane loads FOUR consecutive samples with one 16-byte load, four loads in flight (4 KB per
// wave); the four comparison bits of 8 neighbouring lanes are joined into a word with three DPP steps.
#ifndef CX4_SB_UNROLL
#define CX4_SB_UNROLL 8u     // 16-byte loads a lane has in flight (4: 0.121 ms = 4.4 TB/s on config 4)
#endif
__global__ __launch_bounds__(256) void cx_k_signbits4_flat(const cx_params4 P, const uint32_t nquads) {
    const uint32_t lane = cx_lane_id();
    const uint32_t wave0 = blockIdx.x * 4u + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typedef float cx_f4 __attribute__((ext_vector_type(4)));
    const cx_f4* __restrict__ A = reinterpret_cast<const cx_f4*>(P.grid);
    for (uint32_t q0 = wave0 * (64u * CX4_SB_UNROLL); q0 < nquads; q0 += gridDim.x * (256u * CX4_SB_UNROLL)) {   // uniform
        cx_f4 f[CX4_SB_UNROLL];
#pragma unroll
        for (uint32_t u = 0; u < CX4_SB_UNROLL; u++) f[u] = __builtin_nontemporal_load(A + min(q0 + u * 64u + lane, nquads - 1u));
#pragma unroll
        for (uint32_t u = 0; u < CX4_SB_UNROLL; u++) {
            uint32_t v = ((f[u].x < P.vcmp) ? 1u : 0u) | ((f[u].y < P.vcmp) ? 2u : 0u) | ((f[u].z < P.vcmp) ? 4u : 0u) | ((f[u].w < P.vcmp) ? 8u : 0u);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true) << 4;    // row_shl:1: lane + 1
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xF, 0xF, true) << 8;    // row_shl:2
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, true) << 16;   // row_shl:4
            const uint32_t quad = q0 + u * 64u + lane;
            if ((lane & 7u) == 0u && quad < nquads) P.signbits[quad >> 3] = v;
        }
    }
}

// ---- queue: the active cells (sign change among the 16 clamped corners) from the sign bitmap, 32 cells per lane with
// word-wide OR / AND over the 8 rows of a hyper-voxel and over (l, l+1).  Their linear indices go to ONE flat queue in
// global memory (a wave collects up to 2048 in LDS; one reservation per workgroup at the end, one per wave when its LDS
// queue fills up), so that the stages after this one are spread evenly over the chip however the surface is distributed
// over the grid (the first version processed its cells inside this kernel: a workgroup that sat on the surface
// processed thousands of cells while most others had none -- 0.23 of its 0.28 ms).
__global__ __launch_bounds__(256) void cx_k_queue4(const cx_params4 P, const uint32_t items_per_block) {
    __shared__ uint32_t s_queue[4][CX4_QCAP];
    // the bitmap words that queued cells since the last flush: item index, active cells, position of the first in the wave's stage --
    // what the tetrahedra kernel finds a neighbour cell's queue entry by (P.items), written once the stage's place in the queue is known
    __shared__ uint32_t s_wi[4][CX4_WCAP], s_wa[4][CX4_WCAP], s_wp[4][CX4_WCAP];
    __shared__ uint32_t s_tot[4];
    __shared__ uint32_t s_base;
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.counters_tb = 0ULL;     // the cells kernel (next on the stream) counts into it
    uint32_t* qq = s_queue[wave];
    uint32_t qn = 0, wn = 0;
    const uint32_t nitems = P.nrows * P.nw3;   // one item = one bitmap word = 32 cells of one row
    uint32_t gbase = blockIdx.x * items_per_block + wave * 64u;
    const uint32_t gend = min(blockIdx.x * items_per_block + items_per_block, nitems);
    bool streaming = gbase < gend;
    const uint32_t* __restrict__ W = P.signbits;
    const uint32_t s2 = P.nw3, s1 = P.n2 * P.nw3, s0 = P.n1 * P.n2 * P.nw3;
    for (;;) {
        while (streaming) {
            const uint32_t idx = gbase + lane;
            const bool in = idx < gend;
            const uint32_t ic = in ? idx : (nitems - 1u);
            const uint32_t row = cx_div(ic, P.div_w);
            const uint32_t lw = ic - row * P.nw3;
            const uint32_t i = cx_div(row, P.div_r2);
            const uint32_t r = row - i * (P.n1 * P.n2);
            const uint32_t j = cx_div(r, P.div_r1);
            const uint32_t k = r - j * P.n2;
            const uint32_t o0 = (i + 1u < P.n0) ? s0 : 0u, o1 = (j + 1u < P.n1) ? s1 : 0u, o2 = (k + 1u < P.n2) ? s2 : 0u;
            const bool more = lw + 1u < P.nw3;   // the row goes on in the next word
            // bit positions of this word that are lattice points, and the one whose l+1 does not exist
            const uint32_t left = P.n3 - lw * 32u;
            const uint32_t valid = (left >= 32u) ? 0xFFFFFFFFu : ((1u << left) - 1u);
            const uint32_t edge = (left <= 32u) ? (1u << (left - 1u)) : 0u;
            uint32_t any = 0, all = 0xFFFFFFFFu;
            // all 16 words are requested before the first is used, the second of each pair unconditionally (where the row ends it
            // re-reads the first: a conditional load made the compiler wait for every load at its use)
            uint32_t w8[8], n8[8];
            const uint32_t step = more ? 1u : 0u;
#pragma unroll
            for (uint32_t c = 0; c < 8; c++) {
                const uint32_t at = ic + ((c & 4u) ? o0 : 0u) + ((c & 2u) ? o1 : 0u) + ((c & 1u) ? o2 : 0u);
                w8[c] = W[at];
                n8[c] = W[at + step];
            }
#pragma unroll
            for (uint32_t c = 0; c < 8; c++) {
                const uint32_t w = w8[c];
                const uint32_t nx = more ? n8[c] : 0u;
                uint32_t sh = (w >> 1) | (nx << 31);          // the samples at l+1
                sh = (sh & ~edge) | (w & edge);               // clamped at the end of the row
                any |= w | sh;
                all &= w & sh;
            }
            uint32_t act = in ? (any & ~all & valid) : 0u;
            const uint32_t cnt = __popc(act);
            const uint32_t incl = cx_wave_incl_scan4(cnt);
            const uint32_t tot = (uint32_t)__shfl((int)incl, 63);
            const uint64_t wm = __ballot(act != 0u);
            const uint32_t nwords = (uint32_t)__popcll(wm);
            if (qn + tot > CX4_QCAP || wn + nwords > CX4_WCAP) break;   // wave-uniform: flush what is queued, then redo this step
            uint32_t pos = qn + incl - cnt;
            if (act) {
                const uint32_t r = wn + cx_mbcnt(wm);
                s_wi[wave][r] = ic; s_wa[wave][r] = act; s_wp[wave][r] = pos;
            }
            wn += nwords;
            const uint32_t lin0 = row * P.n3 + lw * 32u;
            while (act) {
                const uint32_t bit = __ffs(act) - 1u;
                act &= act - 1u;
                qq[pos++] = lin0 + bit;
            }
            qn += tot;
            gbase += 256u;
            streaming = gbase < gend;
        }
        const bool final_round = !streaming;
        uint32_t base;
        if (final_round) {
            if (lane == 0) s_tot[wave] = qn;
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t t = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
                s_base = t ? atomicAdd(&P.counters[CX4_CNT_QUEUE], t) : 0u;
            }
            __syncthreads();
            base = s_base;
            for (uint32_t w = 0; w < wave; w++) base += s_tot[w];
        } else {
            uint32_t b0 = 0;
            if (lane == 0) b0 = atomicAdd(&P.counters[CX4_CNT_QUEUE], qn);
            base = __builtin_amdgcn_readfirstlane(b0);
        }
        __builtin_amdgcn_wave_barrier();
        if (base + qn <= P.qcap) {
            for (uint32_t x = lane; x < qn; x += 64u) P.queue[base + x] = qq[x];
            for (uint32_t x = lane; x < wn; x += 64u) P.items[s_wi[wave][x]] = make_uint2(base + s_wp[wave][x], s_wa[wave][x]);
        }
        __builtin_amdgcn_wave_barrier();
        qn = 0; wn = 0;
        if (final_round) break;
    }
}

// ---- cells: one lane per queued cell, flat over the queue.  Exact classification (the reference's tolerances), ONE
// reservation per workgroup (two 64-bit adds: vertices | records, border voxels | tetrahedra), then the cell record, the
// lookup word, and the vertices -- one lane per VERTEX through an LDS slot table and an LDS corner table, so that a
// round's vertex records leave as a few coalesced 16-byte stores (the first version stored them from a per-cell loop
// over the 15 directions: up to 64 cache lines per store instruction).
#define CX4_CELLS_WAVES 8u
struct cx_cells_lds {
    float f[CX4_CELLS_WAVES][16][64];
    uint16_t slot[CX4_CELLS_WAVES][15 * 64];
    uint32_t tot[CX4_CELLS_WAVES][4];
    unsigned long long base[2];
};
__global__ __launch_bounds__(512) void cx_k_cells4(const cx_params4 P) {
    __shared__ cx_cells_lds L;
    const uint32_t nq = P.counters[CX4_CNT_QUEUE];
    if (nq > P.qcap) return;                      // the host grows the queue and runs again
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t per_block = CX4_CELLS_WAVES * 64u;
    for (uint32_t blk = blockIdx.x; blk * per_block < nq; blk += gridDim.x) {   // uniform over the workgroup
        const uint32_t idx = blk * per_block + threadIdx.x;
        const bool have = idx < nq;
        const uint32_t lin = have ? P.queue[idx] : 0u;
        uint32_t q[4];
        cx_unravel4(P, lin, q);
        float f[16];
        const uint32_t vm = cx_load_corners4(P, lin, q, f);
        uint32_t sm = 0;
#pragma unroll
        for (int c = 0; c < 16; c++) sm |= (f[c] < P.vcmp) ? (1u << c) : 0u;
        const uint32_t smv = sm & vm;
        const bool active = have && smv != 0u && smv != vm;
        cx_cell4 R;
        R.emask = 0; R.ntets = 0; R.pskip = 0; R.border = 0;
        if (active) R = cx_classify_cell4(P, f, vm, sm, q);
        const uint32_t nv = __popc(R.emask);
        const bool rec = active && (nv != 0u || R.ntets != 0u);
        uint32_t vtot, ttot;
        const uint32_t vpre = cx_wave_prefix_small<4>(nv, vtot);
        const uint32_t tpre = cx_wave_prefix_small<7>(R.ntets, ttot);
        const uint64_t recm = __ballot(rec);
        const uint32_t ctot = (uint32_t)__popcll(recm);
        const uint32_t btot = (uint32_t)__popcll(__ballot(R.border != 0u));
        if (lane == 0) { L.tot[wave][0] = vtot; L.tot[wave][1] = ttot; L.tot[wave][2] = ctot; L.tot[wave][3] = btot; }
        // corner table and slot table of this wave's round
#pragma unroll
        for (int c = 0; c < 16; c++) L.f[wave][c][lane] = f[c];
        {
            uint32_t m = R.emask, pos = vpre;
            while (m) {
                const uint32_t d = __ffs(m) - 1u;
                m &= m - 1u;
                L.slot[wave][pos++] = (uint16_t)((lane << 4) | d);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t v = 0, t = 0, c = 0, bb = 0;
            for (uint32_t w = 0; w < CX4_CELLS_WAVES; w++) { v += L.tot[w][0]; t += L.tot[w][1]; c += L.tot[w][2]; bb += L.tot[w][3]; }
            unsigned long long* C64 = reinterpret_cast<unsigned long long*>(P.counters);   // words (cells, verts), (tets, border)
            L.base[0] = (v | c) ? atomicAdd(&C64[0], ((unsigned long long)v << 32) | c) : 0ULL;
            L.base[1] = (t | bb) ? atomicAdd(P.counters_tb, ((unsigned long long)bb << 32) | t) : 0ULL;
        }
        __syncthreads();
        uint32_t vbase = (uint32_t)(L.base[0] >> 32), cbase = (uint32_t)L.base[0], tbase = (uint32_t)L.base[1];
        uint32_t vblock = 0, cblock = 0;
        for (uint32_t w = 0; w < CX4_CELLS_WAVES; w++) {
            if (w < wave) { vbase += L.tot[w][0]; tbase += L.tot[w][1]; cbase += L.tot[w][2]; }
            vblock += L.tot[w][0]; cblock += L.tot[w][2];
        }
        const bool vfits = (L.base[0] >> 32) + (unsigned long long)vblock <= (unsigned long long)P.vcap;
        const bool cfits = (L.base[0] & 0xFFFFFFFFULL) + (unsigned long long)cblock <= (unsigned long long)P.ccap;
        const uint32_t vfirst = vbase + vpre;
        // the round's records and tetrahedra are contiguous: the tetrahedra kernel works round by round
        if (lane == 0) P.rounds[idx >> 6] = make_uint4(cbase, cfits ? ctot : 0u, tbase, ttot);
        if (vfits && have) P.info[idx] = ((uint64_t)R.emask << 32) | (uint64_t)vfirst;     // one word per queue entry: 512 contiguous bytes per wave
        if (cfits && rec) {
            uint4 c4;
            c4.x = lin;
            c4.y = sm | ((R.pskip != 0u) ? 0x10000u : 0u);
            c4.z = tbase + tpre;
            c4.w = vfirst;
            P.cells[cbase + cx_mbcnt(recm)] = c4;
        }
        if (vfits) {
            for (uint32_t j0 = 0; j0 < vtot; j0 += 64u) {   // wave-uniform
                const uint32_t j = j0 + lane;
                const bool ok = j < vtot;
                const uint32_t w = L.slot[wave][ok ? j : 0u];
                const uint32_t cell = w >> 4, d = w & 15u;
                const float f0 = L.f[wave][0][cell], fd = L.f[wave][d][cell];
                const uint32_t clin = (uint32_t)__shfl((int)lin, (int)cell);
                uint32_t cq[4];
                cx_unravel4(P, clin, cq);
                const float num = (P.vhi - f0) + P.vlo;
                const float den = fd - f0;
                float t = cx_fraction(num, den);
                if (fabsf(den) <= 1.001e-8f) {
                    const double dd = (double)fd - (double)f0;
                    t = (fabs(dd) <= 1e-8) ? 0.5f : (float)((P.value - (double)f0) / dd);
                }
                float4 r4;
                r4.x = (float)cq[0] + ((d & 8u) ? t : 0.f);
                r4.y = (float)cq[1] + ((d & 4u) ? t : 0.f);
                r4.z = (float)cq[2] + ((d & 2u) ? t : 0.f);
                r4.w = (float)cq[3] + ((d & 1u) ? t : 0.f);
                if (ok) {
                    P.verts[vbase + j] = r4;
                    P.vkeys[vbase + j] = (clin << 4) | d;
                }
            }
        }
        __syncthreads();   // the tables are rewritten by the next round
    }
}

// ---- CPython set order of 4-tuples ------------------------------------------------------------------------
#define CX4_PY_P1 11400714785074694791ULL
#define CX4_PY_P2 14029467366897019727ULL
#define CX4_PY_P5 2870177450012600261ULL
// x is a lattice coordinate as CPython sees it: it may be negative (an array with a rim of samples around the reference's grid has
// its origin at -1); a small int hashes to itself, except hash(-1) == -2
__device__ __forceinline__ uint64_t py_round4(uint64_t acc, uint32_t x) {
    const int32_t sx = (int32_t)x;
    acc += ((sx == -1) ? ~1ULL : (uint64_t)(int64_t)sx) * CX4_PY_P2;
    acc = (acc << 31) | (acc >> 33);
    return acc * CX4_PY_P1;
}
__device__ __forceinline__ uint64_t py_finish4(uint64_t acc) {
    acc += 4ULL ^ (CX4_PY_P5 ^ 3527539ULL);
    return (acc == ~0ULL) ? 1546275796ULL : acc;
}
__global__ void cx_k_hash_xyz(uint64_t* table, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t o0, uint32_t o1, uint32_t o2) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n0 * n1 * n2) return;
    const uint32_t k = idx % n2, r = idx / n2, j = r % n1, i = r / n1;
    table[idx] = py_round4(py_round4(py_round4(CX4_PY_P5, i + o0), j + o1), k + o2);
}
// slots of up to 3 hashes inserted in order into a fresh 8-slot set (setobject.c, mask 7)
__device__ __forceinline__ void py_slots3(const uint64_t h[3], int m, uint32_t slot[3]) {
    uint32_t used = 0;
    for (int n = 0; n < m; n++) {
        uint64_t perturb = h[n];
        uint32_t i = (uint32_t)h[n] & 7u;
        for (int guard = 0; guard < 32 && ((used >> i) & 1u); guard++) {
            perturb >>= 5;
            i = (uint32_t)((i * 5u + 1u + perturb) & 7u);
        }
        used |= 1u << i;
        slot[n] = i;
    }
}

// the same from the LOW 32 bits of the hashes: a probe step reads three bits five places further up, so five steps fit.
// Returns false when an element is still on a used slot after five steps, or when a hash word is all ones (possibly the hash
// CPython replaces by a constant): the caller then takes the 64-bit path for the whole wave.
__device__ __forceinline__ bool py_slots3_lo(const uint32_t h[3], int m, uint32_t slot[3]) {
    uint32_t used = 0;
    bool ok = true;
    for (int n = 0; n < m; n++) {
        uint32_t perturb = h[n];
        uint32_t i = h[n] & 7u;
        ok = ok && (h[n] != 0xFFFFFFFFu);
#pragma unroll
        for (int guard = 0; guard < 5; guard++)
            if ((used >> i) & 1u) {
                perturb >>= 5;
                i = (i * 5u + 1u + perturb) & 7u;
            }
        ok = ok && !((used >> i) & 1u);
        used |= 1u << i;
        slot[n] = i;
    }
    return ok;
}

// =================================================================================================
// K2: tetrahedra.  Phase 0, one lane per cell record: the lookup words of the 15 owner corners go to LDS, the corner
// hashes stay in the lane's registers.  Then, for 4 groups of 6 pentatopes: phase 1 (per cell, pentatope loop unrolled:
// corners and masks are compile-time constants) decides pattern and set-order permutation of each pentatope and writes
// one slot word per tetrahedron; phase 2, one lane per TETRAHEDRON, takes the tetrahedron's four pentatope-local edges
// from the factored table in LDS (3 KB: [pattern][permutation]; the pentatope's corners are one word), turns them into
// (owner corner, direction) and stores 16 bytes next to its neighbours'.  (The first version read the unfactored 147 KB
// table from global memory inside phase 2 -- a dependent gather per tetrahedron round -- and kept the hashes in LDS.)
// =================================================================================================
__device__ uint64_t cx_d_pent_local[32][12] = CX_PENT_LOCAL_INIT;
#define CX_PCW(n) (uint32_t)(CX_PC[n][0] | (CX_PC[n][1] << 4) | (CX_PC[n][2] << 8) | (CX_PC[n][3] << 12) | (CX_PC[n][4] << 16))

// ---- set order from probe codes ---------------------------------------------------------------------------
// CPython inserts an element with hash h into an 8-slot set at the first free slot of the sequence i0 = h & 7,
// i_k = (5 i_(k-1) + 1 + (h >> 5k)) & 7 (setobject.c: no linear probes when the table has 8 slots).  The sequence
// depends on the element only, so it is computed ONCE per corner of the hyper-voxel -- CX4_PROBE_STEPS slots, three bits each, in one
// word -- and every pentatope that contains the corner (each corner is in 6..24 of them) reads its members' slots off
// the codes: XOR with the occupied slot replicated into all fields, first non-zero field.  (The first version ran the
// probing recurrence per pentatope and member: ~240 of the kernel's VALU instructions per pentatope, 60 % of the kernel.)
#ifndef CX4_PROBE_STEPS
#define CX4_PROBE_STEPS 6u       // slots of the probe sequence kept per corner (<= 10).  Six come out of the LOW word of the hash (bits 0-27); a
#endif                           // set whose members still collide after six probes (~4e-6 of them) takes the exact path (cx_pent_perm_exact)
constexpr uint32_t cx_rep_fields(uint32_t n) { return n ? ((cx_rep_fields(n - 1u) << 3) | 1u) : 0u; }
constexpr uint32_t CX4_REP = cx_rep_fields(CX4_PROBE_STEPS);      // bit 0 of each of the 3-bit fields (a constant: a call in an expression is compiled as one)
__device__ __forceinline__ uint32_t cx_probe_code(uint64_t h) {
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 30);
    uint32_t i = lo & 7u, code = i;
#pragma unroll
    for (uint32_t k = 1; k < CX4_PROBE_STEPS; k++) {
        const uint32_t p = (k <= 5u) ? (lo >> (5u * k)) : (hi >> (5u * (k - 6u)));
        i = (i * 5u + 1u + p) & 7u;
        code |= i << (3u * k);
    }
    return code;
}
// fields of x that are non-zero, as their bit 0
__device__ __forceinline__ uint32_t cx_nz_fields(uint32_t x) { return (x | (x >> 1) | (x >> 2)) & CX4_REP; }

// exact set order of one pentatope from the full hashes (when the codes ran out)
__device__ __forceinline__ uint32_t cx_pent_perm_exact(const cx_params4& P, const uint32_t q[4], uint32_t n, uint32_t pat) {
    const bool low_is_two = (__popc(pat) == 2);
    uint64_t h2[3] = {0, 0, 0}, h3[3] = {0, 0, 0};
    uint32_t s2[3], s3[3];
    int n2 = 0, n3 = 0;
    for (int m = 0; m < 5; m++) {
        const uint32_t c = cx_d_pent_corners[n][m];
        const uint32_t ci = min(q[0] + ((c >> 3) & 1u), P.n0 - 1u), cj = min(q[1] + ((c >> 2) & 1u), P.n1 - 1u);
        const uint32_t ck = min(q[2] + ((c >> 1) & 1u), P.n2 - 1u);
        const uint64_t hm = py_finish4(py_round4(P.hash_xyz[(ci * P.n1 + cj) * P.n2 + ck], q[3] + (c & 1u) + P.org[3]));
        const bool is_low = (pat >> m) & 1u;
        if (is_low == low_is_two) { if (n2 < 2) h2[n2] = hm; n2++; }
        else { if (n3 < 3) h3[n3] = hm; n3++; }
    }
    py_slots3(h2, 2, s2);
    py_slots3(h3, 3, s3);
    const uint32_t swapped = (s2[1] < s2[0]) ? 1u : 0u;
    const bool ab = s3[0] < s3[1], ac = s3[0] < s3[2], bc = s3[1] < s3[2];
    const uint32_t p3 = (ab && ac) ? (bc ? 0u : 1u) : ((!ab && bc) ? (ac ? 2u : 3u) : (ab ? 4u : 5u));
    return (p3 << 1) | swapped;
}

// pattern, set-order permutation and tetrahedron count of pentatope N of a cell; `unres`: the codes ran out
template <int N>
__device__ __forceinline__ void cx_pent_decide(const uint32_t (&S)[16], uint32_t sm, bool skip, bool emulate, uint32_t& pat_out,
                                               uint32_t& perm_out, uint32_t& nt_out, bool& unres) {
    pat_out = 0; perm_out = 0; nt_out = 0; unres = false;
    if (skip) return;
    const uint32_t pat = cx_pent_pattern(sm, N);
    const uint32_t nlow = __popc(pat);
    if (nlow == 0u || nlow == 5u) return;
    uint32_t perm_id = 0;
    if (emulate && (nlow == 2u || nlow == 3u)) {
        // the 2-set (A, B) and the 3-set (C, D, E), each in insertion (path) order
        const uint32_t m2 = (nlow == 2u) ? pat : (pat ^ 31u);
        const uint32_t S0 = S[CX_PC[N][0]], S1 = S[CX_PC[N][1]], S2 = S[CX_PC[N][2]], S3 = S[CX_PC[N][3]], S4 = S[CX_PC[N][4]];
        const bool c0 = m2 & 1u, c1 = m2 & 2u, c2 = m2 & 4u, c3 = m2 & 8u, c4 = m2 & 16u;
        const uint32_t A = c0 ? S0 : (c1 ? S1 : (c2 ? S2 : S3));
        const uint32_t B = c4 ? S4 : (c3 ? S3 : (c2 ? S2 : S1));
        const uint32_t C = !c0 ? S0 : (!c1 ? S1 : S2);
        const uint32_t E = !c4 ? S4 : (!c3 ? S3 : S2);
        const uint32_t D = S0 ^ S1 ^ S2 ^ S3 ^ S4 ^ A ^ B ^ C ^ E;
        const uint32_t sA = A & 7u, sC = C & 7u;
        const uint32_t zB = cx_nz_fields(B ^ (sA * CX4_REP));
        const uint32_t sB = (B >> (__ffs(zB) - 1)) & 7u;
        const uint32_t repC = sC * CX4_REP;
        const uint32_t zD = cx_nz_fields(D ^ repC);
        const uint32_t sD = (D >> (__ffs(zD) - 1)) & 7u;
        const uint32_t zE = cx_nz_fields(E ^ repC) & cx_nz_fields(E ^ (sD * CX4_REP));
        const uint32_t sE = (E >> (__ffs(zE) - 1)) & 7u;
        unres = (zB == 0u) || (zD == 0u) || (zE == 0u);
        const uint32_t swapped = (sB < sA) ? 1u : 0u;
        // iteration order of the 3-set as (first, second, third) insertion indices -> itertools.permutations index
        const uint32_t idx = ((sC < sD) ? 1u : 0u) | ((sC < sE) ? 2u : 0u) | ((sD < sE) ? 4u : 0u);   // ab | ac << 1 | bc << 2
        // idx: 7 -> 0, 3 -> 1, 6 -> 2, 4 -> 3, 1 -> 4, 0 -> 5 (2 and 5 cannot happen)
        const uint32_t p3 = (0x02031045u >> (4u * idx)) & 7u;
        perm_id = (p3 << 1) | swapped;
    }
    pat_out = pat; perm_out = perm_id;
    nt_out = (nlow == 2u || nlow == 3u) ? 3u : 1u;
}

#ifndef CX4_TETS_WAVES
#define CX4_TETS_WAVES 4
#endif
// LDS of the tetrahedra kernel (per workgroup of 4 waves)
struct cx_tet_lds {
    // [cell][corner], the cell slow: in phase 2 a run of ~30 consecutive lanes works on tetrahedra of ONE cell and reads different corners
    // of it -- with the cell fast ([corner][cell], until round 4) those all fell on bank (cell % 32): SQ_LDS_BANK_CONFLICT was two thirds of
    // the kernel's LDS cycles (26 of 40 M, a quarter of its time).  Rows of 15 words (odd: the cell lanes' writes spread over all banks too).
    uint32_t vf[4][64][15];        // first vertex of the cells at the 15 owner corners
    uint16_t em[4][64][17];        // their crossing masks (rows of 17 halfwords: the cell lanes' writes of one corner spread over the banks)
    uint16_t slot[4][18 * 64 + 64];// per group: cell lane | pentatope in group << 6 | tetrahedron of the entry << 9 (+ a dump row)
    uint16_t pinfo[4][64][6];      // per group: pattern | permutation id << 5
    uint64_t local[32 * 12];
};

template <int G>
__device__ __forceinline__ void cx_tets_group(const cx_params4& P, cx_tet_lds& L, uint32_t wave, uint32_t lane, const uint32_t (&S)[16],
                                              const uint32_t q[4], uint32_t sm, bool real_voxel, uint32_t pskip, bool emulate,
                                              uint32_t& tnext, const bool emit) {      // emit == false (wave-uniform): only count the group's tetrahedra
    // ---- phase 1: slot words of the group's tetrahedra
    uint32_t cnt = 0, unres = 0;
    uint32_t pats[6], perms[6], nts[6];
    bool u;
    cx_pent_decide<G * 6 + 0>(S, sm, !real_voxel || ((pskip >> (G * 6 + 0)) & 1u), emulate, pats[0], perms[0], nts[0], u); unres |= u ? 1u : 0u;
    cx_pent_decide<G * 6 + 1>(S, sm, !real_voxel || ((pskip >> (G * 6 + 1)) & 1u), emulate, pats[1], perms[1], nts[1], u); unres |= u ? 2u : 0u;
    cx_pent_decide<G * 6 + 2>(S, sm, !real_voxel || ((pskip >> (G * 6 + 2)) & 1u), emulate, pats[2], perms[2], nts[2], u); unres |= u ? 4u : 0u;
    cx_pent_decide<G * 6 + 3>(S, sm, !real_voxel || ((pskip >> (G * 6 + 3)) & 1u), emulate, pats[3], perms[3], nts[3], u); unres |= u ? 8u : 0u;
    cx_pent_decide<G * 6 + 4>(S, sm, !real_voxel || ((pskip >> (G * 6 + 4)) & 1u), emulate, pats[4], perms[4], nts[4], u); unres |= u ? 16u : 0u;
    cx_pent_decide<G * 6 + 5>(S, sm, !real_voxel || ((pskip >> (G * 6 + 5)) & 1u), emulate, pats[5], perms[5], nts[5], u); unres |= u ? 32u : 0u;
#pragma unroll
    for (int pn = 0; pn < 6; pn++) cnt += nts[pn];
    const uint32_t incl = cx_wave_incl_scan4(cnt);
    const uint32_t ttot = (uint32_t)__shfl((int)incl, 63);
    if (!emit) { tnext += ttot; return; }
    uint32_t pos = incl - cnt;
    const uint32_t dump = 18u * 64u + lane;      // where the slot words of absent tetrahedra go (no branches, no loops)
#pragma unroll
    for (int pn = 0; pn < 6; pn++) {
        L.pinfo[wave][lane][pn] = (uint16_t)(pats[pn] | (perms[pn] << 5));
        const uint32_t w = lane | ((uint32_t)pn << 6);
        L.slot[wave][nts[pn] ? pos : dump] = (uint16_t)w;
        L.slot[wave][nts[pn] == 3u ? pos + 1u : dump] = (uint16_t)(w | (1u << 9));
        L.slot[wave][nts[pn] == 3u ? pos + 2u : dump] = (uint16_t)(w | (2u << 9));
        pos += nts[pn];
    }
    // pentatopes whose probe codes ran out: exact order from the full hashes (one copy of the code per group)
    while (unres) {
        const uint32_t pn = __ffs(unres) - 1u;
        unres &= unres - 1u;
        const uint32_t pi = L.pinfo[wave][lane][pn];
        L.pinfo[wave][lane][pn] = (uint16_t)((pi & 31u) | (cx_pent_perm_exact(P, q, (uint32_t)G * 6u + pn, pi & 31u) << 5));
    }
    __builtin_amdgcn_wave_barrier();
    // ---- phase 2: one lane per tetrahedron
#ifdef CX4_ABL_P2   // timing experiments (tools/variants.sh + tools/ab4d.py): no phase 2 / CX4_ABL_STORE 1 no stores, 2 plain stores
    if (P.tcap != 12345u) return;
#endif
    for (uint32_t j0 = 0; j0 < ttot; j0 += 64u) {   // wave-uniform
        const uint32_t j = j0 + lane;
        const bool ok = j < ttot;
        const uint32_t w = L.slot[wave][ok ? j : 0u];
        const uint32_t cell = w & 63u, pn = (w >> 6) & 7u, k = (w >> 9) & 3u;
        const uint32_t pi = L.pinfo[wave][cell][pn];
        // corners of the pentatope (a nibble per local vertex) and its orientation: 6 candidates, compile-time words
        uint32_t pcw = CX_PCW(G * 6 + 0);
        pcw = (pn == 1u) ? CX_PCW(G * 6 + 1) : pcw;
        pcw = (pn == 2u) ? CX_PCW(G * 6 + 2) : pcw;
        pcw = (pn == 3u) ? CX_PCW(G * 6 + 3) : pcw;
        pcw = (pn == 4u) ? CX_PCW(G * 6 + 4) : pcw;
        pcw = (pn == 5u) ? CX_PCW(G * 6 + 5) : pcw;
        const bool flip = ((CX_PENT_FLIP_MASK >> (G * 6)) >> pn) & 1u;
        const uint64_t lw = L.local[(pi & 31u) * 12u + (pi >> 5)];
        const uint32_t t20 = (uint32_t)(lw >> (20u * k)) & 0xFFFFFu;   // 4 local edges: x (2 bits) | y (3 bits), x < y vertices of the pentatope
        int32_t tv[4];
#pragma unroll
        for (uint32_t s_ = 0; s_ < 4; s_++) {
            const uint32_t x = (t20 >> (5u * s_)) & 3u, y = (t20 >> (5u * s_ + 2u)) & 7u;
            const uint32_t c1 = (pcw >> (4u * x)) & 15u, c2 = (pcw >> (4u * y)) & 15u;
            const uint32_t d = c1 ^ c2;
            const uint32_t vf = L.vf[wave][cell][c1], em = L.em[wave][cell][c1];
            tv[s_] = (int32_t)(vf + __popc(em & ((1u << d) - 1u)));
        }
#ifndef CX4_ABL_STORE
#define CX4_ABL_STORE 0
#endif
        if (ok && (CX4_ABL_STORE != 1 || tv[0] == 0x7FFFFFF)) {   // not read again by the pipeline: nontemporal (see cx_march3d.hip)
            typedef int32_t cx_v4i __attribute__((ext_vector_type(4)));
            cx_v4i* dst = reinterpret_cast<cx_v4i*>(P.tets + (size_t)(tnext + j) * 4u);   // the round's tetrahedra, group after group
            const cx_v4i val = cx_v4i{tv[0], tv[1], flip ? tv[3] : tv[2], flip ? tv[2] : tv[3]};
            if (CX4_ABL_STORE == 2) *dst = val; else __builtin_nontemporal_store(val, dst);
        }
    }
    tnext += ttot;
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256, CX4_TETS_WAVES) void cx_k_emit_tets(const cx_params4 P) {
    __shared__ cx_tet_lds L;
    for (uint32_t x = threadIdx.x; x < 32u * 12u; x += 256u) L.local[x] = P.lut[x];
    __syncthreads();
    const uint32_t nq = P.counters[CX4_CNT_QUEUE];
    const unsigned long long tbw = *P.counters_tb;
    if (blockIdx.x == 0 && threadIdx.x == 0) {       // where the host reads them
        P.counters[CX_CNT_TRIS] = (uint32_t)tbw; P.counters[CX_CNT_BORDER] = (uint32_t)(tbw >> 32);
    }
    if ((uint32_t)tbw > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap || P.counters[CX_CNT_CELLS] > P.ccap || nq > P.qcap) return;
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool emulate = (P.flags & CX_DIAG_CPYTHON310) != 0u;
    // One wave per round of the cells kernel (<= 64 records, their tetrahedra one contiguous range), grid-stride.  The rounds left
    // over after the last full pass of the grid's waves (config 4: 13.5 k rounds, 4096 waves: 1 240 of them) go as HALVES -- pentatope
    // groups 0-1 / 2-3 -- to twice as many waves: with whole rounds a third of the waves worked a fourth round-time while the rest
    // stood idle.  The second half counts the first half's tetrahedra (phase 1 only) to know where its own start.
    const uint32_t nrounds = (nq + 63u) >> 6;
    const uint32_t nwaves = gridDim.x * 4u, wid = blockIdx.x * 4u + wave;
    const uint32_t nfull = (nrounds / nwaves) * nwaves;
    const uint32_t nunits = nfull + 2u * (nrounds - nfull);
    for (uint32_t u = wid; u < nunits; u += nwaves) {
        const uint32_t r = (u < nfull) ? u : nfull + ((u - nfull) >> 1);
        const uint32_t part = (u < nfull) ? 0u : 1u + ((u - nfull) & 1u);      // 0: the whole round, 1: groups 0-1, 2: groups 2-3
        const uint4 rd = P.rounds[r];
        const uint32_t cbase = __builtin_amdgcn_readfirstlane(rd.x), ctot = __builtin_amdgcn_readfirstlane(rd.y);
        uint32_t tnext = __builtin_amdgcn_readfirstlane(rd.z);
        const bool have = lane < ctot;
        uint4 c4 = make_uint4(0, 0, 0, 0);
        if (have) c4 = P.cells[cbase + lane];
        const uint32_t lin = c4.x, sm = c4.y & 0xFFFFu;
        uint32_t q[4];
        cx_unravel4(P, lin, q);
        const bool real_voxel = have && q[0] + 1 < P.n0 && q[1] + 1 < P.n1 && q[2] + 1 < P.n2 && q[3] + 1 < P.n3;
        // pentatopes skipped by the reference's tolerances: recomputed exactly for the rare flagged records
        uint32_t pskip = 0;
        if (have && (c4.y & 0x10000u)) {
            float f[16];
            const uint32_t vm = cx_load_corners4(P, lin, q, f);
            const cx_cell4 R = cx_classify_cell4(P, f, vm, sm, q);
            pskip = R.pskip;
        }
        if (!real_voxel) pskip = 0xFFFFFFu;
        // first-vertex index and crossing mask of the 15 corners that can own an edge of this hyper-voxel
        // The queue entries of the (up to) 15 owner corners: first the item words of their bitmap words -- corners c and c | 1 share
        // one unless l + 1 starts the row's next word --, then the entries' words; each batch is requested whole before any of it
        // is used (written as load-use loops the compiler put an `s_waitcnt vmcnt(0)` behind every load: fifteen round trips per round)
        const uint32_t row0 = (q[0] * P.n1 + q[1]) * P.n2 + q[2];
        const uint32_t lw0 = q[3] >> 5, lb0 = q[3] & 31u;
        const bool cross = (lb0 == 31u);                     // l + 1 lies in the next bitmap word of the row
        uint32_t wantm = 0;                                  // bit c: corner c (0 = the cell itself .. 14) owns a crossing edge of this hyper-voxel
#pragma unroll
        for (uint32_t c = 0; c < 15; c++) {
            const uint32_t sc = ((sm >> c) & 1u) ? 0xFFFFu : 0u;
            uint32_t sup = 0;
#pragma unroll
            for (uint32_t c2 = c + 1; c2 < 16; c2++) sup |= ((c2 & c) == c) ? (1u << c2) : 0u;
            if (real_voxel && pskip != 0xFFFFFFu && ((sm ^ sc) & sup) != 0u) wantm |= 1u << c;
        }
        uint2 it0[8], it1[8];
#pragma unroll
        for (uint32_t g = 0; g < 8; g++) {                   // g = corner >> 1: the (i, j, k) offsets
            const uint32_t rowc = row0 + ((g & 4u) ? P.n1 * P.n2 : 0u) + ((g & 2u) ? P.n2 : 0u) + (g & 1u);
            const bool need = ((wantm >> (2u * g)) & 3u) != 0u;
            const uint32_t at = need ? rowc * P.nw3 + lw0 : 0u;
            it0[g] = P.items[at];
            it1[g] = P.items[at + ((need && cross) ? 1u : 0u)];
        }
#pragma unroll
        for (uint32_t g = 0; g < 8; g++) asm volatile("" : "+v"(it0[g].x), "+v"(it0[g].y), "+v"(it1[g].x), "+v"(it1[g].y) :: "memory");
        uint64_t ew[15];
#pragma unroll
        for (uint32_t c = 0; c < 15; c++) {
            const uint2 it = (c & 1u) ? it1[c >> 1] : it0[c >> 1];
            const uint32_t bit = (c & 1u) ? (cross ? 0u : lb0 + 1u) : lb0;
            ew[c] = 0;
            if ((wantm >> c) & 1u) ew[c] = P.info[it.x + __popc(it.y & ((1u << bit) - 1u))];
        }
#pragma unroll
        for (uint32_t c = 0; c < 15; c++) asm volatile("" : "+v"(ew[c]) :: "memory");
#pragma unroll
        for (uint32_t c = 0; c < 15; c++) {
            L.vf[wave][lane][c] = (uint32_t)ew[c];
            L.em[wave][lane][c] = (uint16_t)(ew[c] >> 32);
        }
        // probe codes of the corner hashes (absolute lattice coordinates) for the set-order emulation
        uint32_t S[16];
#pragma unroll
        for (uint32_t c = 0; c < 16; c++) S[c] = 0;
        if (emulate) {
#pragma unroll
            for (uint32_t c = 0; c < 16; c += 2) {
                const uint32_t ci = min(q[0] + ((c >> 3) & 1u), P.n0 - 1u), cj = min(q[1] + ((c >> 2) & 1u), P.n1 - 1u);
                const uint32_t ck = min(q[2] + ((c >> 1) & 1u), P.n2 - 1u);
                const uint64_t pre = P.hash_xyz[(ci * P.n1 + cj) * P.n2 + ck];
                S[c] = cx_probe_code(py_finish4(py_round4(pre, q[3] + P.org[3])));
                S[c + 1] = cx_probe_code(py_finish4(py_round4(pre, q[3] + 1u + P.org[3])));
            }
        }
        cx_tets_group<0>(P, L, wave, lane, S, q, sm, real_voxel, pskip, emulate, tnext, part != 2u);
        cx_tets_group<1>(P, L, wave, lane, S, q, sm, real_voxel, pskip, emulate, tnext, part != 2u);
        if (part != 1u) {      // wave-uniform
            cx_tets_group<2>(P, L, wave, lane, S, q, sm, real_voxel, pskip, emulate, tnext, true);
            cx_tets_group<3>(P, L, wave, lane, S, q, sm, real_voxel, pskip, emulate, tnext, true);
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------------------
const uint64_t* cx_pent_lut_device() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(cx_d_pent_local)) != hipSuccess) return nullptr;
    return (const uint64_t*)p;
}

void cx_launch_signbits4d(const cx_params4& P, hipStream_t s) {
    if (P.n3 % 32u == 0u && ((uintptr_t)P.grid & 15u) == 0u && !cx_debug_knob("CX4_SB_ROWS", 0)) {
        const uint32_t nquads = P.nsamples / 4u;
        uint32_t blocks = (nquads + 256u * CX4_SB_UNROLL - 1u) / (256u * CX4_SB_UNROLL);
        const uint32_t most = 256u * cx_debug_knob("CX4_SB_WGS", 16u);
        if (blocks > most) blocks = most;
        hipLaunchKernelGGL(cx_k_signbits4_flat, dim3(blocks), dim3(256), 0, s, P, nquads);
        return;
    }
    const uint32_t nchunk = (P.n3 + 63u) / 64u;
    const uint32_t total = P.nrows * nchunk;
    const uint32_t waves = (total + 7u) / 8u;
    uint32_t blocks = (waves + 3u) / 4u;
    if (blocks > 256u * 64u) blocks = 256u * 64u;   // grid-stride
    hipLaunchKernelGGL(cx_k_signbits4, dim3(blocks), dim3(256), 0, s, P, nchunk, cx_fdiv_make(nchunk), total);
}
void cx_launch_classify4d(const cx_params4& P, hipStream_t s) {
    const uint32_t nitems = P.nrows * P.nw3;
    uint32_t ipb = (nitems + 3071u) / 3072u;
    ipb = (ipb + 255u) & ~255u;
    if (ipb < 256u) ipb = 256u;
    const uint32_t blocks = (nitems + ipb - 1u) / ipb;
    hipLaunchKernelGGL(cx_k_queue4, dim3(blocks), dim3(256), 0, s, P, ipb);
    uint32_t cblocks = (P.qcap + CX4_CELLS_WAVES * 64u - 1u) / (CX4_CELLS_WAVES * 64u);
    if (cblocks > 1024u) cblocks = 1024u;         // grid-stride: four workgroups per CU
    hipLaunchKernelGGL(cx_k_cells4, dim3(cblocks ? cblocks : 1u), dim3(CX4_CELLS_WAVES * 64u), 0, s, P);
}
void cx_launch_emit_tets(const cx_params4& P, hipStream_t s) {
    uint32_t blocks = (P.qcap + 255u) / 256u;
    if (blocks > 256u * 4u) blocks = 256u * 4u;   // grid-stride: a few workgroups per CU
    hipLaunchKernelGGL(cx_k_emit_tets, dim3(blocks ? blocks : 1u), dim3(256), 0, s, P);
}
void cx_launch_hash_xyz(uint64_t* table, uint32_t n0, uint32_t n1, uint32_t n2, const uint32_t org[4], hipStream_t s) {
    const uint32_t n = n0 * n1 * n2;
    hipLaunchKernelGGL(cx_k_hash_xyz, dim3((n + 255u) / 256u), dim3(256), 0, s, table, n0, n1, n2, org[0], org[1], org[2]);
}
