// cx_march3d.hip -- Level-0 kernels of the 3-D marching-tetrahedra voxel march for gfx950 (wave64).
//
// Lattice "cell" q = lattice point (i,j,k) seen as the lower corner of the voxel [q, q+1].
// A cell OWNS the 7 lattice edges q -> q+d, d = 4di+2dj+dk in 1..7, and (when all 8 corners are
// inside the array) the 6 Kuhn tetrahedra of its voxel.  Vertex id of an edge = (lin(q) << 3) | d.
//
// K1  cx_k_classify<FAST>   one pass over the samples
//       phase A  every wave streams its share of the grid and pushes the ACTIVE cells (sign change
//                among the corners) into a wave-private LDS queue, in linear-index order.
//                FAST: lanes hold 4 consecutive k-samples (one 16-byte load per row); the sign
//                bits of a row are the v_cmp result masks (SGPRs), a cell's 8 corner signs are bit
//                shifts of 8 such masks, so inactive regions cost no VALU work and no LDS.
//                generic: one lane per cell, 8 scalar loads (any shape / alignment).
//       phase B  the wave re-reads the 8 corners of its queued cells (L1/L2 hits), classifies the
//                tetrahedra, counts vertices/triangles; the workgroup reserves output space with ONE
//                atomic per counter (same-address atomics saturate near 88/us on MI355X, so they
//                must stay in the low thousands per launch), then the wave interpolates and writes
//                vertex records, the per-cell lookup word and one record per active cell.
// K2  cx_k_emit_triangles   one lane per active-cell record: expands the tetrahedra into index
//                triples, looking vertex indices up in the per-cell table.
#include <cstdlib>

#include "cx_cell.h"

__device__ __constant__ uint8_t cx_d_tet_corners[6][4] = CX_TET_CORNERS_INIT;
__device__ __constant__ uint32_t cx_d_tet_tris[6][16][2] = CX_TET_TRIS_INIT;
__device__ __constant__ uint8_t cx_d_voxel_ntri[256] = CX_VOXEL_NTRI_INIT;

#ifndef CX_QCAP
#define CX_QCAP 1024u
#endif
#ifndef CX_K1_MIN_WAVES
#define CX_K1_MIN_WAVES 3   // two sample planes in flight per wave need ~130 VGPRs (4 waves/SIMD would spill)
#endif   // active cells a wave can queue before it has to flush on its own
#ifndef CX_RJ
#define CX_RJ 4         // cell rows per wave in the FAST kernel (a workgroup covers 4*CX_RJ rows)
#endif

struct cx_task {        // launch geometry of the FAST kernel
    uint32_t ci;        // cell planes per task
    uint32_t nks, njg, nic;
};

// ---- phase B --------------------------------------------------------------------------------------
struct cx_run {
    uint32_t v, t, c, b;   // running vertex / triangle / cell-record / border-voxel counts
};

// one sweep over the wave's queued cells.  emit == false: only count into `run` (from zero);
// emit == true: `run` holds the reserved bases and advances as records are written.
#define CX_VSTAGE 256u   // vertex records a wave stages in LDS per batch of 64 cells (typical: ~200)
__device__ __forceinline__ void cx_process_queue(const cx_params& P, const uint32_t* q, uint32_t n, uint32_t lane,
                                                 bool emit, cx_run& run, float4* vstage) {
    const uint32_t plane = P.n1 * P.n2;
    for (uint32_t b0 = 0; b0 < n; b0 += 64u) {
        const uint32_t idx = b0 + lane;
        const bool have = idx < n;
        const uint32_t lin = have ? q[idx] : 0u;
        const uint32_t i = cx_div(lin, P.div_plane);
        const uint32_t r = lin - i * plane;
        const uint32_t j = cx_div(r, P.div_row);
        const uint32_t k = r - j * P.n2;
        float f[8];
        const uint32_t vm = cx_load_corners(P, lin, i, j, k, f);
        const uint32_t sm = cx_sign_mask(P, f);
        const uint32_t smv = sm & vm;
        const bool active = have && smv != 0u && smv != vm;
        cx_cell_info R;
        R.sm = sm; R.emask = 0; R.ntri = 0; R.tetskip = 0; R.border = 0;
        if (active) R = cx_classify_cell(P, f, vm, sm, i, j, k);
        const uint32_t nv = __popc(R.emask);
        const bool rec = active && (nv != 0u || R.ntri != 0u);
        uint32_t vtot, ttot;
        const uint32_t vpre = cx_wave_prefix_small<3>(nv, vtot);
        const uint32_t tpre = cx_wave_prefix_small<4>(R.ntri, ttot);
        const uint64_t recm = __ballot(rec);
        const uint32_t ctot = (uint32_t)__popcll(recm);
        const uint32_t btot = (uint32_t)__popcll(__ballot(R.border != 0u));
        if (emit) {
            const uint32_t vfirst = run.v + vpre;
            if (run.v + vtot <= P.vcap) {
                if (vtot <= CX_VSTAGE) {
                    // the wave's vertices of this batch are one contiguous run of the vertex array:
                    // stage them in LDS, then write full 16-byte-per-lane rows
                    if (nv) cx_emit_vertices(P, f, R.emask, lin, i, j, k, [&](uint32_t r2, const float4& rec4) { vstage[vpre + r2] = rec4; });
                    if (!(P.flags & CX_DBG_NO_VERTS))
                        for (uint32_t o = lane; o < vtot; o += 64u) P.verts[run.v + o] = vstage[o];
                } else if (nv && !(P.flags & CX_DBG_NO_VERTS)) {
                    cx_emit_vertices(P, f, R.emask, lin, i, j, k, [&](uint32_t r2, const float4& rec4) { P.verts[vfirst + r2] = rec4; });
                }
                if (nv && !(P.flags & CX_DBG_NO_CELLTAB)) P.celltab[lin] = ((uint64_t)R.emask << 32) | (uint64_t)vfirst;
            }
            if (rec && run.c + ctot <= P.ccap && !(P.flags & CX_DBG_NO_CELLS)) {
                uint4 c4;
                c4.x = lin;
                c4.y = sm | (R.tetskip << 8) | (R.ntri << 16) | (R.emask << 24);
                c4.z = run.t + tpre;
                c4.w = vfirst;
                P.cells[run.c + cx_mbcnt(recm)] = c4;
            }
        }
        run.v += vtot; run.t += ttot; run.c += ctot; run.b += btot;
    }
}

// vertex / triangle / record / border counts of an active cell from its sign mask alone -- exact
// unless a corner is within the reference's np.allclose tolerances of the isovalue (then phase B
// recounts exactly).
struct cx_cnt {
    uint32_t v, t, c, b;
};
__device__ __forceinline__ void cx_count_from_signs(uint32_t sm, uint32_t vm, cx_cnt& acc) {
    const uint32_t s0 = (sm & 1u) ? 0xFFu : 0u;
    const uint32_t nv = __popc(((sm ^ s0) & vm) & 0xFEu);
    const bool real_voxel = (vm == 0xFFu);
    const uint32_t nt = real_voxel ? cx_voxel_ntri(sm) : 0u;
    acc.v += nv;
    acc.t += nt;
    acc.c += (nv | nt) ? 1u : 0u;
    acc.b += real_voxel ? 1u : 0u;
}
__device__ __forceinline__ uint32_t cx_wave_sum(uint32_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += (uint32_t)__shfl_xor((int)x, o);
    return x;
}

// FAST phase A bit layout: one u32 per lane and sample plane, 6 bits per sample row r (0..RJ):
//   bit 6r+m (m=0..3) : sample (row r, k = k0 + 4*lane + m) < isovalue
//   bit 6r+4          : the same for k+1 of m=3 (next lane's m=0, the halo sample, or a clamped repeat)
#define CX_ROWBITS 6u
constexpr uint32_t cx_rowmask(int rows, uint32_t bits) {
    uint32_t m = 0;
    for (int r = 0; r < rows; r++) m |= bits << (6 * r);
    return m;
}
#define CX_M0_MASK cx_rowmask(CX_RJ + 1, 1u)    // bit 6r+0, r = 0..RJ
#define CX_CELL_MASK cx_rowmask(CX_RJ, 0xFu)    // bits 6r+0..3, r = 0..RJ-1  (the 4*RJ cells of a lane)

// =================================================================================================
// K1
// =================================================================================================
template <bool FAST>
__global__ __launch_bounds__(256, CX_K1_MIN_WAVES) void cx_k_classify(const cx_params P, const cx_task T, const uint32_t cells_per_block) {
    __shared__ uint32_t s_queue[4][CX_QCAP];
    __shared__ float4 s_vstage[4][CX_VSTAGE];
    __shared__ uint32_t s_tot[4][4];
    __shared__ uint32_t s_base[4];
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* q = s_queue[wave];
    uint32_t qn = 0;   // wave-uniform
    unsigned long long* stamp = P.stamps ? P.stamps + ((size_t)blockIdx.x * 4u + wave) * 4u : nullptr;
    if (stamp && lane == 0) stamp[0] = __builtin_amdgcn_s_memtime();
    cx_cnt acc = {0, 0, 0, 0};   // per-lane counts of the queued cells (from sign masks)
    float dnear = 3.0e38f;       // per-lane: smallest |f - vcmp| among the samples seen (tolerance screen)
    uint32_t wcur = 0, act0 = 0;
    bool pending = false;        // wave-uniform: a classified step is waiting for queue space
    const float* __restrict__ A = P.grid;
    const uint32_t plane = P.n1 * P.n2;

    // ---- FAST task: block -> (k segment of 256 samples, group of 16 rows, chunk of ci planes)
    uint32_t k0 = 0, j0 = 0, ib = 0, nrows = 0, kofs = 0, last_lane = 0, p = 0, wprev = 0, mk = 0, mj = 0, mr = 0, kofs_c = 0;
    bool lane_valid = false, halo_in = false;
    // ---- generic task: contiguous range of linear cell indices per block
    uint32_t gbase = 0, gend = 0;
    bool streaming;
    if (FAST) {
        uint32_t b = blockIdx.x;
        const uint32_t ks = b % T.nks; b /= T.nks;
        const uint32_t jg = b % T.njg;
        const uint32_t ic = b / T.njg;
        k0 = ks * 256u;
        j0 = jg * (4u * CX_RJ) + wave * CX_RJ;
        p = ic * T.ci;
        ib = min(p + T.ci, P.n0);
        nrows = (j0 < P.n1) ? min((uint32_t)CX_RJ, P.n1 - j0) : 0u;
        kofs = k0 + 4u * lane;
        lane_valid = kofs < P.n2;                      // n2 % 4 == 0
        kofs_c = lane_valid ? kofs : (P.n2 - 4u);
        last_lane = (uint32_t)__popcll(__ballot(lane_valid)) - 1u;
        halo_in = (k0 + 256u) < P.n2;                  // a sample right of this segment exists
        streaming = nrows != 0u && p < ib;
        // cells whose k+1 / j+1 neighbour exists (bit layout of CX_CELL_MASK)
        mr = (nrows >= CX_RJ) ? CX_CELL_MASK : (CX_CELL_MASK & ((1u << (CX_ROWBITS * nrows)) - 1u));   // rows that exist
        if (!lane_valid) mr = 0;   // lanes right of the array hold re-read samples: they own no cells
        mk = mr;
        if (lane == last_lane && !halo_in) mk &= ~(CX_M0_MASK << 3);       // m == 3 at the array edge
        mj = 0;
        for (uint32_t r = 0; r < CX_RJ; r++)
            if (j0 + r + 1u < P.n1) mj |= 0xFu << (CX_ROWBITS * r);
        mj &= mr;
    } else {
        gbase = blockIdx.x * cells_per_block + wave * 64u;
        gend = min(blockIdx.x * cells_per_block + cells_per_block, P.nsamples);
        streaming = gbase < gend;
    }

    // one sample plane of this lane: RJ+1 rows x 4 consecutive k-samples, plus the sample right of the segment
    struct plane_raw {
        float4 v[CX_RJ + 1];
        float hv[CX_RJ + 1];
    };
    auto load_plane = [&](uint32_t pp, plane_raw& R) {
        const uint32_t pc = min(pp, P.n0 - 1u);
#pragma unroll
        for (int r = 0; r <= CX_RJ; r++) {
            const uint32_t jr = min(j0 + (uint32_t)r, P.n1 - 1u);   // rows beyond the array repeat the last row
            const uint32_t rowofs = (pc * P.n1 + jr) * P.n2;
            R.v[r] = *reinterpret_cast<const float4*>(A + rowofs + kofs_c);      // lanes right of the array re-read its last 4 samples
            R.hv[r] = A[rowofs + (halo_in ? k0 + 256u : 0u)];                    // wave-uniform address
        }
    };
    // sign bits of a loaded plane.  f < vcmp  <=>  sign bit of (f - vcmp)  (f == vcmp gives +0; NaN samples
    // are not supported); the same differences feed the tolerance screen (smallest |f - vcmp| seen)
    auto plane_bits = [&](const plane_raw& R) -> uint32_t {
        uint32_t own = 0, halo = 0;
#pragma unroll
        for (int r = 0; r <= CX_RJ; r++) {
            const float dx = R.v[r].x - P.vcmp, dy = R.v[r].y - P.vcmp, dz = R.v[r].z - P.vcmp, dw = R.v[r].w - P.vcmp;
            const float dh = R.hv[r] - P.vcmp;
            own |= (__float_as_uint(dx) >> 31) << (CX_ROWBITS * r + 0);
            own |= (__float_as_uint(dy) >> 31) << (CX_ROWBITS * r + 1);
            own |= (__float_as_uint(dz) >> 31) << (CX_ROWBITS * r + 2);
            own |= (__float_as_uint(dw) >> 31) << (CX_ROWBITS * r + 3);
            halo |= (__float_as_uint(dh) >> 31) << (CX_ROWBITS * r);
            dnear = fminf(dnear, fminf(fminf(fabsf(dx), fabsf(dy)), fminf(fabsf(dz), fminf(fabsf(dw), fabsf(dh)))));
        }
        // k+1 neighbour of m=3: m=0 of the next lane; the last valid lane takes the halo sample or,
        // at the array edge, repeats its own m=3 (clamped corner)
        uint32_t nb = (uint32_t)__shfl_down((int)own, 1) & CX_M0_MASK;
        if (lane == last_lane) nb = halo_in ? halo : ((own >> 3) & CX_M0_MASK);
        return own | (nb << 4);
    };

    plane_raw rawA, rawB;
    bool odd = false;            // wave-uniform: which buffer holds plane p+1
    if (FAST && streaming) {
        load_plane(p, rawB);
        load_plane(p + 1u, rawA);
        wprev = plane_bits(rawB);
    }


    for (;;) {
        // ---- phase A: stream until done or until the queue might not hold another step
        if (FAST) {
            // one step: plane p+1 is in `cur` (loaded one step ago), plane p+2 is requested into `nxt`
            auto step = [&](const plane_raw& cur, plane_raw& nxt) -> bool {
                load_plane(p + 2u, nxt);
                wcur = plane_bits(cur);
                // per sample row: OR / AND over (k, k+1); then over rows (r, r+1); then over both planes
                const uint32_t op = wprev | (wprev >> 1), ap = wprev & (wprev >> 1);
                const uint32_t oc = wcur | (wcur >> 1), ac = wcur & (wcur >> 1);
                const uint32_t o = op | (op >> CX_ROWBITS) | oc | (oc >> CX_ROWBITS);
                const uint32_t a = ap & (ap >> CX_ROWBITS) & ac & (ac >> CX_ROWBITS);
                act0 = o & ~a & mr;
                if (!lane_valid) act0 = 0;
                pending = false;
                if (__ballot(act0 != 0u) != 0ULL) {    // wave-uniform: some cell of this step has a sign change
                    uint32_t tot;
                    const uint32_t pre = cx_wave_prefix_small<5>(__popc(act0), tot);
                    if (qn + tot > CX_QCAP) {          // wave-uniform: no room -> emit what is queued, then resume here
                        pending = true;
                        return false;
                    }
                    uint32_t act = act0;
                    uint32_t pos = qn + pre;
                    const uint32_t lin0 = (p * P.n1 + j0) * P.n2 + kofs;
                    while (act) {
                        const uint32_t bit = __ffs(act) - 1u;
                        act &= act - 1u;
                        const uint32_t r = bit / CX_ROWBITS, m = bit - r * CX_ROWBITS;
                        q[pos++] = lin0 + r * P.n2 + m;
                    }
                    // counts of this lane's 16 cells, all at once on the packed sign words: corner
                    // c = (di,dj,dk) of the cell at bit b is bit b of (plane di word) >> (6*dj + dk).
                    // Cells without a sign change contribute nothing, so no masking by `act0` is needed.
                    const uint32_t mi = ((p + 1u) < P.n0) ? mr : 0u;               // plane p+1 exists
                    const uint32_t real = mk & mj & mi;                           // cells that are voxels
                    const uint32_t b0 = wprev, b1 = wprev >> 1, b2 = wprev >> CX_ROWBITS, b3 = wprev >> (CX_ROWBITS + 1u);
                    const uint32_t b4 = wcur, b5 = wcur >> 1, b6 = wcur >> CX_ROWBITS, b7 = wcur >> (CX_ROWBITS + 1u);
                    const uint32_t x1 = (b0 ^ b1) & mk, x2 = (b0 ^ b2) & mj, x3 = (b0 ^ b3) & mk & mj;   // mk, mj subsets of mr
                    const uint32_t x4 = (b0 ^ b4) & mi, x5 = (b0 ^ b5) & mk & mi, x6 = (b0 ^ b6) & mj & mi, x7 = (b0 ^ b7) & real;
                    acc.v += __popc(x1) + __popc(x2) + __popc(x3) + __popc(x4) + __popc(x5) + __popc(x6) + __popc(x7);
                    const uint32_t owners = x1 | x2 | x3 | x4 | x5 | x6 | x7;
                    acc.c += __popc(owners | (act0 & real));
                    acc.b += __popc(act0 & real);
                    // triangles: per tetrahedron {0,7,c,d} the number of low corners n = b0+b7+bc+bd;
                    // n odd -> 1 triangle, n == 2 -> 2 triangles  (only voxels emit)
                    const uint32_t x07 = b0 ^ b7, y07 = b0 & b7;
                    uint32_t nt = 0;
#define CX_TET_COUNT(bc, bd)                                                          \
    {                                                                                 \
        const uint32_t xcd = (bc) ^ (bd);                                             \
        const uint32_t s0 = x07 ^ xcd;                                                \
        const uint32_t s1 = y07 ^ ((bc) & (bd)) ^ (x07 & xcd);                         \
        nt += __popc(s0 & real) + 2u * __popc(s1 & ~s0 & real);                        \
    }
                    CX_TET_COUNT(b1, b3) CX_TET_COUNT(b3, b2) CX_TET_COUNT(b2, b6)
                    CX_TET_COUNT(b6, b4) CX_TET_COUNT(b4, b5) CX_TET_COUNT(b5, b1)
#undef CX_TET_COUNT
                    acc.t += nt;
                    qn += tot;
                }
                wprev = wcur;
                p++;
                streaming = p < ib;
                return true;
            };
            while (streaming) {
                const bool done = odd ? step(rawB, rawA) : step(rawA, rawB);
                if (!done) break;
                odd = !odd;
            }
        } else {
            while (streaming && qn + 64u <= CX_QCAP) {
                const uint32_t lin = gbase + lane;
                const bool in = lin < gend;
                const uint32_t linc = in ? lin : (P.nsamples - 1u);
                const uint32_t i = cx_div(linc, P.div_plane);
                const uint32_t r = linc - i * plane;
                const uint32_t j = cx_div(r, P.div_row);
                const uint32_t k = r - j * P.n2;
                float f[8];
                const uint32_t vm = cx_load_corners(P, linc, i, j, k, f);
                const uint32_t sm = cx_sign_mask(P, f);
                const uint32_t smv = sm & vm;
                const bool active = in && smv != 0u && smv != vm;
                const uint64_t act = __ballot(active);
                if (active) {
                    q[qn + cx_mbcnt(act)] = lin;
                    cx_count_from_signs(sm, vm, acc);
#pragma unroll
                    for (int c = 0; c < 8; c++) dnear = fminf(dnear, fabsf(f[c] - P.vcmp));
                }
                qn += (uint32_t)__popcll(act);
                gbase += 256u;
                streaming = gbase < gend;
            }
        }
        // ---- phase B: count, reserve, emit.  The last round of a workgroup reserves once for all
        // four waves; a wave whose queue filled up early reserves for itself (dense surfaces only).
        const bool final_round = !streaming && !pending;
        if (stamp && lane == 0 && final_round) stamp[1] = __builtin_amdgcn_s_memtime();
        if (P.flags & CX_DBG_PHASE_A_ONLY) {
            if (final_round) break;
            qn = 0;
            continue;
        }
        cx_run run = {0, 0, 0, 0};
        const bool recount = __ballot(dnear <= P.near_abs) != 0ULL;   // wave-uniform
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1 || recount) {
                cx_process_queue(P, q, qn, lane, pass == 1, run, s_vstage[wave]);
            } else {
                run.v = cx_wave_sum(acc.v); run.t = cx_wave_sum(acc.t);
                run.c = cx_wave_sum(acc.c); run.b = cx_wave_sum(acc.b);
            }
            if (pass == 1 || (P.flags & CX_DBG_COUNT_ONLY)) break;
            if (final_round) {
                if (lane == 0) {
                    s_tot[wave][0] = run.v; s_tot[wave][1] = run.t; s_tot[wave][2] = run.c; s_tot[wave][3] = run.b;
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    const uint32_t v = s_tot[0][0] + s_tot[1][0] + s_tot[2][0] + s_tot[3][0];
                    const uint32_t t = s_tot[0][1] + s_tot[1][1] + s_tot[2][1] + s_tot[3][1];
                    const uint32_t c = s_tot[0][2] + s_tot[1][2] + s_tot[2][2] + s_tot[3][2];
                    const uint32_t bb = s_tot[0][3] + s_tot[1][3] + s_tot[2][3] + s_tot[3][3];
                    s_base[0] = v ? atomicAdd(&P.counters[CX_CNT_VERTS], v) : 0u;
                    s_base[1] = t ? atomicAdd(&P.counters[CX_CNT_TRIS], t) : 0u;
                    s_base[2] = c ? atomicAdd(&P.counters[CX_CNT_CELLS], c) : 0u;
                    if (bb) atomicAdd(&P.counters[CX_CNT_BORDER], bb);
                }
                __syncthreads();
                run.v = s_base[0]; run.t = s_base[1]; run.c = s_base[2];
                for (uint32_t w = 0; w < wave; w++) {
                    run.v += s_tot[w][0]; run.t += s_tot[w][1]; run.c += s_tot[w][2];
                }
                if (stamp && lane == 0) stamp[2] = __builtin_amdgcn_s_memtime();
            } else {
                cx_run base = {0, 0, 0, 0};
                if (lane == 0) {
                    if (run.v) base.v = atomicAdd(&P.counters[CX_CNT_VERTS], run.v);
                    if (run.t) base.t = atomicAdd(&P.counters[CX_CNT_TRIS], run.t);
                    if (run.c) base.c = atomicAdd(&P.counters[CX_CNT_CELLS], run.c);
                    if (run.b) atomicAdd(&P.counters[CX_CNT_BORDER], run.b);
                }
                run.v = __builtin_amdgcn_readfirstlane(base.v);
                run.t = __builtin_amdgcn_readfirstlane(base.t);
                run.c = __builtin_amdgcn_readfirstlane(base.c);
            }
        }
        if (stamp && lane == 0 && final_round) stamp[3] = __builtin_amdgcn_s_memtime();
        qn = 0;
        acc.v = acc.t = acc.c = acc.b = 0;
        dnear = 3.0e38f;
        if (final_round) break;
        if (FAST) {
            // a wave that had to emit in the middle of its task dropped its prefetched planes (so that
            // they do not occupy registers during phase B): fetch plane p+1 again and redo the step
            load_plane(p + 1u, rawA);
            odd = false;
            pending = false;
        }
    }
}

// ---- CPython 3.10 tuple hash + 8-slot set order, used only with
// CX_DIAG_CPYTHON310 to reproduce the quad diagonal the reference picks (tetrahedral.py:592-595).
// hash((x,y,z)) = finish(round(round(round(P5, x), y), z)); the first two rounds depend only on
// (x,y) and come from a table built once per grid shape (cx_k_hash_xy).
#define CX_PY_P1 11400714785074694791ULL
#define CX_PY_P2 14029467366897019727ULL
#define CX_PY_P5 2870177450012600261ULL
__device__ __forceinline__ uint64_t py_round(uint64_t acc, uint32_t x) {
    acc += (uint64_t)x * CX_PY_P2;
    acc = (acc << 31) | (acc >> 33);
    return acc * CX_PY_P1;
}
__device__ __forceinline__ uint64_t py_finish3(uint64_t acc) {
    acc += 3ULL ^ (CX_PY_P5 ^ 3527539ULL);
    return (acc == ~0ULL) ? 1546275796ULL : acc;
}
__global__ void cx_k_hash_xy(uint64_t* table, uint32_t n0, uint32_t n1, uint32_t org0, uint32_t org1) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n0 * n1) return;
    const uint32_t i = idx / n1, j = idx - i * n1;
    table[idx] = py_round(py_round(CX_PY_P5, i + org0), j + org1);
}
// does a 2-element set {first inserted h1, then h2} iterate h2 first?
__device__ __forceinline__ bool py_set2_swapped(uint64_t h1, uint64_t h2) {
    const uint32_t s1 = (uint32_t)h1 & 7u;
    uint32_t s2 = (uint32_t)h2 & 7u;
    uint64_t perturb = h2;
    for (int guard = 0; guard < 16 && s2 == s1; guard++) {
        perturb >>= 5;
        s2 = (uint32_t)((s2 * 5u + 1u + perturb) & 7u);
    }
    return s2 < s1;
}

// tet vertex m of tet t -> cube corner, as compile-time constants (same data as cx_d_tet_corners)
__device__ constexpr uint8_t CX_TC[6][4] = CX_TET_CORNERS_INIT;
#define CX_TC_MASK(t) ((1u << CX_TC[t][0]) | (1u << CX_TC[t][1]) | (1u << CX_TC[t][2]) | (1u << CX_TC[t][3]))

// =================================================================================================
// K2: one lane per active-cell record; expands the 6 tetrahedra into index triples.  A wave whose
// 64 records own one contiguous triangle range stages its indices in LDS and writes them out as
// full 256-byte rows; otherwise (range broken by a reservation boundary) lanes store directly.
// =================================================================================================
#define CX_K2_STAGE 832u    // ints per wave and round (typical: ~600); larger waves store directly
__device__ constexpr uint8_t CX_EDGE[19][2] = CX_EDGES_INIT;
__global__ __launch_bounds__(256) void cx_k_emit_triangles(const cx_params P, const uint64_t* __restrict__ hash_xy) {
    __shared__ int32_t s_stage[4][CX_K2_STAGE];
    __shared__ uint32_t s_lut[6 * 16 * 2];   // triangle LUT: per-lane lookups must not go to memory
    __shared__ int32_t s_eidx[4][19][64];   // vertex index of each of the 19 voxel edges, per lane
    const uint32_t ncells = min(P.counters[CX_CNT_CELLS], P.ccap);
    if (P.counters[CX_CNT_TRIS] > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap) return;  // host re-runs with more room
    if (blockIdx.x * blockDim.x >= ncells) return;                 // whole block idle
    if (threadIdx.x < 6 * 16 * 2) s_lut[threadIdx.x] = (&cx_d_tet_tris[0][0][0])[threadIdx.x];
    __syncthreads();
    const uint32_t plane = P.n1 * P.n2;
    const bool emulate = (P.flags & CX_DIAG_CPYTHON310) != 0u;
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x + wave * 64u >= ncells) return;   // whole wave idle
    const bool have = idx < ncells;
    uint4 c4 = make_uint4(0, 0, 0, 0);
    if (have) c4 = P.cells[idx];
    const uint32_t ntri = have ? ((c4.y >> 16) & 0xFFu) : 0u;
    const uint32_t lin = c4.x;
    const uint32_t sm = c4.y & 0xFFu, tetskip = (c4.y >> 8) & 0x3Fu;
    // first-vertex index and crossing mask of the 7 corners that can own an edge of this voxel
    uint32_t vfirst[7], em[7];
    vfirst[0] = c4.w;
    em[0] = c4.y >> 24;
#pragma unroll
    for (uint32_t c = 1; c < 7; c++) {
        // does corner c own a crossing edge of this voxel?  (a strict superset corner on the other side)
        const uint32_t sc = ((sm >> c) & 1u) ? 0xFFu : 0u;
        uint32_t sup = 0;
#pragma unroll
        for (uint32_t c2 = 0; c2 < 8; c2++) sup |= ((c2 & c) == c && c2 != c) ? (1u << c2) : 0u;
        vfirst[c] = 0; em[c] = 0;
        if (ntri && ((sm ^ sc) & sup) != 0u && !(P.flags & CX_DBG_NO_LOOKUP)) {
            const uint32_t lc = lin + ((c & 4u) ? plane : 0u) + ((c & 2u) ? P.n2 : 0u) + (c & 1u);
            const uint64_t e = P.celltab[lc];
            vfirst[c] = (uint32_t)e;
            em[c] = (uint32_t)(e >> 32);
        }
    }
    // quad diagonal variants of the 2-2 tetrahedra (bit t of `variants`)
    uint32_t variants = 0;
    if (emulate) {
        uint32_t need = 0;   // corners whose hash is needed
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const uint32_t pat = ((sm >> CX_TC[t][0]) & 1u) | (((sm >> CX_TC[t][1]) & 1u) << 1) |
                                 (((sm >> CX_TC[t][2]) & 1u) << 2) | (((sm >> CX_TC[t][3]) & 1u) << 3);
            if (ntri && !((tetskip >> t) & 1u) && __popc(pat) == 2) need |= CX_TC_MASK(t);
        }
        if (__ballot(need != 0u) != 0ULL) {
            const uint32_t ci = cx_div(lin, P.div_plane);
            const uint32_t r = lin - ci * plane;
            const uint32_t cj = cx_div(r, P.div_row);
            const uint32_t ck = r - cj * P.n2;
            // hash prefixes of the 4 (i,j) columns of this voxel (clamped: pseudo cells never need them)
            const uint32_t i1 = min(ci + 1u, P.n0 - 1u), j1 = min(cj + 1u, P.n1 - 1u);
            const uint64_t hxy[4] = {hash_xy[ci * P.n1 + cj], hash_xy[ci * P.n1 + j1], hash_xy[i1 * P.n1 + cj], hash_xy[i1 * P.n1 + j1]};
            uint64_t h[8];
#pragma unroll
            for (uint32_t c = 0; c < 8; c++) {
                h[c] = 0;
                if ((need >> c) & 1u) h[c] = py_finish3(py_round(hxy[c >> 1], ck + (c & 1u) + P.org2));
            }
#pragma unroll
            for (int t = 0; t < 6; t++) {
                const uint32_t pat = ((sm >> CX_TC[t][0]) & 1u) | (((sm >> CX_TC[t][1]) & 1u) << 1) |
                                     (((sm >> CX_TC[t][2]) & 1u) << 2) | (((sm >> CX_TC[t][3]) & 1u) << 3);
                if (__popc(pat) != 2) continue;
                // low set and high set, each in insertion (tet vertex) order
                uint64_t hl0 = 0, hl1 = 0, hh0 = 0, hh1 = 0;
                int nl = 0, nh = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const uint64_t hm = h[CX_TC[t][m]];
                    if ((pat >> m) & 1u) { if (nl == 0) hl0 = hm; else hl1 = hm; nl++; }
                    else { if (nh == 0) hh0 = hm; else hh1 = hm; nh++; }
                }
                if (py_set2_swapped(hl0, hl1) != py_set2_swapped(hh0, hh1)) variants |= 1u << t;
            }
        }
    }
    // The wave's triangles are written in two rounds (tetrahedra 0-2, then 3-5) so that the LDS stage only has to
    // hold half of them: round g goes to [tb0 + (g ? T0 : 0) + prefix_g(lane), ...).  The order of triangles inside
    // the wave's range is free (c4.z only serves this kernel).  A wave whose records do not own one contiguous
    // range (reservation boundary inside the wave) or that is too large for the stage stores directly, per cell.
    uint32_t nround[2] = {0u, 0u};
#pragma unroll
    for (int t = 0; t < 6; t++) {
        const uint32_t pat = ((sm >> CX_TC[t][0]) & 1u) | (((sm >> CX_TC[t][1]) & 1u) << 1) |
                             (((sm >> CX_TC[t][2]) & 1u) << 2) | (((sm >> CX_TC[t][3]) & 1u) << 3);
        const uint32_t np = __popc(pat);
        const uint32_t nt = (ntri == 0u || ((tetskip >> t) & 1u)) ? 0u : ((np == 2u) ? 2u : (np & 1u));
        nround[t / 3] += nt;
    }
    uint32_t ttot, T0, T1;
    const uint32_t tpre = cx_wave_prefix_small<4>(ntri, ttot);
    const uint32_t pre0 = cx_wave_prefix_small<3>(nround[0], T0);
    const uint32_t pre1 = cx_wave_prefix_small<3>(nround[1], T1);
    const uint32_t tb0 = __builtin_amdgcn_readfirstlane(c4.z);   // lane 0 always has a record here
    const bool contiguous = (T0 * 3u <= CX_K2_STAGE) && (T1 * 3u <= CX_K2_STAGE) &&
                            __ballot(ntri != 0u && c4.z != tb0 + tpre) == 0ULL;
    int32_t* stage = s_stage[wave];
    int32_t* direct = P.tris + (size_t)c4.z * 3u;
    // vertex index of every voxel edge (owner corner c1, direction d): first vertex of the owner +
    // rank of d among the owner's crossing edges.  Static register indices; the table lives in LDS so
    // that the runtime edge ids of the triangle LUT become one ds_read each.
#pragma unroll
    for (int e = 0; e < 19; e++) {
        const uint32_t c1 = CX_EDGE[e][0], d = CX_EDGE[e][1];
        s_eidx[wave][e][lane] = (int32_t)(vfirst[c1] + __popc(em[c1] & ((1u << d) - 1u)));
    }
    const int32_t* eidx = &s_eidx[wave][0][lane];
    uint32_t wd = 0;   // running position of the direct path (per cell)
#pragma unroll
    for (int g = 0; g < 2; g++) {
        uint32_t w = (g ? pre1 : pre0) * 3u;
#pragma unroll
        for (int tt = 0; tt < 3; tt++) {
            const int t = g * 3 + tt;
            if (ntri == 0u || ((tetskip >> t) & 1u)) continue;
            const uint32_t pat = ((sm >> CX_TC[t][0]) & 1u) | (((sm >> CX_TC[t][1]) & 1u) << 1) |
                                 (((sm >> CX_TC[t][2]) & 1u) << 2) | (((sm >> CX_TC[t][3]) & 1u) << 3);
            const uint32_t e = s_lut[(t * 16 + pat) * 2 + ((variants >> t) & 1u)];
            const uint32_t n = e >> 30;
            for (uint32_t qd = 0; qd < n; qd++) {
                const uint32_t tri = (e >> (15u * qd)) & 0x7FFFu;
#pragma unroll
                for (uint32_t sidx = 0; sidx < 3; sidx++) {
                    const uint32_t eid = (tri >> (5u * sidx)) & 0x1Fu;
                    const int32_t vi = eidx[eid * 64u];
                    if (contiguous) stage[w++] = vi;
                    else if (!(P.flags & CX_DBG_NO_TRIS)) direct[wd++] = vi;
                }
            }
        }
        if (contiguous && !(P.flags & CX_DBG_NO_TRIS)) {
            int32_t* out = P.tris + ((size_t)tb0 + (g ? T0 : 0u)) * 3u;
            const uint32_t total = (g ? T1 : T0) * 3u;
            for (uint32_t o = lane; o < total; o += 64u) out[o] = stage[o];
        }
    }
}

// ---- launchers --------------------------------------------------------------------------------------
bool cx_fast_classify_supported(const cx_params& P) {
    return (P.n2 % 4u == 0u) && ((reinterpret_cast<uintptr_t>(P.grid) & 15u) == 0u);
}

void cx_launch_classify_fast(const cx_params& P, hipStream_t s) {
    cx_task T;
    T.nks = (P.n2 + 255u) / 256u;
    T.njg = (P.n1 + 4u * CX_RJ - 1u) / (4u * CX_RJ);
    // planes per task: aim at >= ~2048 workgroups, between 4 and 32 planes each
    const uint32_t per_plane = T.nks * T.njg;
    uint32_t target = 3072u;   // 12 workgroups per CU: measured best at 512^3 (tools/quick_time.py sweep)
    if (const char* e = getenv("CX_TASKS")) target = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : target;   // tuning knob
    uint32_t want_chunks = (target + per_plane - 1u) / per_plane;
    if (want_chunks < 1u) want_chunks = 1u;
    uint32_t ci = (P.n0 + want_chunks - 1u) / want_chunks;
    if (ci < 2u) ci = 2u;
    if (ci > 64u) ci = 64u;
    T.ci = ci;
    T.nic = (P.n0 + ci - 1u) / ci;
    const uint32_t blocks = T.nks * T.njg * T.nic;
    hipLaunchKernelGGL(cx_k_classify<true>, dim3(blocks), dim3(256), 0, s, P, T, 0u);
}

void cx_launch_classify_generic(const cx_params& P, hipStream_t s) {
    cx_task T = {0, 0, 0, 0};
    // contiguous cell ranges per block: multiples of 256, at least 16384, about 2048 blocks
    uint32_t cpb = (P.nsamples + 2047u) / 2048u;
    cpb = (cpb + 255u) & ~255u;
    if (cpb < 16384u) cpb = 16384u;
    const uint32_t blocks = (P.nsamples + cpb - 1u) / cpb;
    hipLaunchKernelGGL(cx_k_classify<false>, dim3(blocks), dim3(256), 0, s, P, T, cpb);
}

void cx_launch_emit_triangles(const cx_params& P, const uint64_t* hash_xy, hipStream_t s) {
    // one lane per record; the record count lives on the device, so launch for the capacity and let
    // idle waves exit at once
    const uint32_t blocks = (P.ccap + 255u) / 256u;
    hipLaunchKernelGGL(cx_k_emit_triangles, dim3(blocks ? blocks : 1u), dim3(256), 0, s, P, hash_xy);
}

void cx_launch_hash_xy(uint64_t* table, uint32_t n0, uint32_t n1, uint32_t org0, uint32_t org1, hipStream_t s) {
    const uint32_t n = n0 * n1;
    hipLaunchKernelGGL(cx_k_hash_xy, dim3((n + 255u) / 256u), dim3(256), 0, s, table, n0, n1, org0, org1);
}
