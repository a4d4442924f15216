// cx_march3d.hip -- Level-0 kernels of the 3-D marching-tetrahedra voxel march for gfx950 (wave64).
//
// Lattice "cell" q = lattice point (i,j,k) seen as the lower corner of the voxel [q, q+1].
// A cell OWNS the 7 lattice edges q -> q+d, d = 4di+2dj+dk in 1..7, and (when all 8 corners are
// inside the array) the 6 Kuhn tetrahedra of its voxel.  Vertex id of an edge = (lin(q) << 3) | d.
//
// Staged pipeline (any grid with n2 >= 4) -- no atomics, no barriers:
//   S1  cx_k_stream            one pass over the samples: sign bits packed per lane, active cells (sign
//                              change among the 8 corners) appended to a per-wave queue in global memory
//                              as packed entries; the queue is cut into batches of >= CX_BATCH_MIN cells
//                              with the vertex / triangle counts that precede them (from the packed words).
//   S2  cx_k_scan_waves        exclusive scan of the per-wave totals -> output offsets, counters, and a
//                              flat list of all batches.
//   S3  cx_k_emit_vertices     one wave per batch: vertex records (one lane per vertex), per-cell table,
//                              cell records.
//   K2  cx_k_emit_triangles    one lane per cell record: expands the tetrahedra into index triples, looking
//                              vertex indices up in the per-cell table.
// Rows shorter than 4 samples, or on request (CX_KERNEL_GENERIC): cx_k_classify_generic (one lane per cell, LDS queue, one reservation atomic per
// workgroup and counter -- same-address atomics saturate near 88/us on MI355X) followed by K2.
#include <cstdlib>

#include "cx_cell.h"

__device__ __constant__ uint8_t cx_d_tet_corners[6][4] = CX_TET_CORNERS_INIT;
__device__ __constant__ uint32_t cx_d_tet_tris[6][16][2] = CX_TET_TRIS_INIT;
__device__ __constant__ uint8_t cx_d_voxel_ntri[256] = CX_VOXEL_NTRI_INIT;

#ifndef CX_QCAP
#define CX_QCAP 1024u       // generic kernel: active cells a wave can queue in LDS before it flushes on its own
#endif
#ifndef CX_K1_MIN_WAVES
#define CX_K1_MIN_WAVES 3   // stream kernel: two sample planes in flight per wave
#endif
#ifndef CX_S1_DEPTH
#define CX_S1_DEPTH 1        // sample planes in flight ahead of the one the stream kernel works on (1 or 2; 2 measured slower: 0.131 vs 0.125 ms, 168 registers)
#endif
#ifndef CX_RJ
#define CX_RJ 4             // cell rows per wave in the stream kernel (a workgroup covers 4*CX_RJ rows)
#endif
#ifndef CX_S1_LAZY
#define CX_S1_LAZY 1         // stream kernel: queued cells are counted in full rounds of 64 when the stage is flushed or a batch closes (0: one round per plane step)
#endif
#ifndef CX_S3_MIN_SHARE
#define CX_S3_MIN_SHARE 4u   // rounds a vertex-stage wave takes at least (small surfaces: fewer waves rather than waves that only start up)
#endif
#ifndef CX_BATCH_MIN
#define CX_BATCH_MIN 512u   // a streaming wave closes a batch once it holds this many cells (128..1024 measured: 512 best)
#endif

// the word the vertex stage leaves per queue entry (staged pipeline): bits 0-31 first vertex of the cell, 32-39 crossing mask
// (bit d: the cell owns a crossing on its edge in direction d), 40-43 triangles of its voxel, 44-49 tetrahedra that emit nothing
// (the reference's tolerance skips; all six for a cell that is no voxel).  Everything the triangle stage needs about a cell besides
// its queue entry: it walks queue entries, not cell records (cx_k_emit_triangles_e).
__device__ __forceinline__ uint64_t cx_info_word(uint32_t vfirst, uint32_t emask, uint32_t ntri, uint32_t tetskip) {
    return (uint64_t)vfirst | ((uint64_t)((emask & 0xFFu) | ((ntri & 0xFu) << 8) | ((tetskip & 0x3Fu) << 12)) << 32);
}

// ---- per-cell path ------------------------------------------------------------------------------
struct cx_run {
    uint32_t v, t, c, b;   // running vertex / triangle / cell-record / border-voxel counts
    uint32_t s;            // count pass: owned crossings whose two samples differ by <= 1e-8 (the reference's ratio = 0.5 rule)
};

// one sweep over queued cells, re-reading their corners and applying the reference's tolerance rules
// exactly.  emit == false: only count into `run` (from zero); emit == true: `run` holds the bases and
// advances as vertex records and table entries (RECORDS: and cell records) are written.
#define CX_VSTAGE 256u   // vertex records a wave stages in LDS per batch of 64 cells (typical: ~200)
#define CX_VSTAGE_LDS 384u   // float4 slots per wave in the vertex stage's LDS (fast path: 2 x 448 slot words + 64 x 9 corner samples)
template <bool RECORDS, typename LinOf>
__device__ __forceinline__ void cx_process_queue(const cx_params& P, LinOf lin_of, uint32_t n, uint32_t lane,
                                                 bool emit, cx_run& run, cx_vrec* vstage, uint64_t* info = nullptr) {
    const uint32_t plane = P.n1 * P.n2;
    for (uint32_t b0 = 0; b0 < n; b0 += 64u) {
        const uint32_t idx = b0 + lane;
        const bool have = idx < n;
        const uint32_t lin = have ? lin_of(idx) : 0u;
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
        if (!emit) {
            bool flat = false;   // a crossing the fp32 interpolation of the fast path must not take
#pragma unroll
            for (uint32_t d = 1; d < 8; d++) flat |= ((R.emask >> d) & 1u) && fabsf(f[d] - f[0]) <= 1.001e-8f;
            run.s += (uint32_t)__popcll(__ballot(flat));
        }
        if (emit) {
            const uint32_t vfirst = run.v + vpre;
            if (run.v + vtot <= P.vcap) {
                if (vtot <= CX_VSTAGE) {
                    // the wave's vertices of this batch are one contiguous run of the vertex array:
                    // stage them in LDS, then write full 16-byte-per-lane rows
                    if (nv) cx_emit_vertices(P, f, R.emask, lin, i, j, k, [&](uint32_t r2, const cx_vrec& rec4) { vstage[vpre + r2] = rec4; });
                    if (!(P.flags & CX_DBG_NO_VERTS))
                        for (uint32_t o = lane; o < vtot; o += 64u) P.verts[run.v + o] = vstage[o];
                } else if (nv && !(P.flags & CX_DBG_NO_VERTS)) {
                    cx_emit_vertices(P, f, R.emask, lin, i, j, k, [&](uint32_t r2, const cx_vrec& rec4) { P.verts[vfirst + r2] = rec4; });
                }
                // (first vertex, crossing mask) of the cell for the triangle stage: per queue entry (staged pipeline), or in
                // the table of one entry per sample (generic classify kernel)
                if (info) { if (have && !(P.flags & CX_DBG_NO_CELLTAB)) info[idx] = cx_info_word(vfirst, R.emask, R.ntri, R.tetskip); }
                else if (nv && !(P.flags & CX_DBG_NO_CELLTAB)) P.celltab[lin] = ((uint64_t)R.emask << 32) | (uint64_t)vfirst;
            }
            if (RECORDS && (P.write_records || !info) && rec && run.c + ctot <= P.ccap && !(P.flags & CX_DBG_NO_CELLS)) {
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

// ---- packed queue entries of the staged pipeline (one u32 per active cell):
//   bits 0-9    corner signs as extracted from the packed plane words: bits 0,1 = corners 0,1;
//               bits 2,3 = corners 4,5; bits 6,7 = corners 2,3; bits 8,9 = corners 6,7
//   bits 10-14  position of the cell inside its lane's packed word (6*row + m)
//   bits 15-20  streaming lane (k = k0 + 4*lane + m)
//   bits 21-27  plane offset inside the task
// Where a wave waits for the loads it issued at the top of an iteration: before the iteration's stores (1: rounds 1 and 2 --
// a wait issued behind stores retires them too, `s_waitcnt vmcnt` counts loads and stores in issue order) or after them (0).
// tools/micro/mix_rate.hip settles it: 4 scattered gathers + 6 coalesced stores per iteration cost 1011 cycles per CU when the
// wave waits for the gathers before it stores and 662 when it stores first -- the stores of one iteration and the gathers of
// the next then travel together instead of one after the other.
#ifndef CX_PIN_BEFORE_STORES
#define CX_PIN_BEFORE_STORES 0
#endif
#ifndef CX_VR
#define CX_VR 4u          // rounds of 64 vertices whose loads are issued together
#endif
struct cx_fast_geom {
    uint32_t pstart, j0, k0;
};
__device__ __forceinline__ void cx_decode_entry(const cx_params& P, const cx_fast_geom& G, uint32_t e, uint32_t& i,
                                                uint32_t& j, uint32_t& k) {
    const uint32_t bit = (e >> 10) & 31u;
    const uint32_t r = (bit * 11u) >> 6;   // bit / 6 for bit < 32
    i = G.pstart + ((e >> 21) & 127u);
    j = G.j0 + r;
    k = G.k0 + 4u * ((e >> 15) & 63u) + (bit - 6u * r);
}
__device__ __forceinline__ uint32_t cx_entry_lin(const cx_params& P, const cx_fast_geom& G, uint32_t e) {
    uint32_t i, j, k;
    cx_decode_entry(P, G, e, i, j, k);
    return (i * P.n1 + j) * P.n2 + k;
}
__device__ __forceinline__ uint32_t cx_entry_signs(uint32_t e) {
    return (e & 3u) | ((e >> 4) & 0xCu) | ((e << 2) & 0x30u) | ((e >> 2) & 0xC0u);
}
// validity mask of the 8 corners of cell (i,j,k)
__device__ __forceinline__ uint32_t cx_corner_valid(const cx_params& P, uint32_t i, uint32_t j, uint32_t k) {
    const bool vi = (i + 1u < P.n0), vj = (j + 1u < P.n1), vk = (k + 1u < P.n2);
    uint32_t vm = 1u | (vk ? 2u : 0u) | (vj ? 4u : 0u) | ((vj && vk) ? 8u : 0u);
    vm |= vi ? (vm << 4) : 0u;
    return vm;
}

// vertex records, per-cell table entries and cell records of n queued cells, common case (no sample of
// the streaming wave's region within the tolerance screen, so the corner signs decide everything): table
// entries and cell records one lane per CELL, vertices one lane per VERTEX (two sample loads, one division,
// one coalesced 16-byte store) -- no divergent per-direction loop.
// Vertex records and triangles are not read again by the pipeline: nontemporal stores keep them from sitting dirty
// in L2 / Infinity Cache until the NEXT extraction's stream kernel has to push them out (measured: stream kernel
// 0.148 -> 0.129 ms inside the pipeline, whole extraction -5 %).  Table entries and cell records are read by the
// triangle kernel right away and stay ordinary stores.  -DCX_NT_VERTS=0 / -DCX_NT_TRIS=0 for A/B.
#ifndef CX_NT_VERTS
#define CX_NT_VERTS 1
#endif
#ifndef CX_NT_TRIS
#define CX_NT_TRIS 1
#endif
#ifndef CX_S1_SHIFT_OR
#define CX_S1_SHIFT_OR 0
#endif
#ifndef CX_NT_GRID
#define CX_NT_GRID 0
#endif
typedef float cx_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t cx_v2u __attribute__((ext_vector_type(2)));
#if CX_NT_VERTS
#define CX_STORE_VERT(ptr, val) __builtin_nontemporal_store(cx_v2u{(val).x, (val).y}, reinterpret_cast<cx_v2u*>(ptr))
#else
#define CX_STORE_VERT(ptr, val) (*(ptr) = (val))
#endif
// one round of 64 queued cells, as the vertex stage carries it from its front half (decode, prefix
// sums, slot table, corner loads issued) to its back half (interpolation, stores)
struct cx_vround {
    uint32_t e, lin, sm, emask, ntri, vpre, tpre, vtot, ttot, ctot, real_voxel, vk;
    uint64_t recm;
    cx_run base;                       // first vertex / triangle / record of the round
    float f[8];                        // the 8 corner samples of the lane's cell
    uint32_t e_next;                   // entries of the following round
};
#define CX_CORNER_ROW 9u               // dwords per cell in the LDS corner table (8 + 1: bank spread)
__device__ __forceinline__ void cx_vround_front(const cx_params& P, const cx_fast_geom& G, const uint32_t* q, uint32_t n, uint32_t b0,
                                                uint32_t lane, uint32_t e, const cx_run& base, uint32_t* slot,
                                                const uint8_t* ntri_lut, cx_vround& R) {
    const uint32_t idx = b0 + lane;
    const bool have = idx < n;
    R.e = e;
    R.e_next = (idx + 64u < n) ? q[idx + 64u] : 0u;
    uint32_t i, j, k;
    cx_decode_entry(P, G, e, i, j, k);
    R.lin = (i * P.n1 + j) * P.n2 + k;
    R.sm = cx_entry_signs(e);
    // The samples the round's vertices interpolate between, loaded per CELL: the (k, k+1) pairs of the 4 (i,j) rows of
    // the voxel = 4 loads per lane whose addresses follow the queue order (runs along k inside a few rows).  One lane per
    // VERTEX loading its own two samples, as this stage did first, touched 2.7 x the cache lines: the vector L1 takes
    // about one line per 3-4 cycles per CU, and that -- not bytes -- bounds these kernels.
    // The loaded pairs stay untouched until the back half (cx_vround_corners): any arithmetic on them here would make the
    // wave wait for the gathers right where it issues them (`s_waitcnt vmcnt` in front of the first use) instead of
    // interpolating the previous round's vertices meanwhile.
    uint32_t vm = cx_corner_valid(P, i, j, k);
    R.vk = (vm >> 1) & 1u;
    if (P.flags & CX_DBG_NO_VLOADS) {
#pragma unroll
        for (int c = 0; c < 8; c++) R.f[c] = (float)(c + 1);
        R.vk = 1u;
    } else {
        const float* __restrict__ A = P.grid;
        const uint32_t lin = have ? R.lin : 0u;
        const uint32_t oi = (have && (vm & 0x10u)) ? P.n1 * P.n2 : 0u, oj = (have && (vm & 4u)) ? P.n2 : 0u;
        // at the array edge in k read the pair (k-1, k) instead and repeat k (as cx_load_corners does)
        const uint32_t base = (R.vk || !have) ? lin : lin - 1u;
        const cx_f2 p0 = *reinterpret_cast<const cx_f2*>(A + base);
        const cx_f2 p1 = *reinterpret_cast<const cx_f2*>(A + base + oj);
        const cx_f2 p2 = *reinterpret_cast<const cx_f2*>(A + base + oi);
        const cx_f2 p3 = *reinterpret_cast<const cx_f2*>(A + base + oi + oj);
        R.f[0] = p0.x; R.f[1] = p0.y; R.f[2] = p1.x; R.f[3] = p1.y;
        R.f[4] = p2.x; R.f[5] = p2.y; R.f[6] = p3.x; R.f[7] = p3.y;
        if (!have) { vm = 0; R.vk = 1u; }
    }
    R.real_voxel = (have && vm == 0xFFu) ? 1u : 0u;
    const uint32_t s0 = (R.sm & 1u) ? 0xFFu : 0u;
    R.emask = have ? (((R.sm ^ s0) & vm) & 0xFEu) : 0u;
    R.ntri = R.real_voxel ? (uint32_t)ntri_lut[R.sm] : 0u;
    const uint32_t nv = __popc(R.emask);
    R.vpre = cx_wave_prefix_small<3>(nv, R.vtot);
    R.tpre = cx_wave_prefix_small<4>(R.ntri, R.ttot);
    R.recm = __ballot((nv | R.ntri) != 0u);
    R.ctot = (uint32_t)__popcll(R.recm);
    R.base = base;
    // vertex o of this round (o = vpre + rank of d in emask) -> (cell lane, direction d)
#pragma unroll
    for (uint32_t d = 1; d < 8; d++)
        if ((R.emask >> d) & 1u) slot[R.vpre + __popc(R.emask & ((1u << d) - 1u))] = (lane << 3) | d;
    __builtin_amdgcn_wave_barrier();
}
// the loads issued by the front half have to be complete here
__device__ __forceinline__ void cx_vround_pin(cx_vround& R) {
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) asm volatile("" : "+v"(R.f[c]) :: "memory");
    asm volatile("" : "+v"(R.e_next) :: "memory");
}
__device__ __forceinline__ cx_vrec cx_vertex_record(const cx_params& P, const cx_fast_geom& G, uint32_t e2, uint32_t d, float f0, float f1) {
    const uint32_t lin2 = cx_entry_lin(P, G, e2);
    // same arithmetic as cx_emit_vertices; |f1 - f0| > 1e-8 here (cx_k_stream keeps waves with flatter crossings off this path)
    const float t = cx_fraction((P.vhi - f0) + P.vlo, f1 - f0);
    return make_uint2((lin2 << 3) | d, __float_as_uint(t));
}

// vertex records, (first vertex, crossing mask) words and cell records of n queued cells, common case (no sample of
// the streaming wave's region within the tolerance screen, or none that changes anything: the corner signs decide
// everything): words and cell records one lane per CELL, vertices one lane per VERTEX (its two samples from the round's
// corner table in LDS, one division, one coalesced 16-byte store) -- no divergent per-direction loop.  Rounds of 64 cells
// are software-pipelined: the corner loads of round s+1 are issued, and waited for, before the stores of round s go
// out (`s_waitcnt vmcnt` retires loads and stores in issue order: a load behind a store waits for it).
// LDS per wave: two slot tables of 448 words (double buffered) and the corner table of 64 x CX_CORNER_ROW words.
// `first`: the cell the walk starts at (a multiple of 64; `run` holds what precedes it), `n`: where it ends -- a wave takes a
// range of a batch's rounds (cx_k_emit_vertices)
__device__ __forceinline__ void cx_emit_queue_fast(const cx_params& P, const cx_fast_geom& G, const uint32_t* q, uint32_t first, uint32_t n,
                                                   uint32_t lane, cx_run run, uint32_t* slot2, const uint8_t* ntri_lut, uint64_t* info, unsigned long long* tacc = nullptr) {
    float* corners = reinterpret_cast<float*>(slot2 + 2u * 448u);
    cx_vround Ra, Rb;
    uint32_t e0 = (first + lane < n) ? q[first + lane] : 0u;
    asm volatile("" : "+v"(e0) :: "memory");
    cx_vround_front(P, G, q, n, first, lane, e0, run, slot2, ntri_lut, Ra);
    cx_vround_pin(Ra);
    uint32_t par = 0;
#ifdef CX_S3_STAMPS
#define CX_S3_T(k) { const unsigned long long tn = __builtin_amdgcn_s_memrealtime(); tacc[k] += tn - tprev; tprev = tn; }
    unsigned long long tprev = __builtin_amdgcn_s_memrealtime();
#else
#define CX_S3_T(k)
#endif
    for (uint32_t b0 = first; b0 < n; b0 += 64u) {
        const bool more = b0 + 64u < n;   // wave-uniform
        if (more) {
            cx_run nb = Ra.base;
            nb.v += Ra.vtot; nb.t += Ra.ttot; nb.c += Ra.ctot;
            cx_vround_front(P, G, q, n, b0 + 64u, lane, Ra.e_next, nb, slot2 + (par ^ 1u) * 448u, ntri_lut, Rb);
        }
        CX_S3_T(0)
        // back half of round b0: the corner samples of the 64 cells go to LDS ...
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) corners[lane * CX_CORNER_ROW + c] = ((c & 1u) || Ra.vk) ? Ra.f[c] : Ra.f[c + 1u];   // no k+1: the pair is (k-1, k)
        __builtin_amdgcn_wave_barrier();
        const bool vroom = Ra.base.v + Ra.vtot <= P.vcap;   // wave-uniform
        const uint32_t* slot = slot2 + par * 448u;
        // ... the first CX_VR x 64 vertices are interpolated before the next round's loads are waited for ...
        cx_vrec rec4[CX_VR];
#pragma unroll
        for (uint32_t r = 0; r < CX_VR; r++) {
            rec4[r] = make_uint2(0u, 0u);
            if (64u * r >= Ra.vtot) continue;   // wave-uniform
            const uint32_t o = 64u * r + lane;
            const uint32_t sl = slot[(o < Ra.vtot) ? o : 0u];
            const uint32_t cell = sl >> 3, d = sl & 7u;
            const uint32_t e2 = (uint32_t)__shfl((int)Ra.e, (int)cell);
            rec4[r] = cx_vertex_record(P, G, e2, d, corners[cell * CX_CORNER_ROW], corners[cell * CX_CORNER_ROW + d]);
        }
        CX_S3_T(1)
#if CX_PIN_BEFORE_STORES
        if (more) cx_vround_pin(Rb);
#endif
        CX_S3_T(2)
        // ... then the stores
        if (vroom) {
#pragma unroll
            for (uint32_t r = 0; r < CX_VR; r++) {
                const uint32_t o = 64u * r + lane;
                if (o < Ra.vtot && !(P.flags & CX_DBG_NO_VERTS)) CX_STORE_VERT(&P.verts[Ra.base.v + o], rec4[r]);
            }
            for (uint32_t o0 = 64u * CX_VR; o0 < Ra.vtot; o0 += 64u) {   // more than CX_VR x 64 vertices: the rest one round at a time
                const uint32_t o = o0 + lane;
                const uint32_t sl = slot[(o < Ra.vtot) ? o : 0u];
                const uint32_t cell = sl >> 3, d = sl & 7u;
                const uint32_t e2 = (uint32_t)__shfl((int)Ra.e, (int)cell);
                const cx_vrec r4 = cx_vertex_record(P, G, e2, d, corners[cell * CX_CORNER_ROW], corners[cell * CX_CORNER_ROW + d]);
                if (o < Ra.vtot && !(P.flags & CX_DBG_NO_VERTS)) CX_STORE_VERT(&P.verts[Ra.base.v + o], r4);
            }
            // (first vertex, crossing mask) of every queued cell, one 8-byte word per queue entry: 512 contiguous bytes per round
            // (was: scattered into a table of one entry per sample -- a cache line per cell, written and later gathered)
            if (b0 + lane < n && !(P.flags & CX_DBG_NO_CELLTAB))
                info[b0 + lane] = cx_info_word(Ra.base.v + Ra.vpre, Ra.emask, Ra.ntri, Ra.real_voxel ? 0u : 0x3Fu);
        }
        if (P.write_records && (Ra.emask | Ra.ntri) != 0u && Ra.base.c + Ra.ctot <= P.ccap && !(P.flags & CX_DBG_NO_CELLS)) {
            uint4 c4;
            c4.x = Ra.lin;
            c4.y = Ra.sm | ((Ra.real_voxel ? 0u : 0x3Fu) << 8) | (Ra.ntri << 16) | (Ra.emask << 24);
            c4.z = Ra.base.t + Ra.tpre;
            c4.w = Ra.base.v + Ra.vpre;
            P.cells[Ra.base.c + cx_mbcnt(Ra.recm)] = c4;
        }
        __builtin_amdgcn_wave_barrier();
#if !CX_PIN_BEFORE_STORES
        if (more) cx_vround_pin(Rb);   // the next round's samples are waited for AFTER this round's stores went out (see CX_PIN_BEFORE_STORES)
#endif
        CX_S3_T(3)
        if (more) Ra = Rb;
        par ^= 1u;
    }
}

// The same walk with the interpolation fractions READ from the stream kernel's stream (P.tq) instead of computed from gathered
// samples: vertex n of a streaming wave has its fraction at tq[w * wcap + n], so a round's fractions are one coalesced run.  What
// is left of the vertex stage is bookkeeping: the prefix sums of a round, the slot table that turns "vertex o of the round" into
// (cell, direction), the edge id, and the stores.  `tq_wave`: the wave's region minus its first vertex (indexed by GLOBAL vertex).
struct cx_tround {
    uint32_t e, lin, sm, emask, ntri, vpre, tpre, vtot, ttot, ctot, real_voxel;
    uint64_t recm;
    cx_run base;
    uint32_t tv[CX_VR];                // fractions of the round's first CX_VR x 64 vertices, as loaded
    uint32_t e_next;
};
__device__ __forceinline__ void cx_tround_front(const cx_params& P, const cx_fast_geom& G, const uint32_t* q, uint32_t n, uint32_t b0,
                                                uint32_t lane, uint32_t e, const cx_run& base, uint32_t* slot, const uint8_t* ntri_lut,
                                                const uint32_t* tq_wave, cx_tround& R) {
    const uint32_t idx = b0 + lane;
    const bool have = idx < n;
    R.e = e;
    R.e_next = (idx + 64u < n) ? q[idx + 64u] : 0u;
    uint32_t i, j, k;
    cx_decode_entry(P, G, e, i, j, k);
    R.lin = (i * P.n1 + j) * P.n2 + k;
    R.sm = cx_entry_signs(e);
    const uint32_t vm = have ? cx_corner_valid(P, i, j, k) : 0u;
    R.real_voxel = (have && vm == 0xFFu) ? 1u : 0u;
    const uint32_t s0 = (R.sm & 1u) ? 0xFFu : 0u;
    R.emask = have ? (((R.sm ^ s0) & vm) & 0xFEu) : 0u;
    R.ntri = R.real_voxel ? (uint32_t)ntri_lut[R.sm] : 0u;
    const uint32_t nv = __popc(R.emask);
    R.vpre = cx_wave_prefix_small<3>(nv, R.vtot);
    R.tpre = cx_wave_prefix_small<4>(R.ntri, R.ttot);
    R.recm = __ballot((nv | R.ntri) != 0u);
    R.ctot = (uint32_t)__popcll(R.recm);
    R.base = base;
    // the round's fractions: requested here, used in the back half (no arithmetic on them in between: a use would wait for them here)
#pragma unroll
    for (uint32_t r = 0; r < CX_VR; r++) {
        const uint32_t o = 64u * r + lane;
        R.tv[r] = 0u;
        if (o < R.vtot && !(P.flags & CX_DBG_NO_VLOADS)) R.tv[r] = __builtin_nontemporal_load(tq_wave + base.v + o);
    }
#pragma unroll
    for (uint32_t d = 1; d < 8; d++)
        if ((R.emask >> d) & 1u) slot[R.vpre + __popc(R.emask & ((1u << d) - 1u))] = (lane << 3) | d;
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void cx_tround_pin(cx_tround& R) {
#pragma unroll
    for (uint32_t r = 0; r < CX_VR; r++) asm volatile("" : "+v"(R.tv[r]) :: "memory");
    asm volatile("" : "+v"(R.e_next) :: "memory");
}
__device__ __forceinline__ void cx_emit_queue_t(const cx_params& P, const cx_fast_geom& G, const uint32_t* q, uint32_t first, uint32_t n,
                                                uint32_t lane, cx_run run, uint32_t* slot2, const uint8_t* ntri_lut, uint64_t* info,
                                                const uint32_t* tq_wave) {
    cx_tround Ra, Rb;
    uint32_t e0 = (first + lane < n) ? q[first + lane] : 0u;
    asm volatile("" : "+v"(e0) :: "memory");
    cx_tround_front(P, G, q, n, first, lane, e0, run, slot2, ntri_lut, tq_wave, Ra);
    cx_tround_pin(Ra);
    uint32_t par = 0;
    for (uint32_t b0 = first; b0 < n; b0 += 64u) {
        const bool more = b0 + 64u < n;   // wave-uniform
        if (more) {
            cx_run nb = Ra.base;
            nb.v += Ra.vtot; nb.t += Ra.ttot; nb.c += Ra.ctot;
            cx_tround_front(P, G, q, n, b0 + 64u, lane, Ra.e_next, nb, slot2 + (par ^ 1u) * 448u, ntri_lut, tq_wave, Rb);
        }
        const bool vroom = Ra.base.v + Ra.vtot <= P.vcap;   // wave-uniform
        const uint32_t* slot = slot2 + par * 448u;
        if (vroom) {
#pragma unroll
            for (uint32_t r = 0; r < CX_VR; r++) {
                if (64u * r >= Ra.vtot) continue;   // wave-uniform
                const uint32_t o = 64u * r + lane;
                const uint32_t sl = slot[(o < Ra.vtot) ? o : 0u];
                const uint32_t cell = sl >> 3, d = sl & 7u;
                const uint32_t e2 = (uint32_t)__shfl((int)Ra.e, (int)cell);
                const cx_vrec rec = make_uint2((cx_entry_lin(P, G, e2) << 3) | d, Ra.tv[r]);
                if (o < Ra.vtot && !(P.flags & CX_DBG_NO_VERTS)) CX_STORE_VERT(&P.verts[Ra.base.v + o], rec);
            }
            for (uint32_t o0 = 64u * CX_VR; o0 < Ra.vtot; o0 += 64u) {   // more than CX_VR x 64 vertices: the rest one round at a time
                const uint32_t o = o0 + lane;
                const uint32_t sl = slot[(o < Ra.vtot) ? o : 0u];
                const uint32_t cell = sl >> 3, d = sl & 7u;
                const uint32_t e2 = (uint32_t)__shfl((int)Ra.e, (int)cell);
                const uint32_t tb = (o < Ra.vtot) ? tq_wave[Ra.base.v + o] : 0u;
                const cx_vrec rec = make_uint2((cx_entry_lin(P, G, e2) << 3) | d, tb);
                if (o < Ra.vtot && !(P.flags & CX_DBG_NO_VERTS)) CX_STORE_VERT(&P.verts[Ra.base.v + o], rec);
            }
            if (b0 + lane < n && !(P.flags & CX_DBG_NO_CELLTAB))
                info[b0 + lane] = cx_info_word(Ra.base.v + Ra.vpre, Ra.emask, Ra.ntri, Ra.real_voxel ? 0u : 0x3Fu);
        }
        if (P.write_records && (Ra.emask | Ra.ntri) != 0u && Ra.base.c + Ra.ctot <= P.ccap && !(P.flags & CX_DBG_NO_CELLS)) {
            uint4 c4;
            c4.x = Ra.lin;
            c4.y = Ra.sm | ((Ra.real_voxel ? 0u : 0x3Fu) << 8) | (Ra.ntri << 16) | (Ra.emask << 24);
            c4.z = Ra.base.t + Ra.tpre;
            c4.w = Ra.base.v + Ra.vpre;
            P.cells[Ra.base.c + cx_mbcnt(Ra.recm)] = c4;
        }
        __builtin_amdgcn_wave_barrier();
        if (more) cx_tround_pin(Rb);   // the next round's fractions are waited for AFTER this round's stores went out
        if (more) Ra = Rb;
        par ^= 1u;
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

// inclusive prefix sum over the wave with DPP row shifts / broadcasts (no LDS crossbar round trips)
__device__ __forceinline__ uint32_t cx_wave_incl_scan(uint32_t x, uint32_t lane) {
    (void)lane;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
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

// which of the 4*RJ cells of streaming lane `lane` of the wave tile at (k0, j0) exist (mr), have a k+1 neighbour inside
// the array (mk) and a j+1 neighbour (mj); bit layout of CX_CELL_MASK.  Used by the stream kernel and by the fused emit
// kernel, which recomputes the stream kernel's vertex numbering for cells of OTHER waves: one definition for both.
struct cx_lmasks {
    uint32_t mr, mk, mj;
};
__device__ __forceinline__ cx_lmasks cx_lane_masks(const cx_params& P, uint32_t k0, uint32_t j0, uint32_t lane) {
    const uint32_t kofs = k0 + 4u * lane;
    const bool lane_valid = kofs < P.n2;
    const uint32_t nvalid = lane_valid ? min(4u, P.n2 - kofs) : 0u;   // samples of the row this lane holds
    const uint32_t nrows = (j0 < P.n1) ? min((uint32_t)CX_RJ, P.n1 - j0) : 0u;
    const uint32_t last_lane = min(63u, (P.n2 - k0 - 1u) >> 2);       // lane that holds the last sample of the segment
    const bool halo_in = (k0 + 256u) < P.n2;                          // a sample right of this segment exists
    cx_lmasks M;
    M.mr = (nrows >= CX_RJ) ? CX_CELL_MASK : (CX_CELL_MASK & ((1u << (CX_ROWBITS * nrows)) - 1u));   // rows that exist
    M.mr &= cx_rowmask(CX_RJ, 1u) * ((1u << nvalid) - 1u);            // cells that exist in this lane
    if (!lane_valid) M.mr = 0;                                        // lanes right of the array own no cells
    M.mk = M.mr;
    if (lane == last_lane && !halo_in) M.mk &= ~(CX_M0_MASK << (nvalid - 1u));   // the last sample of the row has no k+1
    M.mj = 0;
#pragma unroll
    for (uint32_t r = 0; r < CX_RJ; r++)
        if (j0 + r + 1u < P.n1) M.mj |= 0xFu << (CX_ROWBITS * r);
    M.mj &= M.mr;
    return M;
}
// crossing words of a lane: bit b of x[d] set <=> the cell at bit b owns a crossing on its edge in direction d.
// A, B = packed sign words of the cell plane and of the plane above; mi = M.mr if that plane exists, else 0.
__device__ __forceinline__ void cx_cross_words(uint32_t A, uint32_t B, const cx_lmasks& M, uint32_t mi, uint32_t x[8]) {
    x[0] = 0;
    x[1] = (A ^ (A >> 1)) & M.mk;
    x[2] = (A ^ (A >> CX_ROWBITS)) & M.mj;
    x[3] = (A ^ (A >> (CX_ROWBITS + 1u))) & M.mk & M.mj;
    x[4] = (A ^ B) & mi;
    x[5] = (A ^ (B >> 1)) & M.mk & mi;
    x[6] = (A ^ (B >> CX_ROWBITS)) & M.mj & mi;
    x[7] = (A ^ (B >> (CX_ROWBITS + 1u))) & M.mk & M.mj & mi;
}

// =================================================================================================
// generic classify kernel (any shape / alignment): one lane per cell, wave-private LDS queue, the
// workgroup reserves output space with ONE atomic per counter, then emits (per-cell path).
// =================================================================================================
__global__ __launch_bounds__(256) void cx_k_classify_generic(const cx_params P, const uint32_t cells_per_block) {
    __shared__ uint32_t s_queue[4][CX_QCAP];
    __shared__ cx_vrec s_vstage[4][CX_VSTAGE];
    __shared__ uint32_t s_tot[4][4];
    __shared__ uint32_t s_base[4];
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* q = s_queue[wave];
    uint32_t qn = 0;   // wave-uniform
    cx_cnt acc = {0, 0, 0, 0};   // per-lane counts of the queued cells (from sign masks)
    float dnear = 3.0e38f;       // per-lane: smallest |f - vcmp| among the corners of the queued cells
    const uint32_t plane = P.n1 * P.n2;
    uint32_t gbase = blockIdx.x * cells_per_block + wave * 64u;
    const uint32_t gend = min(blockIdx.x * cells_per_block + cells_per_block, P.nsamples);
    bool streaming = gbase < gend;
    for (;;) {
        // ---- phase A: stream until done or until the queue might not hold another step
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
        // ---- phase B: count, reserve, emit.  The last round of a workgroup reserves once for all
        // four waves; a wave whose queue filled up early reserves for itself (dense surfaces only).
        const bool final_round = !streaming;
        if (P.flags & CX_DBG_PHASE_A_ONLY) {
            if (final_round) break;
            qn = 0;
            continue;
        }
        cx_run run = {0, 0, 0, 0};
        auto lin_of = [&](uint32_t x) { return q[x]; };
        if (__ballot(dnear <= P.near_abs) != 0ULL) {   // wave-uniform: a sample inside the tolerance screen, count exactly
            cx_process_queue<true>(P, lin_of, qn, lane, false, run, s_vstage[wave]);
        } else {
            run.v = cx_wave_sum(acc.v); run.t = cx_wave_sum(acc.t);
            run.c = cx_wave_sum(acc.c); run.b = cx_wave_sum(acc.b);
        }
        acc.v = acc.t = acc.c = acc.b = 0;
        dnear = 3.0e38f;
        if (P.flags & CX_DBG_COUNT_ONLY) {
            if (final_round) break;
            qn = 0;
            continue;
        }
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
        cx_process_queue<true>(P, lin_of, qn, lane, true, run, s_vstage[wave]);
        qn = 0;
        if (final_round) break;
    }
}

// =================================================================================================
// staged pipeline:  S1 stream -> S2 scan -> S3 vertices -> S4 triangles
// Every streaming wave owns a queue region in global memory (one u32 per cell of its task, written
// densely from the front) and a list of batch records; nothing waits for anything inside a kernel.
// =================================================================================================
// logical task of a workgroup: the hardware deals workgroups round-robin to the 8 XCDs, each with
// its own L2; give every XCD one contiguous range of tasks so that neighbours (which share halo
// rows and planes) run on the same XCD at about the same time.
__device__ __forceinline__ uint32_t cx_task_of_block(const cx_task& T) {
    return (blockIdx.x & 7u) * T.chunk + (blockIdx.x >> 3);
}
struct cx_tile {     // what a wave covers: planes [p, ib), rows j0.., samples k0 + 4*lane..
    uint32_t k0, j0, p, ib, nrows;
};
__device__ __forceinline__ cx_tile cx_tile_of(const cx_params& P, const cx_task& T, uint32_t b, uint32_t wave) {
    cx_tile t;
    const uint32_t ks = b % T.nks; b /= T.nks;
    const uint32_t jg = b % T.njg;
    const uint32_t ic = b / T.njg;
    t.k0 = ks * 256u;
    t.j0 = jg * (4u * CX_RJ) + wave * CX_RJ;
    t.p = ic * T.ci;
    t.ib = min(t.p + T.ci, P.n0);
    t.nrows = (t.j0 < P.n1) ? min((uint32_t)CX_RJ, P.n1 - t.j0) : 0u;
    return t;
}

// ---- S1: one pass over the samples.  Lanes hold 4 consecutive k-samples of RJ+1 rows (one 16-byte
// load per row and plane, two planes in flight); the 20 sign bits of a plane go into ONE u32 per lane,
// a cell's activity and the per-lane vertex / triangle / record counts come from bitwise ops on two
// such words for 16 cells at a time.
// Queue entries and batch records are staged in LDS and copied out in bulk: `s_waitcnt vmcnt` retires
// loads, stores and atomics in issue order, so a store issued between two plane loads would make the
// second wait for the store's round trip (measured: +45 % kernel time with one store per active step).
#define CX_SQ 512u      // queue entries a wave stages; a step adds at most 1024: more than CX_SQ go in two halves of the lanes (32 lanes hold at most 512 cells)
#define CX_PLW 260u     // floats per row of a staged sample plane: 256 samples of the segment + the one right of it (+ 3: bank spread)
#define CX_SBR 32u      // batch records a wave stages: all a wave can close (at most one per CX_BATCH_MIN cells of its (CX_SWP - 1) x 1024)
static_assert((CX_SWP - 1u) * 1024u / CX_BATCH_MIN + 1u <= CX_SBR, "a streaming wave can close more batches than its LDS stage holds");
static_assert(CX_SQ >= 512u, "half a wave's lanes queue up to 512 cells per step");
// ALIGNED: n2 % 4 == 0 and a 16-byte aligned grid (rows start on 16-byte boundaries, every lane holds 4 samples
// of one row).  Otherwise the 16-byte loads are only 4-byte aligned and the lane that holds the end of a row
// loads the row's last 4 samples and shifts them into place, repeating the last one (a clamped corner).
template <bool ALIGNED>
__device__ __forceinline__ void cx_stream_tile(const cx_params& P, const cx_task& T, const uint32_t b) {
    __shared__ uint32_t s_q[4][CX_SQ];
    __shared__ uint32_t s_br[4][CX_SBR][5];
    // the two sample planes a step's cells lie between, per wave, row by row as the array has them ([CX_RJ + 1 rows][256 samples + the
    // one to the right]): what lets ONE LANE PER QUEUED CELL read its 8 corners, wherever the cell's streaming lane holds them in
    // registers -- the interpolation fractions of the crossings are computed HERE, where the samples already are, instead of being
    // gathered again from HBM by the vertex stage (round 4: those gathers pulled 274 MB of 128-byte lines per 512^3 extraction,
    // half of the grid, for 100 MB of vertex records)
    // (dynamic LDS: 41.6 KB per workgroup, asked for at launch only when the extraction hands fractions on -- without it the stream
    // kernel keeps its 11 KB and leaves the LDS of its CU to the other extraction's emit stages)
    extern __shared__ __attribute__((aligned(16))) float s_pl_dyn[];
    __shared__ uint32_t s_tot[4][8];
    __shared__ uint8_t s_ntri[256];      // triangles of a voxel by its corner sign mask
#ifdef CX_S1_OCC_PAD   // experiment: fewer workgroups per CU (what does the stream kernel lose with less occupancy?)
    __shared__ uint32_t s_pad[CX_S1_OCC_PAD / 4];
    if (P.n0 == 0xFFFFFFFFu) reinterpret_cast<volatile uint32_t*>(s_pad)[threadIdx.x] = 1u;
#endif
    if (b >= T.nblocks) return;
    s_ntri[threadIdx.x] = cx_d_voxel_ntri[threadIdx.x];
    if (b == 0u && threadIdx.x < 3u && P.torder) P.counters[CX_CNT_TCLS + threadIdx.x] = 0u;   // the scan kernel counts the tile kernel's work classes into these
    __syncthreads();
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* q = s_q[wave];
    uint32_t ql = 0, qflushed = 0;      // wave-uniform: entries staged / already in global memory
    uint32_t nbl = 0, nbflushed = 0;    // the same for batch records
    const uint32_t w = b * 4u + wave;
    unsigned long long* stamp = P.stamps ? P.stamps + (size_t)w * 4u : nullptr;
#ifndef CX_S3_STAMPS
    if (stamp && lane == 0) stamp[0] = __builtin_amdgcn_s_memtime();
#endif
    const cx_tile tile = cx_tile_of(P, T, b, wave);
    const uint32_t k0 = tile.k0, j0 = tile.j0, ib = tile.ib, nrows = tile.nrows;
    uint32_t p = tile.p;
    uint32_t* __restrict__ gq = P.queue + (size_t)w * T.wcap;
    cx_brec* __restrict__ brec = P.brec + (size_t)w * T.bcap;
    const float* __restrict__ A = P.grid;
    cx_cnt acc = {0, 0, 0, 0};   // per-lane counts of the cells queued since the last batch record (b: all)
    uint32_t qn = 0, qstart = 0, nb = 0;   // wave-uniform: queued cells, start of the open batch, closed batches
    uint32_t rv = 0, rt = 0, rc = 0;       // wave-uniform: vertices / triangles / records of the closed batches
    uint32_t nr = 0;                       // wave-uniform: rounds of 64 cells of the closed batches
    uint32_t tv = 0;                       // wave-uniform: fractions written to the wave's stream so far (== vertices of its cells)
    bool overflow_t = false;               // wave-uniform: the stream of fractions ran out of room
    float dnear = 3.0e38f;       // per-lane: smallest |f - vcmp| among the samples seen (tolerance screen)
    cx_fast_geom G;
    G.pstart = p; G.j0 = j0; G.k0 = k0;
    // P.qlimit: entries this wave's region holds.  A single extraction gives every wave room for all its cells (T.wcap); the levels of
    // cx_extract3d_levels share ONE pool, each with a slice of every wave's region -- a wave that finds more cells than its slice
    // holds stops storing them and raises the overflow flag (chunk slot 7): the host then gives that call full-size regions.
    bool overflow = false;
    // The counts of the queued cells (vertices, triangles, records, border voxels: what the batch records and the scan need) are only
    // ever used as sums over a batch, so they need not be taken step by step -- with ~30 cells per step half of the lanes of a
    // counting round stood idle.  Staged entries are counted in FULL rounds of 64 right before they leave the stage or a batch is
    // closed (CX_S1_LAZY; the fraction stream P.tq works step by step and keeps the old form).
    uint32_t qc = 0;                 // wave-uniform: staged entries already counted
    // all eight corners of every cell of this wave's tile inside the array: the validity mask is 0xFF (wave-uniform)
    const bool interior = (tile.ib < P.n0) && (j0 + (uint32_t)CX_RJ < P.n1) && (k0 + 256u < P.n2);
    auto count_pending = [&]() {
        __builtin_amdgcn_wave_barrier();
        for (uint32_t o0 = qc; o0 < ql; o0 += 64u) {       // wave-uniform
            const uint32_t o = o0 + lane;
            const bool have = o < ql;
            const uint32_t e = have ? q[o] : 0u;
            uint32_t vm = have ? 0xFFu : 0u;
            if (!interior) {
                uint32_t ci, cj, ck;
                cx_decode_entry(P, G, e, ci, cj, ck);
                vm = have ? cx_corner_valid(P, ci, cj, ck) : 0u;
            }
            const uint32_t sm = cx_entry_signs(e);
            const uint32_t s0 = (sm & 1u) ? 0xFFu : 0u;
            const uint32_t nv = __popc(((sm ^ s0) & vm) & 0xFEu);
            const bool real = (vm == 0xFFu);
            const uint32_t nt = real ? (uint32_t)s_ntri[sm] : 0u;
            acc.v += nv;
            acc.t += nt;
            acc.c += (nv | nt) ? 1u : 0u;
            acc.b += real ? 1u : 0u;
        }
        qc = ql;
    };
    auto flush_queue = [&]() {
#if CX_S1_LAZY
        if (!P.tq) count_pending();
#endif
        __builtin_amdgcn_wave_barrier();
        if (qflushed + ql <= P.qlimit) {
            for (uint32_t o = lane; o < ql; o += 64u) gq[qflushed + o] = q[o];
            qflushed += ql;
        } else {
            overflow = true;
        }
        __builtin_amdgcn_wave_barrier();
        ql = 0; qc = 0;
    };
    auto flush_brec = [&]() {
        __builtin_amdgcn_wave_barrier();
        if (lane < nbl) {
            cx_brec R;
            R.qoff = s_br[wave][lane][0]; R.n = s_br[wave][lane][1]; R.vpre = s_br[wave][lane][2];
            R.tpre = s_br[wave][lane][3]; R.cpre = s_br[wave][lane][4]; R.near = 0; R.pad0 = 0; R.pad1 = 0;
            brec[nbflushed + lane] = R;
        }
        __builtin_amdgcn_wave_barrier();
        nbflushed += nbl;
        nbl = 0;
    };
    auto close_batch = [&]() {
#if CX_S1_LAZY
        if (!P.tq) count_pending();
#endif
        const uint32_t bv = cx_wave_sum(acc.v), bt = cx_wave_sum(acc.t), bc = cx_wave_sum(acc.c);
        if (lane == 0) {
            s_br[wave][nbl][0] = qstart; s_br[wave][nbl][1] = qn - qstart; s_br[wave][nbl][2] = rv;
            s_br[wave][nbl][3] = rt; s_br[wave][nbl][4] = rc;
        }
        nbl++; nb++; rv += bv; rt += bt; rc += bc;
        nr += (qn - qstart + 63u) >> 6;
        qstart = qn;
        acc.v = acc.t = acc.c = 0;
    };
    if (nrows != 0u && p < ib) {
        const uint32_t kofs = k0 + 4u * lane;
        const bool lane_valid = kofs < P.n2;
        const uint32_t nvalid = lane_valid ? min(4u, P.n2 - kofs) : 0u;   // samples of the row this lane holds (ALIGNED: 4 or 0)
        const uint32_t kofs_c = (nvalid == 4u) ? kofs : (P.n2 - 4u);      // other lanes (re-)read the row's last 4 samples
        const uint32_t kshift = kofs - kofs_c;                            // ... and shift them into place (1..3; garbage for lanes off the row)
        const uint32_t last_lane = (uint32_t)__popcll(__ballot(lane_valid)) - 1u;
        const bool halo_in = (k0 + 256u) < P.n2;                  // a sample right of this segment exists
        // cells that exist / whose k+1 / j+1 neighbour exists (bit layout of CX_CELL_MASK)
        const cx_lmasks LM = cx_lane_masks(P, k0, j0, lane);
        const uint32_t mr = LM.mr, mk = LM.mk, mj = LM.mj;
        (void)mk; (void)mj;

        // one sample plane of this lane: RJ+1 rows x 4 consecutive k-samples, plus the sample right of the segment
        struct plane_raw {
            float4 v[CX_RJ + 1];
            float hv[CX_RJ + 1];
        };
        const uint32_t plane_last = min(ib, P.n0 - 1u);
        auto load_plane = [&](uint32_t pp, plane_raw& R) {
            const uint32_t pc = min(pp, plane_last);   // never beyond the last plane this task needs: a request past it re-reads that plane (a cache hit)
#pragma unroll
            for (int r = 0; r <= CX_RJ; r++) {
                const uint32_t jr = min(j0 + (uint32_t)r, P.n1 - 1u);   // rows beyond the array repeat the last row
                const uint32_t rowofs = (pc * P.n1 + jr) * P.n2;
#if CX_NT_GRID   // A/B: streaming loads of the grid (do not keep it in L2 / Infinity Cache)
                {
                    const cx_v4f t4 = __builtin_nontemporal_load(reinterpret_cast<const cx_v4f*>(A + rowofs + kofs_c));
                    R.v[r] = make_float4(t4.x, t4.y, t4.z, t4.w);
                }
#else
                R.v[r] = *reinterpret_cast<const float4*>(A + rowofs + kofs_c);      // lanes right of the array re-read its last 4 samples
#endif
                R.hv[r] = A[rowofs + (halo_in ? k0 + 256u : 0u)];                    // wave-uniform address
            }
        };
        // sign bits of a loaded plane.  f < vcmp  <=>  sign bit of (f - vcmp)  (f == vcmp gives +0 -- also for f = -0.0 at isovalue 0:
        // the host passes that threshold as -0.0, cx_fill_value_params; NaN samples are not supported); the same differences feed the
        // tolerance screen (smallest |f - vcmp| seen)
        float (*ring)[CX_RJ + 1][CX_PLW] = reinterpret_cast<float (*)[CX_RJ + 1][CX_PLW]>(s_pl_dyn + (size_t)wave * (2u * (CX_RJ + 1u) * CX_PLW));
        const bool stage_t = P.tq != nullptr;      // wave-uniform: this extraction hands fractions on (else the vertex stage gathers samples)
#ifdef CX_FORCE_RING      // experiment: what the staging of the planes costs by itself
        const bool stage_ring = true;
#else
        const bool stage_ring = stage_t;
#endif
        auto plane_bits = [&](const plane_raw& R, const uint32_t slot) -> uint32_t {
            // The sign bits are shifted in from the right, last row first: one v_alignbit_b32 per sample ({word, difference} >> 31 = the
            // word moved up by one with the sign bit of the difference behind it) instead of a shift and a shift-or; rows are
            // CX_ROWBITS apart: two more places after each row of four.  (The stream kernel is bound by its instructions.)
            uint32_t own = 0, halo = 0;
#pragma unroll
            for (int r = CX_RJ; r >= 0; r--) {
                float vx = R.v[r].x, vy = R.v[r].y, vz = R.v[r].z;
                const float vw = R.v[r].w;
                if (!ALIGNED) {   // lane at the end of the row: samples kofs.. of (n2-4 .. n2-1), the last one repeated
                    vx = (kshift == 1u) ? vy : ((kshift == 2u) ? vz : ((kshift == 3u) ? vw : vx));
                    vy = (kshift == 0u) ? vy : ((kshift == 1u) ? vz : vw);
                    vz = (kshift == 0u) ? vz : vw;
                }
                if (stage_ring) {   // the lane's four samples of row r, in place; the sample right of the segment behind them
                    *reinterpret_cast<float4*>(&ring[slot][r][4u * lane]) = make_float4(vx, vy, vz, vw);
                    if (lane == 0u) ring[slot][r][256] = R.hv[r];
                }
                const float dx = vx - P.vcmp, dy = vy - P.vcmp, dz = vz - P.vcmp, dw = vw - P.vcmp;
                const float dh = R.hv[r] - P.vcmp;
#if CX_S1_SHIFT_OR      // A/B: a shift and a shift-or per sample
                own |= (__float_as_uint(dx) >> 31) << (CX_ROWBITS * r + 0);
                own |= (__float_as_uint(dy) >> 31) << (CX_ROWBITS * r + 1);
                own |= (__float_as_uint(dz) >> 31) << (CX_ROWBITS * r + 2);
                own |= (__float_as_uint(dw) >> 31) << (CX_ROWBITS * r + 3);
                halo |= (__float_as_uint(dh) >> 31) << (CX_ROWBITS * r);
#else
                own <<= (CX_ROWBITS - 4u);                                     // the two unused places of the row above
                own = __builtin_amdgcn_alignbit(own, __float_as_uint(dw), 31);   // bit 3 of the row
                own = __builtin_amdgcn_alignbit(own, __float_as_uint(dz), 31);
                own = __builtin_amdgcn_alignbit(own, __float_as_uint(dy), 31);
                own = __builtin_amdgcn_alignbit(own, __float_as_uint(dx), 31);   // bit 0
                halo <<= (CX_ROWBITS - 1u);
                halo = __builtin_amdgcn_alignbit(halo, __float_as_uint(dh), 31);
#endif
                dnear = fminf(dnear, fminf(fminf(fabsf(dx), fabsf(dy)), fminf(fabsf(dz), fminf(fabsf(dw), fabsf(dh)))));
            }
            // k+1 neighbour of m=3: m=0 of the next lane; the last valid lane takes the halo sample or,
            // at the array edge, repeats its own m=3 (clamped corner)
            // lane l takes the word of lane l + 1 (DPP wave_shl:1, a VALU move; __shfl_down is a ds_bpermute: an LDS round trip per step)
            uint32_t nbr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own, 0x130, 0xF, 0xF, false) & CX_M0_MASK;
            if (lane == last_lane) nbr = halo_in ? halo : ((own >> 3) & CX_M0_MASK);
            return own | (nbr << 4);
        };

        // Sample planes p .. ib are needed (cell plane ib - 1 reads sample plane ib).  The prefetch runs past ib; those requests are
        // issued all the same -- a branch around them makes the compiler wait for ALL outstanding loads at the next use, which undoes
        // the prefetch (measured: 0.126 -> 0.138 ms) -- but re-read plane ib, a cache hit, instead of the neighbour task's plane ib + 1
        // (rounds 1 and 2: one plane in ci + 2, 6 % of the kernel's HBM reads).
        // CX_S1_DEPTH planes are in flight ahead of the one being worked on (a second one was tried in round 3: no gain, the kernel
        // is not starved for samples).
#if CX_S1_DEPTH == 2
        plane_raw rawA, rawB, rawC;
        load_plane(p, rawB);
        load_plane(p + 1u, rawA);
        load_plane(p + 2u, rawC);
        uint32_t wprev = plane_bits(rawB, 0u);
        // one step: plane p+1 is in `cur`, plane p+2 is on its way, plane p+3 is requested into `nxt` (whose plane is used up)
        auto step = [&](const plane_raw& cur, plane_raw& nxt) {
            load_plane(p + 3u, nxt);
#elif CX_S1_DEPTH == 3
        // TWO planes in flight with two buffers: a buffer is requested again right after its sign bits are taken (its samples are
        // dead from then on), so while a step's active cells are worked on both the next plane and the one after are on their way
        plane_raw rawA, rawB;
        load_plane(p, rawB);
        load_plane(p + 1u, rawA);
        uint32_t wprev = plane_bits(rawB, 0u);
        load_plane(p + 2u, rawB);
        auto step = [&](plane_raw& cur, plane_raw& unused_) {
            (void)unused_;
            const uint32_t wcur = plane_bits(cur, (p + 1u - G.pstart) & 1u);
            load_plane(p + 3u, cur);
#else
        plane_raw rawA, rawB;
        load_plane(p, rawB);
        load_plane(p + 1u, rawA);
        uint32_t wprev = plane_bits(rawB, 0u);
        // one step: plane p+1 is in `cur` (loaded one step ago), plane p+2 is requested into `nxt`
        auto step = [&](const plane_raw& cur, plane_raw& nxt) {
            load_plane(p + 2u, nxt);
#endif
#if CX_S1_DEPTH != 3
            const uint32_t wcur = plane_bits(cur, (p + 1u - G.pstart) & 1u);
#endif
            // per sample row: OR / AND over (k, k+1); then over rows (r, r+1); then over both planes
            const uint32_t op = wprev | (wprev >> 1), ap = wprev & (wprev >> 1);
            const uint32_t oc = wcur | (wcur >> 1), ac = wcur & (wcur >> 1);
            const uint32_t o = op | (op >> CX_ROWBITS) | oc | (oc >> CX_ROWBITS);
            const uint32_t a = ap & (ap >> CX_ROWBITS) & ac & (ac >> CX_ROWBITS);
            const uint32_t act0 = o & ~a & mr;
#ifdef CX_S1_ABL
            if (CX_S1_ABL == 2) { wprev = wcur; p++; return; }
#endif
            if (__ballot(act0 != 0u) != 0ULL) {    // wave-uniform: some cell of this step has a sign change
#if CX_S1_LAZY
                // exclusive prefix of the lanes' cell counts by DPP row shifts (9 instructions; five ballots with their mbcnt pairs: 35)
                const uint32_t cnt0 = __popc(act0);
                const uint32_t incl0 = cx_wave_incl_scan(cnt0, lane);
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl0, 63);
                const uint32_t pre = incl0 - cnt0;
#else
                uint32_t tot;
                const uint32_t pre = cx_wave_prefix_small<5>(__popc(act0), tot);
#endif
                const uint32_t sidx = p - G.pstart;
                const uint32_t ebase = (lane << 15) | (sidx << 21);
                // where the lane's cells of this step sit in the wave's queue, and which they are (bit 4r+m): lets the triangle stage
                // find the queue entry of ANY active cell of the volume without a table per sample.  Straight to global memory, one
                // coalesced 256-byte store per step that queued cells (round 3 staged all steps' words in 17 KB of LDS and wrote them
                // out at the end, active or not; measured: the direct store is 3 % FASTER, and the LDS goes to the sample planes)
#ifdef CX_S1_ABL
                if (P.qa && CX_S1_ABL != 3) (P.qa + (size_t)w * (CX_SWP * 64u))[sidx * 64u + lane] =
#else
                if (P.qa) (P.qa + (size_t)w * (CX_SWP * 64u))[sidx * 64u + lane] =
#endif
                    ((qn + pre) << 16) | (act0 & 0xFu) | ((act0 >> 2) & 0xF0u) | ((act0 >> 4) & 0xF00u) | ((act0 >> 6) & 0xF000u);
                // The step's cells go through the entry stage in the order of the lanes; more than the stage holds (CX_SQ; up to 1024
                // on white noise) go in two halves of the lanes -- the order in the queue is the same either way.
#ifdef CX_S1_ABL      // timing experiments (tools/variants.sh + tools/stream_ab.py; the queues are garbage): 1 = the cells are counted but not
                      // written to the stage, 2 = nothing of the active block at all, 3 = everything but the per-step queue word store
                if (CX_S1_ABL == 1) { qn += tot; wprev = wcur; p++; return; }
#endif
                const uint32_t nhalf = (tot > CX_SQ) ? 2u : 1u;            // wave-uniform
                const uint32_t mid = (uint32_t)__builtin_amdgcn_readlane((int)pre, 32);   // cells of lanes 0..31
                const float* __restrict__ plo = &ring[sidx & 1u][0][0];          // sample plane p   (the cells' lower corners)
                const float* __restrict__ phi = &ring[(sidx + 1u) & 1u][0][0];   // sample plane p+1
                float* __restrict__ tqw = stage_t ? P.tq + (size_t)w * T.wcap : nullptr;
                for (uint32_t h = 0; h < nhalf; h++) {
                    const uint32_t cbase = (nhalf == 2u && h == 1u) ? mid : 0u;
                    const uint32_t ctot = (nhalf == 2u) ? (h == 0u ? mid : tot - mid) : tot;
                    if (ql + ctot > CX_SQ) flush_queue();   // wave-uniform
                    uint32_t act = (nhalf == 1u || (lane >> 5) == h) ? act0 : 0u;
                    uint32_t pos = ql + pre - cbase;
                    while (act) {
                        const uint32_t bit = __ffs(act) - 1u;
                        act &= act - 1u;
                        // corners (k,k+1) of rows (r,r+1): bits (bit, bit+1, bit+6, bit+7) of the two plane words
                        q[pos++] = ebase | (bit << 10) | ((wprev >> bit) & 0xC3u) | (((wcur >> bit) & 0xC3u) << 2);
                    }
                    // One lane per queued ENTRY: the counts of the cells just queued, with the vertex stage's own formulas
                    // (cx_vround_front), and -- with the samples of both planes in LDS -- the fraction t = (v - f(q)) / (f(q+d) - f(q))
                    // of every crossing the cell owns, written to the wave's stream of fractions in the order the vertex stage numbers
                    // the vertices (cell after cell, direction after direction).
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t o0 = 0; o0 < ((CX_S1_LAZY && !stage_t) ? 0u : ctot); o0 += 64u) {      // wave-uniform
                        const uint32_t o = o0 + lane;
                        const bool have = o < ctot;
                        const uint32_t e = have ? q[ql + o] : 0u;
                        uint32_t ci, cj, ck;
                        cx_decode_entry(P, G, e, ci, cj, ck);
                        const uint32_t vm = have ? cx_corner_valid(P, ci, cj, ck) : 0u;
                        const uint32_t sm = cx_entry_signs(e);
                        const uint32_t s0 = (sm & 1u) ? 0xFFu : 0u;
                        const uint32_t emask = ((sm ^ s0) & vm) & 0xFEu;
                        const uint32_t nv = __popc(emask);
                        const bool real = (vm == 0xFFu);
                        const uint32_t nt = real ? (uint32_t)s_ntri[sm] : 0u;
                        acc.v += nv;
                        acc.t += nt;
                        acc.c += (nv | nt) ? 1u : 0u;
                        acc.b += real ? 1u : 0u;
                        if (stage_t) {      // wave-uniform
                            // corner c = 4 di + 2 dj + dk of the cell: plane di, row rr + dj, sample + dk.  All eight are requested at
                            // once, ahead of the prefix sums below (written as seven guarded loads, each crossing waited for its own
                            // LDS round trip: measured +43 us on the kernel), and all seven fractions are computed -- two
                            // instructions each -- whether the edge is crossed or not; only the stores are guarded.
                            const uint32_t bit = (e >> 10) & 31u;
                            const uint32_t rr = (bit * 11u) >> 6;
                            const uint32_t at = rr * CX_PLW + 4u * ((e >> 15) & 63u) + (bit - 6u * rr);
                            float f0 = plo[at], f1 = plo[at + 1u], f2 = plo[at + CX_PLW], f3 = plo[at + CX_PLW + 1u];
                            float f4 = phi[at], f5 = phi[at + 1u], f6 = phi[at + CX_PLW], f7 = phi[at + CX_PLW + 1u];
                            uint32_t vtot;
                            const uint32_t vpre = cx_wave_prefix_small<3>(nv, vtot);
                            if (tv + vtot <= P.tlimit) {    // wave-uniform
                                asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
                                const float num = (P.vhi - f0) + P.vlo;
                                const float t1 = cx_fraction(num, f1 - f0), t2 = cx_fraction(num, f2 - f0), t3 = cx_fraction(num, f3 - f0);
                                const float t4 = cx_fraction(num, f4 - f0), t5 = cx_fraction(num, f5 - f0), t6 = cx_fraction(num, f6 - f0);
                                const float t7 = cx_fraction(num, f7 - f0);
                                float* __restrict__ dst = tqw + tv + vpre;
                                if (emask & 0x02u) dst[0] = t1;
                                if (emask & 0x04u) dst[__popc(emask & 0x02u)] = t2;
                                if (emask & 0x08u) dst[__popc(emask & 0x06u)] = t3;
                                if (emask & 0x10u) dst[__popc(emask & 0x0Eu)] = t4;
                                if (emask & 0x20u) dst[__popc(emask & 0x1Eu)] = t5;
                                if (emask & 0x40u) dst[__popc(emask & 0x3Eu)] = t6;
                                if (emask & 0x80u) dst[__popc(emask & 0x7Eu)] = t7;
                            } else {
                                overflow_t = true;          // more crossings than the wave's region holds: its cells take the per-cell path downstream
                            }
                            tv += vtot;
                        }
                    }
                    ql += ctot;
                    qn += ctot;
                    __builtin_amdgcn_wave_barrier();
                }
                if (qn - qstart >= CX_BATCH_MIN) close_batch();
            }
            wprev = wcur;
            p++;
        };
#if CX_S1_DEPTH == 2
        while (p < ib) {
            step(rawA, rawB);
            if (p >= ib) break;
            step(rawC, rawA);
            if (p >= ib) break;
            step(rawB, rawC);
        }
#else
        while (p < ib) {
            step(rawA, rawB);
            if (p >= ib) break;
            step(rawB, rawA);
        }
#endif
    }
    if (qn > qstart) close_batch();
    flush_queue();
    flush_brec();
#ifndef CX_S3_STAMPS
    if (stamp && lane == 0) stamp[1] = __builtin_amdgcn_s_memtime();
#endif
    cx_run run;
    run.v = rv; run.t = rt; run.c = rc; run.b = cx_wave_sum(acc.b);
    const bool near = __ballot(dnear <= P.near_abs) != 0ULL && !(P.flags & CX_DBG_NO_NEAR);   // wave-uniform
    bool really_near = false;
    if (near && qn != 0u && !overflow) {
        // a sample inside the screen: the reference's tolerance rules may drop tetrahedra or vertices.
        // Count exactly (per-cell path over the wave's own queue; its stores are complete after the
        // fence) and hand the whole queue over as ONE batch that takes the per-cell path downstream.
        __threadfence();
        cx_run ex = {0, 0, 0, 0, 0};
        cx_process_queue<false>(P, [&](uint32_t x) { return cx_entry_lin(P, G, __builtin_nontemporal_load(gq + x)); }, qn, lane, false, ex, nullptr);
        // The tolerance rules only ever REMOVE crossings and triangles from what the signs give.  Equal totals therefore
        // mean equal masks in every cell: the wave stays on the common path (its batches, its numbering), only the border
        // voxel count is the exact one.  Otherwise: ONE batch that takes the per-cell path downstream.
        really_near = !(ex.v == rv && ex.t == rt && ex.c == rc && ex.s == 0u);
        run.b = ex.b;
        if (really_near) { run.v = ex.v; run.t = ex.t; run.c = ex.c; }
    }
    // a wave whose stream of fractions ran out of room (more crossings than one per cell of its tile: white noise) hands its cells
    // to the per-cell path too -- which interpolates from the samples itself; its counts are the ones the signs gave
    if (overflow_t && qn != 0u && !overflow) really_near = true;
    if (really_near) {
        nb = 1;
        nr = (qn + 63u) >> 6;
        if (lane == 0) {
            cx_brec R;
            R.qoff = 0; R.n = qn; R.vpre = 0; R.tpre = 0; R.cpre = 0; R.near = 1; R.pad0 = 0; R.pad1 = 0;
            brec[0] = R;
        }
    }
    if (lane == 0) {
        cx_wsum S;
        S.nb = nb; S.v = run.v; S.t = run.t; S.c = run.c; S.b = run.b; S.nq = qn; S.near = really_near ? 1u : 0u; S.nr = nr;
        P.wsum[w] = S;
        s_tot[wave][0] = run.v; s_tot[wave][1] = run.t; s_tot[wave][2] = run.c; s_tot[wave][3] = run.b; s_tot[wave][4] = nb;
        s_tot[wave][5] = (really_near && qn != 0u) ? 1u : 0u;
        s_tot[wave][6] = nr;
        s_tot[wave][7] = overflow ? 1u : 0u;
    }
    // totals of every 256 waves (64 workgroups), so that the scan needs ONE round of loads for everything before its chunk
    // (the per-wave totals were just written by CUs all over the chip: each dependent round of loads from them costs ~2 us).
    // One atomic per workgroup and counter.
    __syncthreads();
    if (threadIdx.x < 8u) {
        const uint32_t sum = s_tot[0][threadIdx.x] + s_tot[1][threadIdx.x] + s_tot[2][threadIdx.x] + s_tot[3][threadIdx.x];
        uint32_t* cs = P.chunksum + (size_t)(b >> 6) * 8u;
        if (sum) atomicAdd(cs + threadIdx.x, sum);      // (slot 5: number of waves on the tolerance path; non-zero = flag; slot 6: rounds; slot 7: waves whose queue slice overflowed)
    }
}
template <bool ALIGNED>
__global__ __launch_bounds__(256, CX_K1_MIN_WAVES) void cx_k_stream(const cx_params P, const cx_task T) {
    cx_stream_tile<ALIGNED>(P, T, cx_task_of_block(T));
}
// Several isovalues of ONE grid: the same pass, with workgroups numbered (tile, level), the level running fastest
// inside an XCD's sequence: the nlevels workgroups that stream one tile run next to each other on one XCD, so the tile comes
// from HBM once and from that XCD's L2 / the Infinity Cache for the other levels.  LP[level] = parameters of a level
// (isovalue, its queues and side tables).
// The parameters of up to CX_LEVELS_PER_LAUNCH levels travel BY VALUE as a kernel argument.  Read from a device array, as they were
// until the end of round 3, their pointers are generic as far as the compiler knows: loads and stores through them came out as FLAT
// instructions, which count on both memory counters -- every wait in the kernel was `vmcnt(0) lgkmcnt(0)`, the plane requested for the
// NEXT step was waited for together with the current one.  Pointers inside a kernel argument are known to be global: the waits are
// the counted ones of cx_k_stream again.  (It did not change the time -- 2.33 ms for 8 levels either way: the kernel is bound by its
// instructions, DESIGN section 10 -- but it is the code the single-level kernel runs.)
#define CX_LEVELS_PER_LAUNCH 8u
struct cx_params_pack {
    cx_params p[CX_LEVELS_PER_LAUNCH];
};
template <bool ALIGNED>
__global__ __launch_bounds__(256, CX_K1_MIN_WAVES) void cx_k_stream_levels(const cx_params_pack LP, const cx_task T, const uint32_t nlevels) {
    const uint32_t seq = blockIdx.x >> 3;                   // position in this XCD's sequence of workgroups
    const uint32_t tile_seq = seq / nlevels, level = seq - tile_seq * nlevels;
    cx_stream_tile<ALIGNED>(LP.p[level], T, (blockIdx.x & 7u) * T.chunk + tile_seq);
}

// ---- S2 in one launch: workgroup g owns the streaming waves [256 g, 256 g + 256).  It sums the totals of ALL waves before its
// chunk itself (every thread reads one wave of each earlier chunk: g x 8 KB per workgroup out of L2 -- no partial sums to
// wait for, no second kernel, no atomics), scans its own 256 waves, and writes their output offsets and batch descriptors;
// the last workgroup, which has seen every wave, writes the counters.  (One workgroup scanning all 12 k waves of a 512^3
// grid was bound by what a single CU can pull: 22 us + 5 us for the list kernel.)
__device__ __forceinline__ void cx_scan_list_chunk(const cx_params& P, const cx_task& T, const uint32_t nw, const uint32_t g, const uint32_t nchunks) {
    __shared__ uint32_t s_part[4][8];    // per wave of the workgroup: totals of the earlier chunks (6), near | overflow << 1, rounds of ALL chunks
    __shared__ uint32_t s_own[4][6];     // per wave: inclusive totals of its 64 streaming waves
    const uint32_t tid = threadIdx.x, lane = cx_lane_id(), wave = tid >> 6;
    uint32_t acc[6] = {0, 0, 0, 0, 0, 0}, near_any = 0, rall = 0, over_any = 0;   // v, t, c, b, nb, nr
    for (uint32_t c = tid; c < nchunks; c += 256u) {     // thread c: the totals of chunk c (every workgroup looks at all of them: near flag, total rounds)
        const uint4 lo = *reinterpret_cast<const uint4*>(P.chunksum + (size_t)c * 8u);
        const uint4 hi = *reinterpret_cast<const uint4*>(P.chunksum + (size_t)c * 8u + 4u);
        if (c < g) { acc[0] += lo.x; acc[1] += lo.y; acc[2] += lo.z; acc[3] += lo.w; acc[4] += hi.x; acc[5] += hi.z; }
        near_any |= hi.y;
        rall += hi.z;
        over_any |= hi.w;
    }
    const uint32_t w = g * 256u + tid;
    cx_wsum S;
    S.nb = S.v = S.t = S.c = S.b = S.nq = S.near = S.nr = 0;
    if (w < nw) S = P.wsum[w];
    near_any |= (S.near != 0u && S.nq != 0u) ? 1u : 0u;
    const uint32_t x[6] = {S.v, S.t, S.c, S.b, S.nb, S.nr};
    uint32_t inc[6];
#pragma unroll
    for (int m = 0; m < 6; m++) {
        inc[m] = cx_wave_incl_scan(x[m], lane);
        const uint32_t before = cx_wave_sum(acc[m]);
        if (lane == 63u) { s_own[wave][m] = inc[m]; s_part[wave][m] = before; }
    }
    {
        const uint32_t nr = (__ballot(near_any != 0u) != 0ULL) ? 1u : 0u;
        const uint32_t ra = cx_wave_sum(rall);
        const uint32_t ov = (__ballot(over_any != 0u) != 0ULL) ? 1u : 0u;
        if (lane == 63u) { s_part[wave][6] = nr | (ov << 1); s_part[wave][7] = ra; }
    }
    __syncthreads();
    uint32_t ex[6];
#pragma unroll
    for (int m = 0; m < 6; m++) {
        uint32_t base = s_part[0][m] + s_part[1][m] + s_part[2][m] + s_part[3][m];
        for (uint32_t ww = 0; ww < wave; ww++) base += s_own[ww][m];
        ex[m] = base + inc[m] - x[m];
    }
    // the vertex stage divides the rounds of 64 queued cells evenly among its P.nvw waves: wave m takes rounds [m q, m q + q)
    const uint32_t rounds = s_part[0][7] + s_part[1][7] + s_part[2][7] + s_part[3][7];
    const uint32_t q = max((rounds + P.nvw - 1u) / P.nvw, CX_S3_MIN_SHARE);
    const uint32_t qk = max((rounds + P.nkw - 1u) / P.nkw, CX_S3_MIN_SHARE);
    if (w < nw) {
        cx_wbase B;
        B.v = ex[0]; B.t = ex[1]; B.c = ex[2]; B.boff = ex[4];
        P.wbase[w] = B;
        // the flat batch list: self-contained descriptors
        const cx_tile tile = cx_tile_of(P, T, w >> 2, w & 3u);
        uint32_t rb = ex[5];
        for (uint32_t i = 0; i < S.nb; i++) {
            if (ex[4] + i >= P.fcap) break;
            const cx_brec R = P.brec[(size_t)w * T.bcap + i];
            cx_bdesc D;
            D.w = w; D.qofs = w * T.wcap + R.qoff; D.n = R.n; D.near = R.near;
            D.vbase = ex[0] + R.vpre; D.tbase = ex[1] + R.tpre; D.cbase = ex[2] + R.cpre; D.rbase = rb;
            D.p = tile.p; D.j0 = tile.j0; D.k0 = tile.k0; D.nsteps = tile.ib - tile.p;
            P.flat[ex[4] + i] = D;
            // the waves whose share starts inside this batch
            const uint32_t nrb = (R.n + 63u) >> 6;
            for (uint32_t m = (rb + q - 1u) / q; m * q < rb + nrb; m++) P.rstart[m] = ex[4] + i;
            for (uint32_t m = (rb + qk - 1u) / qk; m * qk < rb + nrb; m++) P.kstart[m] = ex[4] + i;   // the same for the triangle stage's waves
            rb += nrb;
        }
    }
    if (P.torder) {
        // the tile kernel's order of work (cx_tile3d.h): half tiles (the queues of streaming waves 2m, 2m + 1) by class of queue
        // entries, most first -- with the tiles in launch order the kernel ended in a tail of its heaviest workgroups (60 of 245 us
        // at 512^3) -- and the empty ones in no list at all.  One reservation per workgroup and class.
        __shared__ uint32_t s_cls[3][4], s_clsbase[3];
        const uint32_t nqp = S.nq + (uint32_t)__shfl_down((int)S.nq, 1);        // even lanes: entries of the half tile
        const bool isht = !(tid & 1u) && w < nw;
        const uint32_t cls = !isht ? 4u : (nqp >= (P.tile_cap * 5u >> 3) ? 0u : (nqp >= (P.tile_cap * 5u >> 4) ? 1u : (nqp ? 2u : 3u)));
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t c = 0; c < 3u; c++) {
            const uint64_t m = __ballot(cls == c);
            if (cls == c) rank = cx_mbcnt(m);
            if (lane == 0) s_cls[c][wave] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (tid < 3u) {
            const uint32_t n = s_cls[tid][0] + s_cls[tid][1] + s_cls[tid][2] + s_cls[tid][3];
            s_clsbase[tid] = n ? atomicAdd(&P.counters[CX_CNT_TCLS + tid], n) : 0u;
        }
        __syncthreads();
        if (cls < 3u) {
            uint32_t at = s_clsbase[cls] + rank;
            for (uint32_t ww = 0; ww < wave; ww++) at += s_cls[cls][ww];
            P.torder[(size_t)cls * (nw >> 1) + at] = w >> 1;
        } else if (cls == 3u) {
            P.bndn[w >> 1] = 0u;          // nothing crosses this half tile: no boundary voxels
        }
    }
    if (g == nchunks - 1u && tid == 255u) {   // this thread's inclusive totals are the grand totals
        P.counters[CX_CNT_VERTS] = ex[0] + x[0]; P.counters[CX_CNT_TRIS] = ex[1] + x[1];
        P.counters[CX_CNT_CELLS] = ex[2] + x[2]; P.counters[CX_CNT_BORDER] = ex[3] + x[3];
        P.counters[CX_CNT_BATCHES] = ex[4] + x[4];
        P.counters[CX_CNT_ROUNDS] = rounds;
        const uint32_t fl = s_part[0][6] | s_part[1][6] | s_part[2][6] | s_part[3][6];
        P.counters[CX_CNT_NEAR] = fl & 1u;
        P.counters[CX_CNT_OVERFLOW] = (fl >> 1) & 1u;
        P.counters[CX_CNT_TILEOVF] = 0u;
    }
}
__global__ __launch_bounds__(256) void cx_k_scan_list(const cx_params P, const cx_task T, const uint32_t nw) {
    cx_scan_list_chunk(P, T, nw, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void cx_k_scan_list_levels(const cx_params* __restrict__ LP, const cx_task T, const uint32_t nw) {
    const cx_params P = LP[blockIdx.y];
    cx_scan_list_chunk(P, T, nw, blockIdx.x, gridDim.x);
}

// what `rounds` full rounds of 64 queued cells at the front of a batch hold (vertices, triangles, records), from the queue
// entries alone -- the same formulas as cx_vround_front.  A wave that starts inside a batch (below) adds this to the batch's bases.
__device__ __forceinline__ void cx_skip_rounds(const cx_params& P, const cx_fast_geom& G, const uint32_t* q, uint32_t rounds, uint32_t lane,
                                               const uint8_t* ntri_lut, cx_run& run) {
    uint32_t v = 0, t = 0, c = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t e = q[r * 64u + lane];
        uint32_t i, j, k;
        cx_decode_entry(P, G, e, i, j, k);
        const uint32_t vm = cx_corner_valid(P, i, j, k);
        const uint32_t sm = cx_entry_signs(e);
        const uint32_t s0 = (sm & 1u) ? 0xFFu : 0u;
        const uint32_t nv = __popc(((sm ^ s0) & vm) & 0xFEu);
        const uint32_t nt = (vm == 0xFFu) ? (uint32_t)ntri_lut[sm] : 0u;
        v += nv; t += nt; c += (nv | nt) ? 1u : 0u;
    }
    run.v += cx_wave_sum(v); run.t += cx_wave_sum(t); run.c += cx_wave_sum(c);
}

// ---- S3: vertex records, (first vertex, crossing mask) words and cell records.  The ROUNDS of 64 queued cells of all batches
// are divided evenly among the waves of the grid (wave m: rounds [m q, m q + q) of the flat batch list; the scan kernel left the
// batch each share starts in, P.rstart): batches hold 1 to 24 rounds, and with whole batches dealt out to waves -- as this
// kernel did first -- it lasted as long as its unluckiest wave (25 rounds against a mean of 7) while the chip stood half empty
// for the last third of its time.  A wave that starts inside a batch counts what the rounds before hold (cx_skip_rounds).
// A batch on the tolerance path (per-cell path, counts not derivable from the signs) goes as a whole to the wave in whose
// share its first round falls.
#ifndef CX_S3_MIN_WAVES
#define CX_S3_MIN_WAVES 1
#endif
template <bool TQ>     // TQ: the interpolation fractions come from the stream kernel's stream (P.tq); else the samples are gathered here
__global__ __launch_bounds__(256, CX_S3_MIN_WAVES) void cx_k_emit_vertices(const cx_params P, const cx_task T) {
#ifdef CX_OCC_PAD   // experiment: fewer workgroups per CU (what does the time do with the occupancy?)
    __shared__ uint32_t s_pad[CX_OCC_PAD / 4];
    if (P.n0 == 0xFFFFFFFFu) reinterpret_cast<volatile uint32_t*>(s_pad)[threadIdx.x] = 1u;
#endif
    __shared__ float4 s_vstage[4][CX_VSTAGE_LDS];
    __shared__ uint8_t s_ntri[256];
    s_ntri[threadIdx.x] = cx_d_voxel_ntri[threadIdx.x];
    __syncthreads();
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nbatches = min(P.counters[CX_CNT_BATCHES], P.fcap);
    const uint32_t rounds = P.counters[CX_CNT_ROUNDS];
    const uint32_t nvw = gridDim.x * 4u;     // == P.nvw
    const uint32_t share = max((rounds + nvw - 1u) / nvw, CX_S3_MIN_SHARE);
    const uint32_t lo = (blockIdx.x * 4u + wave) * share;
    if (lo >= rounds || nbatches == 0u) return;
    const uint32_t hi = min(rounds, lo + share);
    uint32_t f = P.rstart[blockIdx.x * 4u + wave];
    if (f >= nbatches) return;
    cx_bdesc D = P.flat[f];
#ifdef CX_S3_STAMPS
    unsigned long long tacc[4] = {0, 0, 0, 0}, nrounds = 0;
    const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
#endif
    for (;;) {   // waves are independent
        const uint32_t fn = f + 1u;
        cx_bdesc Dn = D;
        if (fn < nbatches) Dn = P.flat[fn];   // next descriptor in flight while this batch is processed
        const uint32_t nrb = (D.n + 63u) >> 6;
        const uint32_t r0 = max(lo, D.rbase) - D.rbase, r1 = min(hi, D.rbase + nrb) - D.rbase;   // this wave's rounds of the batch
        cx_fast_geom G;
        G.pstart = D.p; G.j0 = D.j0; G.k0 = D.k0;
        const uint32_t* __restrict__ q = P.queue + D.qofs;
        cx_run run;
        run.v = D.vbase; run.t = D.tbase; run.c = D.cbase; run.b = 0;
        uint64_t* __restrict__ info = P.info64 + D.qofs;
        if (!D.near) {
            if (r0) cx_skip_rounds(P, G, q, r0, lane, s_ntri, run);
#ifdef CX_S3_STAMPS
            nrounds += r1 - r0;
            cx_emit_queue_fast(P, G, q, r0 * 64u, min(D.n, r1 * 64u), lane, run, reinterpret_cast<uint32_t*>(s_vstage[wave]), s_ntri, info, tacc);
#else
            if (TQ)        // the fractions come from the stream kernel (the pointer is the wave's region minus its first vertex: indexed by GLOBAL vertex)
                cx_emit_queue_t(P, G, q, r0 * 64u, min(D.n, r1 * 64u), lane, run, reinterpret_cast<uint32_t*>(s_vstage[wave]), s_ntri, info,
                                reinterpret_cast<const uint32_t*>(P.tq) + ((size_t)D.w * T.wcap) - __builtin_amdgcn_readfirstlane(P.wbase[D.w].v));
            else
                cx_emit_queue_fast(P, G, q, r0 * 64u, min(D.n, r1 * 64u), lane, run, reinterpret_cast<uint32_t*>(s_vstage[wave]), s_ntri, info);
#endif
        } else if (r0 == 0u) {
            cx_process_queue<true>(P, [&](uint32_t x) { return cx_entry_lin(P, G, q[x]); }, D.n, lane, true, run, reinterpret_cast<cx_vrec*>(s_vstage[wave]), info);
        }
        if (D.rbase + nrb >= hi || fn >= nbatches) break;
        f = fn;
        D = Dn;
    }
#ifdef CX_S3_STAMPS
    if (P.stamps && lane == 0) {   // per wave: start, end, time in the 4 parts of a round, rounds
        unsigned long long* st = P.stamps + (size_t)(blockIdx.x * 4u + wave) * 8u;
        st[0] = tstart; st[1] = __builtin_amdgcn_s_memrealtime();
        st[2] = tacc[0]; st[3] = tacc[1]; st[4] = tacc[2]; st[5] = tacc[3]; st[6] = nrounds;
    }
#endif
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
// the same round for a lattice coordinate that may be negative (an array with a rim of samples around the reference's
// grid has its origin at -1): CPython hashes a small int to itself, except hash(-1) == -2
__device__ __forceinline__ uint64_t py_round_signed(uint64_t acc, int32_t x) {
    const uint64_t lane = (x == -1) ? ~1ULL : (uint64_t)(int64_t)x;
    acc += lane * CX_PY_P2;
    acc = (acc << 31) | (acc >> 33);
    return acc * CX_PY_P1;
}
// the second half of py_round: acc already holds accumulator + lane * P2
__device__ __forceinline__ uint64_t py_round_tail(uint64_t acc) {
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
    table[idx] = py_round_signed(py_round_signed(CX_PY_P5, (int32_t)i + (int32_t)org0), (int32_t)j + (int32_t)org1);
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

// the same from the low 32 bits of the hashes (five probe steps fit); `unres`: still on one slot after five steps
__device__ __forceinline__ bool py_set2_swapped_lo(uint32_t h1, uint32_t h2, bool& unres) {
    const uint32_t s1 = h1 & 7u;
    uint32_t s2 = h2 & 7u;
    uint32_t perturb = h2;
    for (int guard = 0; guard < 5 && s2 == s1; guard++) {
        perturb >>= 5;
        s2 = (s2 * 5u + 1u + perturb) & 7u;
    }
    unres = (s2 == s1);
    return s2 < s1;
}

// The same decision from one byte per lattice point (cx_k_hash_bytes, built once per grid shape and origin): bits 0-2 = slot
// of hash((i,j,k)) in an 8-slot table, bits 3-5 = the first slot of its probe sequence that differs from it (what it takes
// when another element sits in its slot; the same slot again if 16 probes never leave it).
// py_set2_swapped(h1, h2) == (t2 != s1 ? t2 < s1 : alt2 < s1): the loop above stops at the first probe of h2 that differs
// from s1 == t2.  Second element iterates first?  (set built by inserting a then b)
__device__ __forceinline__ bool cx_set2_swapped(uint32_t ca, uint32_t cb) {
    const uint32_t sa = ca & 7u, tb = cb & 7u, ab = (cb >> 3) & 7u;
    return ((tb != sa) ? tb : ab) < sa;
}

// tet vertex m of tet t -> cube corner, as compile-time constants (same data as cx_d_tet_corners)
__device__ constexpr uint8_t CX_TC[6][4] = CX_TET_CORNERS_INIT;
#define CX_TC_MASK(t) ((1u << CX_TC[t][0]) | (1u << CX_TC[t][1]) | (1u << CX_TC[t][2]) | (1u << CX_TC[t][3]))

// =================================================================================================
// K2 triangles.  Phase 1, one lane per cell record: the (first vertex, crossing mask) pairs of the 7
// corners that can own an edge of the voxel go to LDS, and every triangle of the cell gets a slot word
// (cell lane, LUT entry, which of its 2 triangles, rank in the cell).  Phase 2, one lane per TRIANGLE:
// three (owner corner, direction) pairs from the LUT -> three vertex indices -> one 12-byte store,
// consecutive lanes writing consecutive triangles.  No divergent per-tetrahedron expansion.
// =================================================================================================
__device__ constexpr uint32_t CX_TRI_CD[6][16][2][2] = CX_TET_TRIS_CD_INIT;
#ifndef CX_VE_ROW
#define CX_VE_ROW 65
#endif
// what depends on the corner sign mask of a voxel alone, as tables (built at compile time from the tetrahedron list)
constexpr uint8_t CX_TCH[6][4] = CX_TET_CORNERS_INIT;
struct cx_pats_tab {
    uint32_t w[256][2];    // [sm]: x = 6 tetrahedron patterns x 4 bits (bit m: tet vertex m is low) | mask of the 2-2 tetrahedra << 24;
                           //       y = mask of the 1-3 / 3-1 tetrahedra | neighbour cells 1..6 that own a crossing edge of the voxel << 8
    uint16_t c22[6 * 16];  // [tet * 16 + pattern] of a 2-2 tetrahedron: cube corners of its low pair and of its high pair (tet
                           // vertex order = the reference's set insertion order), 3 bits each
};
constexpr cx_pats_tab cx_make_pats() {
    cx_pats_tab T{};
    for (uint32_t sm = 0; sm < 256u; sm++) {
        uint32_t pw = 0, m22 = 0, m1 = 0, want = 0;
        for (uint32_t t = 0; t < 6u; t++) {
            uint32_t pat = 0, np = 0;
            for (uint32_t m = 0; m < 4u; m++) {
                const uint32_t b = (sm >> CX_TCH[t][m]) & 1u;
                pat |= b << m;
                np += b;
            }
            pw |= pat << (4u * t);
            if (np == 2u) m22 |= 1u << t;
            if (np == 1u || np == 3u) m1 |= 1u << t;
        }
        for (uint32_t c = 1; c < 7u; c++) {
            // does corner c own a crossing edge of this voxel?  (a strict superset corner on the other side)
            const uint32_t sc = ((sm >> c) & 1u) ? 0xFFu : 0u;
            uint32_t sup = 0;
            for (uint32_t c2 = 0; c2 < 8u; c2++) sup |= ((c2 & c) == c && c2 != c) ? (1u << c2) : 0u;
            if (((sm ^ sc) & sup) != 0u) want |= 1u << c;
        }
        T.w[sm][0] = pw | (m22 << 24);
        T.w[sm][1] = m1 | (want << 8);
    }
    for (uint32_t t = 0; t < 6u; t++)
        for (uint32_t pat = 0; pat < 16u; pat++) {
            uint32_t lo[2] = {0, 0}, hi[2] = {0, 0}, nl = 0, nh = 0;
            for (uint32_t m = 0; m < 4u; m++) {
                if ((pat >> m) & 1u) { if (nl < 2u) lo[nl] = CX_TCH[t][m]; nl++; }
                else { if (nh < 2u) hi[nh] = CX_TCH[t][m]; nh++; }
            }
            T.c22[t * 16u + pat] = (uint16_t)((nl == 2u) ? (lo[0] | (lo[1] << 3) | (hi[0] << 6) | (hi[1] << 9)) : 0u);
        }
    return T;
}
__device__ constexpr cx_pats_tab CX_PATS = cx_make_pats();
union cx_tri_u {
    struct {
        uint16_t slot[12 * 64];  // one word per triangle
        uint32_t tfirst[64];     // first triangle index of each cell minus its rank in the wave
    } a;
    uint32_t hs[8][64];          // low words of the 8 corner hashes of each cell (dead before the slot words are written)
};
struct cx_tri_lds {
    static constexpr bool packed = false;
    uint2 ve[4][7][CX_VE_ROW];   // per wave: (first vertex, crossing mask) of corner c of each cell (rows padded: bank spread)
    cx_tri_u u[4];               // per wave
    uint16_t slot22[4][6 * 64];  // per wave: one word per 2-2 tetrahedron: cell lane | tet << 6 | pattern << 9
    uint32_t var[4][64];         // per wave: quad diagonal variants of each cell (bit t)
    uint32_t lut[6 * 16 * 2 * 2];
    uint2 pats[256];
    uint16_t c22[6 * 16];
};
// the same tables with ONE word per (corner, cell): first vertex relative to one of two bases (bits 0-22; bit 31 picks the base) |
// crossing mask << 23 -- the form the tile kernel keeps per queue entry in LDS (cx_tile3d.h): half the bytes, four workgroups per CU
struct cx_tri_lds_p {
    static constexpr bool packed = true;
    uint32_t ve[4][7][CX_VE_ROW];
    cx_tri_u u[4];
    uint16_t slot22[4][6 * 64];
    uint32_t var[4][64];
    uint32_t lut[6 * 16 * 2 * 2];
    uint2 pats[256];
    uint16_t c22[6 * 16];
    uint32_t vb[2];              // the two bases
};
template <typename LDS>
__device__ __forceinline__ void cx_tri_lds_init(LDS& L) {
    for (uint32_t x = threadIdx.x; x < 6 * 16 * 2 * 2; x += blockDim.x) L.lut[x] = (&CX_TRI_CD[0][0][0][0])[x];
    for (uint32_t x = threadIdx.x; x < 256u; x += blockDim.x) L.pats[x] = make_uint2(CX_PATS.w[x][0], CX_PATS.w[x][1]);
    for (uint32_t x = threadIdx.x; x < 96u; x += blockDim.x) L.c22[x] = CX_PATS.c22[x];
}
// slot word: bits 0-5 cell lane, 6 second triangle of the entry, 7-14 LUT entry (tet*32 + pattern*2 + variant)

// everything of one cell record that comes from memory.  The loads of record n+1 are issued before
// record n is expanded and are waited for BEFORE the triangle stores of record n are issued:
// `s_waitcnt vmcnt` retires loads and stores in issue order, so a load behind a store waits for the
// store's round trip as well.
struct cx_tri_in {
    uint4 rec;            // cell record (zero: no record)
    uint2 nb[6];          // table entries (first vertex, crossing mask) of corners 1..6
    uint64_t hxy[4];      // CPython hash prefixes of the 4 (i,j) columns of the voxel (CX_DIAG_CPYTHON310 without P.hbytes)
    uint32_t ck;          // k of the cell
    uint32_t hb[4];       // set-order codes of the 8 corners, (k, k+1) pairs of the 4 (i,j) columns (P.hbytes)
};
__device__ __forceinline__ uint32_t cx_need_hash(uint32_t sm, uint32_t tetskip, uint32_t ntri) {
    uint32_t need = 0;   // corners whose hash decides a quad diagonal
#pragma unroll
    for (int t = 0; t < 6; t++) {
        const uint32_t pat = ((sm >> CX_TC[t][0]) & 1u) | (((sm >> CX_TC[t][1]) & 1u) << 1) |
                             (((sm >> CX_TC[t][2]) & 1u) << 2) | (((sm >> CX_TC[t][3]) & 1u) << 3);
        if (ntri && !((tetskip >> t) & 1u) && __popc(pat) == 2) need |= CX_TC_MASK(t);
    }
    return need;
}
__device__ __forceinline__ void cx_tri_fetch(const cx_params& P, const uint64_t* __restrict__ hash_xy, const uint4& rec,
                                             cx_tri_in& I) {
    const uint32_t plane = P.n1 * P.n2;
    I.rec = rec;
    const uint32_t lin = rec.x, sm = rec.y & 0xFFu, tetskip = (rec.y >> 8) & 0x3Fu, ntri = (rec.y >> 16) & 0xFFu;
#pragma unroll
    for (uint32_t c = 1; c < 7; c++) {
        // does corner c own a crossing edge of this voxel?  (a strict superset corner on the other side)
        const uint32_t sc = ((sm >> c) & 1u) ? 0xFFu : 0u;
        uint32_t sup = 0;
#pragma unroll
        for (uint32_t c2 = 0; c2 < 8; c2++) sup |= ((c2 & c) == c && c2 != c) ? (1u << c2) : 0u;
        uint2 pr = make_uint2(0u, 0u);
        if (ntri && ((sm ^ sc) & sup) != 0u && !(P.flags & CX_DBG_NO_LOOKUP)) {
            const uint32_t lc = lin + ((c & 4u) ? plane : 0u) + ((c & 2u) ? P.n2 : 0u) + (c & 1u);
            const uint64_t e = P.celltab[lc];
            pr = make_uint2((uint32_t)e, (uint32_t)(e >> 32));
        }
        I.nb[c - 1u] = pr;
    }
    I.ck = 0;
    I.hxy[0] = I.hxy[1] = I.hxy[2] = I.hxy[3] = 0;
    I.hb[0] = I.hb[1] = I.hb[2] = I.hb[3] = 0;
    if (P.hbytes) {
        if (cx_need_hash(sm, tetskip, ntri) != 0u) {   // only voxels: all 8 corners are inside the array
            const uint8_t* __restrict__ hb = P.hbytes + lin;
            uint16_t t0, t1, t2, t3;
            __builtin_memcpy(&t0, hb, 2); __builtin_memcpy(&t1, hb + P.n2, 2);
            __builtin_memcpy(&t2, hb + plane, 2); __builtin_memcpy(&t3, hb + plane + P.n2, 2);
            I.hb[0] = t0; I.hb[1] = t1; I.hb[2] = t2; I.hb[3] = t3;
        }
    } else if ((P.flags & CX_DIAG_CPYTHON310) && cx_need_hash(sm, tetskip, ntri) != 0u) {
        const uint32_t ci = cx_div(lin, P.div_plane);
        const uint32_t r = lin - ci * plane;
        const uint32_t cj = cx_div(r, P.div_row);
        I.ck = r - cj * P.n2;
        // hash prefixes of the 4 (i,j) columns of this voxel (clamped: pseudo cells never need them)
        const uint32_t i1 = min(ci + 1u, P.n0 - 1u), j1 = min(cj + 1u, P.n1 - 1u);
        I.hxy[0] = hash_xy[ci * P.n1 + cj]; I.hxy[1] = hash_xy[ci * P.n1 + j1];
        I.hxy[2] = hash_xy[i1 * P.n1 + cj]; I.hxy[3] = hash_xy[i1 * P.n1 + j1];
    }
}
// the loads behind I (and the next record) have to be complete here
__device__ __forceinline__ void cx_tri_pin(cx_tri_in& I, uint4& nxt) {
    asm volatile("" : "+v"(I.rec.x), "+v"(I.rec.y), "+v"(I.rec.z), "+v"(I.rec.w), "+v"(nxt.x), "+v"(nxt.y), "+v"(nxt.z), "+v"(nxt.w) :: "memory");
    asm volatile("" : "+v"(I.nb[0].x), "+v"(I.nb[0].y), "+v"(I.nb[1].x), "+v"(I.nb[1].y), "+v"(I.nb[2].x), "+v"(I.nb[2].y) :: "memory");
    asm volatile("" : "+v"(I.nb[3].x), "+v"(I.nb[3].y), "+v"(I.nb[4].x), "+v"(I.nb[4].y), "+v"(I.nb[5].x), "+v"(I.nb[5].y) :: "memory");
    asm volatile("" : "+v"(I.hxy[0]), "+v"(I.hxy[1]), "+v"(I.hxy[2]), "+v"(I.hxy[3]), "+v"(I.ck) :: "memory");
    asm volatile("" : "+v"(I.hb[0]), "+v"(I.hb[1]), "+v"(I.hb[2]), "+v"(I.hb[3]) :: "memory");
}

// phase 1 of one record per lane: LDS tables and slot words; returns the wave's triangle count
// NEG_ORIGIN: the array starts at a negative lattice point (rim of extra samples): signed hash lanes.  A kernel of its
// own, because the extra path costs the common one ~9 % when it is a run-time branch (registers, code size)
//
// The triangle kernels are bound by VALU issue, so everything that depends on the corner sign mask
// alone comes from a table (CX_PATS: the 6 tetrahedron patterns, which tetrahedra split 2-2 / 1-3, which neighbour cells own an
// edge of the voxel), and the CPython set order of the quad diagonals is decided one lane per 2-2 TETRAHEDRON instead of six
// predicated tetrahedra per cell lane: a voxel has 1.7 of them on average, so two rounds of 64 lanes replace six unrolled
// hash comparisons per lane.  The corner hashes (low words) of a round's cells go through LDS (hs), overlaying the slot
// words, which are written afterwards.
template <bool NEG_ORIGIN, typename LDS>
__device__ __forceinline__ uint32_t cx_tri_phase1(const cx_params& P, LDS& L, uint32_t lane, uint32_t wave, const cx_tri_in& I) {
    const uint32_t sm = I.rec.y & 0xFFu, tetskip = (I.rec.y >> 8) & 0x3Fu, ntri = (I.rec.y >> 16) & 0xFFu;
    if constexpr (LDS::packed) {     // rec.w and nb[].x carry the packed words as they are
        L.ve[wave][0][lane] = I.rec.w;
#pragma unroll
        for (uint32_t c = 1; c < 7; c++) L.ve[wave][c][lane] = I.nb[c - 1u].x;
    } else {
        L.ve[wave][0][lane] = make_uint2(I.rec.w, I.rec.y >> 24);
#pragma unroll
        for (uint32_t c = 1; c < 7; c++) L.ve[wave][c][lane] = I.nb[c - 1u];
    }
    const uint2 pw = L.pats[sm];                                  // x: 6 patterns x 4 bits | 2-2 mask << 24; y: 1-3 mask | wanted neighbours << 8
    const uint32_t em6 = ntri ? (~tetskip & 0x3Fu) : 0u;          // tetrahedra that emit
    const uint32_t m22 = (pw.x >> 24) & em6, m1 = pw.y & em6;
    // quad diagonal variants of the 2-2 tetrahedra (bit t of `variants`)
    uint32_t variants = 0;
    if (P.hbytes) {
        // code of corner c = byte (c & 1) of column c >> 1
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const uint32_t pat = (pw.x >> (4 * t)) & 15u;
            if (!((m22 >> t) & 1u)) continue;
            uint32_t l0 = 0, l1 = 0, h0 = 0, h1 = 0;
            int nl = 0, nh = 0;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t cc = CX_TC[t][m];
                const uint32_t code = (I.hb[cc >> 1] >> (8u * (cc & 1u))) & 0xFFu;
                if ((pat >> m) & 1u) { if (nl == 0) l0 = code; else l1 = code; nl++; }
                else { if (nh == 0) h0 = code; else h1 = code; nh++; }
            }
            if (cx_set2_swapped(l0, l1) != cx_set2_swapped(h0, h1)) variants |= 1u << t;
        }
    } else if ((P.flags & CX_DIAG_CPYTHON310) && __ballot(m22 != 0u) != 0ULL) {
        // (k + origin) * P2 of the tuple hash's last round for k and k + 1 (as CPython sees the lattice coordinate: hash(-1) == -2)
        uint64_t kp0, kp1;
        if (!NEG_ORIGIN) {
            kp0 = (uint64_t)(I.ck + P.org2) * CX_PY_P2;
            kp1 = kp0 + CX_PY_P2;
        } else {
            const int32_t v0 = (int32_t)I.ck + (int32_t)P.org2, v1 = v0 + 1;
            kp0 = ((v0 == -1) ? ~1ULL : (uint64_t)(int64_t)v0) * CX_PY_P2;
            kp1 = ((v1 == -1) ? ~1ULL : (uint64_t)(int64_t)v1) * CX_PY_P2;
        }
        bool exact = (P.flags & CX_DBG_HASH64) != 0u;
        if (!exact) {
            // Fast form: only the LOW 32 bits of the corner hashes -- a probe step of the set order reads three bits five places
            // further up, so five steps fit.  A pair still on one slot after five steps, or a hash word of all ones (possibly
            // the hash CPython replaces by a constant), sends the whole wave through the 64-bit form below.
            bool bad = false;
            uint32_t (*hs)[64] = L.u[wave].hs;
#pragma unroll
            for (uint32_t c = 0; c < 8; c++) {
                const uint64_t sacc = I.hxy[c >> 1] + ((c & 1u) ? kp1 : kp0);
                const uint32_t rot = __builtin_amdgcn_alignbit((uint32_t)sacc, (uint32_t)(sacc >> 32), 1);   // low word of rotl64(s, 31)
                const uint32_t h = rot * (uint32_t)CX_PY_P1 + (uint32_t)(3ULL ^ (CX_PY_P5 ^ 3527539ULL));
                bad = bad || (h == 0xFFFFFFFFu);
                hs[c][lane] = h;
            }
            bad = bad && m22 != 0u;
            L.var[wave][lane] = 0u;
            uint32_t tot22;
            const uint32_t pre22 = cx_wave_prefix_small<3>(__popc(m22), tot22);
#pragma unroll
            for (uint32_t t = 0; t < 6; t++)
                if ((m22 >> t) & 1u)
                    L.slot22[wave][pre22 + __popc(m22 & ((1u << t) - 1u))] = (uint16_t)(lane | (t << 6) | (((pw.x >> (4u * t)) & 15u) << 9));
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j0 = 0; j0 < tot22; j0 += 64u) {   // wave-uniform: one lane per 2-2 tetrahedron
                const uint32_t j = j0 + lane;
                const bool ok = j < tot22;
                const uint32_t w = L.slot22[wave][ok ? j : 0u];
                const uint32_t cell = w & 63u, t = (w >> 6) & 7u;
                const uint32_t cc = L.c22[(t << 4) | (w >> 9)];   // corners of the low pair and of the high pair, 3 bits each
                bool u1, u2;
                const bool a = py_set2_swapped_lo(hs[cc & 7u][cell], hs[(cc >> 3) & 7u][cell], u1);
                const bool b = py_set2_swapped_lo(hs[(cc >> 6) & 7u][cell], hs[(cc >> 9) & 7u][cell], u2);
                if (ok && a != b) atomicOr(&L.var[wave][cell], 1u << t);
                bad = bad || (ok && (u1 || u2));
            }
            __builtin_amdgcn_wave_barrier();
            variants = L.var[wave][lane];
            exact = __ballot(bad) != 0ULL;
        }
        if (exact) {
            variants = 0;
            uint64_t h[8];
#pragma unroll
            for (uint32_t c = 0; c < 8; c++) h[c] = py_finish3(py_round_tail(I.hxy[c >> 1] + ((c & 1u) ? kp1 : kp0)));
#pragma unroll
            for (int t = 0; t < 6; t++) {
                if (!((m22 >> t) & 1u)) continue;
                const uint32_t pat = (pw.x >> (4 * t)) & 15u;
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
        __builtin_amdgcn_wave_barrier();   // hs is dead: the slot words below overlay it
    }
    // slot words of this cell's triangles, in tetrahedron order
    uint32_t ttot;
    const uint32_t tpre = cx_wave_prefix_small<4>(ntri, ttot);
    L.u[wave].a.tfirst[lane] = I.rec.z - tpre;   // triangle j of the wave goes to tfirst[cell] + j
    uint32_t pos = tpre;
    uint16_t* slot = L.u[wave].a.slot;
#pragma unroll
    for (uint32_t t = 0; t < 6; t++) {
        const uint32_t two = (m22 >> t) & 1u, one = (m1 >> t) & 1u;
        const uint32_t word = lane | (t << 12) | (((pw.x >> (4u * t)) & 15u) << 8) | (((variants >> t) & 1u) << 7);
        if (two | one) slot[pos] = (uint16_t)word;
        if (two) slot[pos + 1u] = (uint16_t)(word | (1u << 6));
        pos += 2u * two + one;
    }
    __builtin_amdgcn_wave_barrier();
    return ttot;
}
// phase 2, one lane per triangle: stores only
template <typename LDS>
__device__ __forceinline__ void cx_tri_phase2(const cx_params& P, LDS& L, uint32_t lane, uint32_t wave, uint32_t ttot) {
    for (uint32_t j0 = 0; j0 < ttot; j0 += 64u) {   // wave-uniform
        const uint32_t j = j0 + lane;
        const bool ok = j < ttot;
        const uint32_t w = L.u[wave].a.slot[ok ? j : 0u];
        const uint32_t cell = w & 63u;
        const uint32_t tw = L.lut[((w >> 7) & 0xFFu) * 2u + ((w >> 6) & 1u)];
        int32_t vi[3];
#pragma unroll
        for (uint32_t sidx = 0; sidx < 3; sidx++) {
            const uint32_t cd = (tw >> (6u * sidx)) & 63u;
            if constexpr (LDS::packed) {
                const uint32_t wd = L.ve[wave][cd & 7u][cell];
                vi[sidx] = (int32_t)(L.vb[wd >> 31] + (wd & 0x7FFFFFu) + __popc(((wd >> 23) & 0xFEu) & ((1u << (cd >> 3)) - 1u)));
            } else {
                const uint2 pr = L.ve[wave][cd & 7u][cell];
                vi[sidx] = (int32_t)(pr.x + __popc(pr.y & ((1u << (cd >> 3)) - 1u)));
            }
        }
        if (ok && !(P.flags & CX_DBG_NO_TRIS)) {
            // 32-bit wrap-around on purpose: tfirst = first - rank may be "negative" when the wave's cells come from
            // different reservations (generic path); the sum is the triangle index again
            int32_t* out = P.tris + (size_t)(uint32_t)(L.u[wave].a.tfirst[cell] + j) * 3u;
#if CX_NT_TRIS
            typedef int32_t cx_v3i __attribute__((ext_vector_type(3)));
            __builtin_nontemporal_store(cx_v3i{vi[0], vi[1], vi[2]}, reinterpret_cast<cx_v3i*>(out));
#else
            out[0] = vi[0]; out[1] = vi[1]; out[2] = vi[2];
#endif
        }
    }
    __builtin_amdgcn_wave_barrier();
}

#ifndef CX_NT_RECS
#define CX_NT_RECS 1     // the cell records are read once: nontemporal loads (A/B: ~1.5 %, within noise)
#endif
__device__ __forceinline__ uint4 cx_load_record(const uint4* p) {
#if CX_NT_RECS
    typedef uint32_t cx_v4u __attribute__((ext_vector_type(4)));
    const cx_v4u v = __builtin_nontemporal_load(reinterpret_cast<const cx_v4u*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
// one lane per cell record; waves walk the record array grid-stride (the record count lives on the device)
#ifndef CX_K2_MIN_WAVES
#define CX_K2_MIN_WAVES 1
#endif
template <bool NEG_ORIGIN>
__global__ __launch_bounds__(256, CX_K2_MIN_WAVES) void cx_k_emit_triangles(const cx_params P, const uint64_t* __restrict__ hash_xy) {
    __shared__ cx_tri_lds L;
    const uint32_t ncells = min(P.counters[CX_CNT_CELLS], P.ccap);
    if (P.counters[CX_CNT_TRIS] > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap) return;  // host re-runs with more room
    // (Records are taken in launch order.  Measured and dropped, third session of round 4: every XCD taking ONE contiguous eighth of each
    // stride -- workgroup b -> block (b % 8) * (grid / 8) + b / 8 --, so that the queue / info words of neighbour cells come from the XCD's
    // own L2 more often: 0.126 -> 0.136 ms, 0.330 -> 0.340-0.345 ms per step.)
    if (blockIdx.x * blockDim.x >= ncells) return;                 // whole block idle
    cx_tri_lds_init(L);
    __syncthreads();
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    cx_tri_in Ia, Ib;
    cx_tri_fetch(P, hash_xy, (idx < ncells) ? cx_load_record(P.cells + idx) : zero, Ia);
    uint4 rec_b = (idx + stride < ncells) ? cx_load_record(P.cells + idx + stride) : zero;
    cx_tri_pin(Ia, rec_b);          // nothing is in flight when the loop is entered
    while (idx - lane < ncells) {   // wave-uniform
        const uint32_t nidx = idx + stride;
        cx_tri_fetch(P, hash_xy, rec_b, Ib);                                    // loads of the next record ...
        uint4 rec_c = (nidx + stride < ncells) ? cx_load_record(P.cells + nidx + stride) : zero;   // ... and the record after it
        const uint32_t ttot = cx_tri_phase1<NEG_ORIGIN>(P, L, lane, wave, Ia);
        cx_tri_pin(Ib, rec_c);                                                  // ... are back before the stores go out
        cx_tri_phase2(P, L, lane, wave, ttot);
        Ia = Ib;
        rec_b = rec_c;
        idx = nidx;
    }
}

// =================================================================================================
// K2 for the staged pipeline: the same triangle stage, but the (first vertex, crossing mask) pairs of the neighbour cells come
// from the vertex stage's word per QUEUE ENTRY (P.info64) instead of a table of one entry per sample.  The queue entry of
// any active lattice cell Y is found by arithmetic: the stream kernel left, per plane step and streaming lane, the queue
// position of the lane's first active cell and the 16-bit set of its active cells (P.qa), so
//     entry(Y) = wave(Y) * wcap + (qa >> 16) + popc(qa & cells below Y).
// Both tables are dense where they are used: a round of 64 records touches a few cache lines per lookup instead of one per
// cell and neighbour (the vector L1 handles about one line per 3-4 cycles per CU whatever the bytes asked for: these kernels
// are bound by the NUMBER of lines their gathers and scatters touch, not by bytes -- DESIGN.md section 4).
// Three records are in flight per lane: record s+2 has its queue words requested, record s+1 its info words, record s is
// expanded and stored.
// =================================================================================================
static_assert(CX_RJ == 4, "the wave / row arithmetic below assumes 4 rows per streaming wave");
struct cx_tri_qa {          // a record between its first stage (queue words requested) and its second
    uint4 rec;
    uint32_t qa[6];         // per neighbour cell X + c, c = 1..6: the queue word of its streaming lane and plane step, as loaded
    uint32_t geo;           // bits 0-5: neighbour c is wanted (bit c-1); 6: X + dk is in the next k block; 7: X + dj in the next wave;
                            // 8: X + di in the next chunk of planes; 9-10: row of X in its wave; 11-12: sample of X in its lane
    uint32_t w;             // streaming wave of the cell
    uint32_t ij, ck;        // (i * n1 + j) and k of the cell's lattice point
};
// byte-offset loads with a 32-bit offset from a uniform base (global_load with an SGPR base and a VGPR offset: no 64-bit
// address arithmetic per lane); the tables they read are smaller than 4 GiB
__device__ __forceinline__ uint32_t cx_ld_u32_at(const uint32_t* base, uint32_t byte_off) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + byte_off);
}
// a cell at lattice point (ci, cj, ck) = plane step ps of streaming wave w (row rr of its 4, lane ls, sample mm): request the queue
// words of the neighbour cells that own an edge of its voxel
__device__ __forceinline__ void cx_triq_issue(const cx_params& P, const cx_task& T, const cx_tri_lds& L, const uint4& rec, uint32_t w, uint32_t ci,
                                              uint32_t cj, uint32_t ck, uint32_t ps, uint32_t nsteps, cx_tri_qa& A) {
    A.rec = rec;
    const uint32_t sm = rec.y & 0xFFu, ntri = (rec.y >> 16) & 0xFFu;
    const uint32_t rr = cj & 3u, ls = (ck >> 2) & 63u, mm = ck & 3u;
    const uint32_t wv = w & 3u;
    A.w = w;
    const bool m3 = (mm == 3u), r3 = (rr == 3u), pl = (ps + 1u == nsteps);
    const bool kfl = m3 && ls == 63u;
    // neighbour cells that own a crossing edge of this voxel (bit c), from the table
    const uint32_t want = (ntri && !(P.flags & CX_DBG_NO_LOOKUP)) ? ((L.pats[sm].y >> 9) & 0x3Fu) : 0u;
    A.geo = want | (kfl ? 64u : 0u) | (r3 ? 128u : 0u) | (pl ? 256u : 0u) | (rr << 9) | (mm << 11);
    A.ij = ci * P.n1 + cj; A.ck = ck;
    // word index of the queue word of X + c in P.qa = [wave][plane step][lane]: a sum of one term per axis.  The neighbour
    // is in the same lane word unless X is in the last sample column (next lane, or lane 0 of the next k block), the last
    // row (next wave, or the first wave of the next block in j) or the last plane step (the next chunk of planes)
    constexpr uint32_t WQ = CX_SWP * 64u;
    const uint32_t bk1 = kfl ? 4u * WQ : (m3 ? ls + 1u : ls);
    const uint32_t bj1 = r3 ? ((wv == 3u) ? (4u * T.nks - 3u) * WQ : WQ) : 0u;
    const uint32_t bp0 = ps << 6, bp1 = pl ? 4u * T.nks * T.njg * WQ : ((ps + 1u) << 6);
    const uint32_t base = A.w * WQ;
    const uint32_t B0 = base + bp0, B1 = base + bp1, B0j = B0 + bj1, B1j = B1 + bj1;
    const uint32_t idx[6] = {B0 + bk1, B0j + ls, B0j + bk1, B1 + ls, B1 + bk1, B1j + ls};
#pragma unroll
    for (uint32_t c = 0; c < 6; c++) {
        A.qa[c] = 0;
#ifdef CX_ABL_QA     // timing experiment: no queue-word gathers
        if ((want >> c) & 1u) A.qa[c] = (idx[c] * 977u) & 0x00FFFFFFu;
#else
        if ((want >> c) & 1u) A.qa[c] = cx_ld_u32_at(P.qa, idx[c] << 2);
#endif
    }
}
// ... of a cell record (the record-walking kernel): the lattice point from its linear index
__device__ __forceinline__ void cx_triq_stage1(const cx_params& P, const cx_task& T, const cx_tri_lds& L, const uint4& rec,
                                               cx_tri_qa& A) {
    const uint32_t plane = P.n1 * P.n2;
    const uint32_t lin = rec.x;
    const uint32_t ci = cx_div(lin, P.div_plane);
    const uint32_t rem = lin - ci * plane;
    const uint32_t cj = cx_div(rem, P.div_row);
    const uint32_t ck = rem - cj * P.n2;
    // where the cell sits in the streaming layout
    const uint32_t ic = cx_div(ci, P.div_ci);
    const uint32_t ps = ci - ic * T.ci;
    const uint32_t nsteps = min(T.ci, P.n0 - ic * T.ci);
    const uint32_t w = 4u * ((ck >> 8) + T.nks * ((cj >> 4) + T.njg * ic)) + ((cj >> 2) & 3u);
    cx_triq_issue(P, T, L, rec, w, ci, cj, ck, ps, nsteps, A);
}
// ... of a queue entry e of streaming wave w with its info word iw (the entry-walking kernel): the tile from the wave's number,
// the position in the tile from the entry; tfirst = number of the cell's first triangle
__device__ __forceinline__ void cx_trie_stage1(const cx_params& P, const cx_task& T, const cx_tri_lds& L, uint32_t e, uint64_t iw, uint32_t w,
                                               uint32_t tfirst, cx_tri_qa& A) {
    const uint32_t b = w >> 2;
    const uint32_t t1 = cx_div(b, T.div_nks);
    const uint32_t ks = b - t1 * T.nks;
    const uint32_t ic = cx_div(t1, T.div_njg);
    const uint32_t jg = t1 - ic * T.njg;
    const uint32_t p = ic * T.ci;
    const uint32_t bit = (e >> 10) & 31u;
    const uint32_t rr = (bit * 11u) >> 6;
    const uint32_t ps = (e >> 21) & 127u;
    const uint32_t ci = p + ps, cj = jg * (4u * CX_RJ) + (w & 3u) * CX_RJ + rr, ck = ks * 256u + 4u * ((e >> 15) & 63u) + (bit - 6u * rr);
    const uint32_t hi = (uint32_t)(iw >> 32);
    // the cell's record as the later stages read it: {-, signs | tetskip << 8 | triangles << 16 | crossing mask << 24, first triangle, first vertex}
    const uint4 rec = make_uint4(0u, cx_entry_signs(e) | (((hi >> 12) & 0x3Fu) << 8) | (((hi >> 8) & 0xFu) << 16) | ((hi & 0xFFu) << 24), tfirst, (uint32_t)iw);
    cx_triq_issue(P, T, L, rec, w, ci, cj, ck, ps, min(T.ci, P.n0 - p), A);
}
__device__ __forceinline__ void cx_triq_pin1(cx_tri_qa& A, uint4& nxt) {
    asm volatile("" : "+v"(nxt.x), "+v"(nxt.y), "+v"(nxt.z), "+v"(nxt.w) :: "memory");
    asm volatile("" : "+v"(A.qa[0]), "+v"(A.qa[1]), "+v"(A.qa[2]), "+v"(A.qa[3]), "+v"(A.qa[4]), "+v"(A.qa[5]) :: "memory");
}
__device__ __forceinline__ void cx_triq_stage2(const cx_params& P, const cx_task& T, const uint64_t* __restrict__ hash_xy, const cx_tri_qa& A,
                                               cx_tri_in& I) {
    I.rec = A.rec;
    // queue entry of X + c = first entry of its wave + (queue position of its lane's first cell) + active cells of the lane below it
    const bool kfl = (A.geo >> 6) & 1u, r3 = (A.geo >> 7) & 1u, pl = (A.geo >> 8) & 1u;
    const uint32_t rr = (A.geo >> 9) & 3u, mm = (A.geo >> 11) & 3u;
    const uint32_t qk1 = kfl ? 4u * T.wcap : 0u;
    const uint32_t qj1 = r3 ? (((A.w & 3u) == 3u) ? (4u * T.nks - 3u) * T.wcap : T.wcap) : 0u;
    const uint32_t qp1 = pl ? 4u * T.nks * T.njg * T.wcap : 0u;
    const uint32_t Q0 = A.w * T.wcap, Q1 = Q0 + qp1, Q0j = Q0 + qj1, Q1j = Q1 + qj1;
    const uint32_t qb[6] = {Q0 + qk1, Q0j, Q0j + qk1, Q1, Q1 + qk1, Q1j};
    // bit of X + c in its lane's 16-bit set of active cells (bit 4r + m)
    const uint32_t b00 = 4u * rr + mm, b01 = 4u * rr + ((mm + 1u) & 3u), b10 = 4u * ((rr + 1u) & 3u) + mm, b11 = 4u * ((rr + 1u) & 3u) + ((mm + 1u) & 3u);
    const uint32_t bit[6] = {b01, b10, b11, b00, b01, b10};
#pragma unroll
    for (uint32_t c = 0; c < 6; c++) {
        uint2 pr = make_uint2(0u, 0u);
        if ((A.geo >> c) & 1u) {
            const uint32_t qa = A.qa[c];
            const uint32_t below = qa & ((1u << bit[c]) - 1u);     // bit[c] <= 15: only bits of the 16-bit set
            const uint32_t at = qb[c] + (qa >> 16) + __popc(below);
#ifdef CX_ABL_INFO   // timing experiment: no info-word gathers
            const uint64_t e = ((uint64_t)0xFE << 32) | (uint64_t)at;
#else
            const uint64_t e = P.info64[at];
#endif
            pr = make_uint2((uint32_t)e, (uint32_t)(e >> 32));   // (bits above the crossing mask ride along: phase 2 only looks below bit 8.  No arithmetic on the loaded word here -- it would put a wait behind every gather)
        }
        I.nb[c] = pr;
    }
    I.ck = A.ck;
    I.hb[0] = I.hb[1] = I.hb[2] = I.hb[3] = 0;
    I.hxy[0] = I.hxy[1] = I.hxy[2] = I.hxy[3] = 0;
    const uint32_t sm = A.rec.y & 0xFFu, tetskip = (A.rec.y >> 8) & 0x3Fu, ntri = (A.rec.y >> 16) & 0xFFu;
    if ((P.flags & CX_DIAG_CPYTHON310) && cx_need_hash(sm, tetskip, ntri) != 0u) {
        // only voxels get here (cj + 1 < n1, ci + 1 < n0): the prefixes of columns (i, j), (i, j+1) are neighbours in the
        // table, ONE 16-byte load per plane (8-byte aligned)
        struct __attribute__((packed, aligned(8))) u64x2 { uint64_t a, b; };
        const u64x2 p0 = *reinterpret_cast<const u64x2*>(hash_xy + A.ij);
        const u64x2 p1 = *reinterpret_cast<const u64x2*>(hash_xy + (A.ij + P.n1));
        I.hxy[0] = p0.a; I.hxy[1] = p0.b; I.hxy[2] = p1.a; I.hxy[3] = p1.b;
    }
}
__device__ __forceinline__ void cx_triq_pin2(cx_tri_in& I) {
    asm volatile("" : "+v"(I.hxy[0]), "+v"(I.hxy[1]), "+v"(I.hxy[2]), "+v"(I.hxy[3]) :: "memory");
    asm volatile("" : "+v"(I.nb[0].x), "+v"(I.nb[0].y), "+v"(I.nb[1].x), "+v"(I.nb[1].y), "+v"(I.nb[2].x), "+v"(I.nb[2].y) :: "memory");
    asm volatile("" : "+v"(I.nb[3].x), "+v"(I.nb[3].y), "+v"(I.nb[4].x), "+v"(I.nb[4].y), "+v"(I.nb[5].x), "+v"(I.nb[5].y) :: "memory");
}
#ifndef CX_K2Q_MIN_WAVES
#define CX_K2Q_MIN_WAVES 5   // 96 registers: one more wave per SIMD than the allocator takes on its own (98)
#endif
template <bool NEG_ORIGIN>
__global__ __launch_bounds__(256, CX_K2Q_MIN_WAVES) void cx_k_emit_triangles_q(const cx_params P, const cx_task T, const uint64_t* __restrict__ hash_xy) {
#ifdef CX_OCC_PAD   // experiment: fewer workgroups per CU (what does the time do with the occupancy?)
    __shared__ uint32_t s_pad[CX_OCC_PAD / 4];
    if (P.n0 == 0xFFFFFFFFu) reinterpret_cast<volatile uint32_t*>(s_pad)[threadIdx.x] = 1u;
#endif
    __shared__ cx_tri_lds L;
    const uint32_t ncells = min(P.counters[CX_CNT_CELLS], P.ccap);
    if (P.counters[CX_CNT_TRIS] > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap) return;  // host re-runs with more room
    // (Records are taken in launch order.  Measured and dropped, third session of round 4: every XCD taking ONE contiguous eighth of each
    // stride -- workgroup b -> block (b % 8) * (grid / 8) + b / 8 --, so that the queue / info words of neighbour cells come from the XCD's
    // own L2 more often: 0.126 -> 0.136 ms, 0.330 -> 0.340-0.345 ms per step.)
    if (blockIdx.x * blockDim.x >= ncells) return;                 // whole block idle
    cx_tri_lds_init(L);
    __syncthreads();
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    // A record is requested three steps before it is expanded -- UNCONDITIONALLY (from a clamped index) and without touching the
    // result: written as `(at < ncells) ? load : zero`, the load sat in a branch whose result the compiler copied into other
    // registers right behind it, i.e. every iteration began by waiting for the record it had just asked for.  Whether the record
    // exists is decided when it is used (`valid`).
    auto record = [&](uint32_t at) { return cx_load_record(P.cells + min(at, ncells - 1u)); };
    auto valid = [&](const uint4& r, uint32_t at) { return (at < ncells) ? r : zero; };
    // prologue: record idx through both stages, record idx + stride through the first, record idx + 2 stride loaded
    cx_tri_qa Ab, Ac;
    cx_tri_in Ia, Ib;
    uint4 rec_c = record(idx + 2u * stride);
    {
        uint4 r0 = record(idx), r1 = record(idx + stride);
        asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w) :: "memory");
        cx_triq_stage1(P, T, L, valid(r0, idx), Ac);
        cx_triq_stage1(P, T, L, valid(r1, idx + stride), Ab);
        cx_triq_pin1(Ac, rec_c);
        cx_triq_pin1(Ab, rec_c);
        cx_triq_stage2(P, T, hash_xy, Ac, Ia);
        cx_triq_pin2(Ia);
    }
    while (idx - lane < ncells) {   // wave-uniform
        const uint32_t nidx = idx + stride;
        uint4 rec_d = record(nidx + 2u * stride);          // the record three steps ahead
        cx_triq_stage1(P, T, L, valid(rec_c, nidx + stride), Ac);   // queue words of the record two steps ahead
        cx_triq_stage2(P, T, hash_xy, Ab, Ib);              // info words (and hash prefixes) of the next record
        const uint32_t ttot = cx_tri_phase1<NEG_ORIGIN>(P, L, lane, wave, Ia);
#if CX_PIN_BEFORE_STORES
        cx_triq_pin1(Ac, rec_d);                            // ... all back before the stores go out
        cx_triq_pin2(Ib);
        cx_tri_phase2(P, L, lane, wave, ttot);
#else
        cx_tri_phase2(P, L, lane, wave, ttot);
        cx_triq_pin1(Ac, rec_d);                            // ... waited for after the stores went out
        cx_triq_pin2(Ib);
#endif
        Ia = Ib;
        Ab = Ac;
        rec_c = rec_d;
        idx = nidx;
    }
}

// =================================================================================================
// K2 walking QUEUE ENTRIES (cx_k_emit_triangles_e, the default of the staged pipeline from round 3 on).  The vertex stage no longer
// writes a 16-byte record per cell for this kernel to read back (128 MB of round trip at 512^3): everything a record held is in
// the cell's queue entry (signs, position in its streaming wave's tile) and in the word the vertex stage leaves per entry (first
// vertex, crossing mask, triangles, tolerance skips), and the number of a cell's first triangle follows from a running count --
// triangles are numbered along the flat batch list, cell after cell.
// Work division as in the vertex stage: wave m takes rounds [m q, m q + q) of the flat batch list (P.rstart: the batch its share
// starts in) and, where that is inside a batch, counts the triangles of the rounds before (info words; a batch on the tolerance
// path is not divided).  Unlike the vertex stage it PACKS its cells: a round of 64 lanes is filled across batch boundaries
// (up to four pieces), so only the last round of a wave's share is partial -- this kernel is bound by VALU issue per round.
// Three rounds are in flight per wave, as in the record-walking kernel: round s+3 has its entries and info words requested,
// round s+2 its queue words, round s+1 the neighbours' info words (and hash prefixes), round s is expanded and stored.
// =================================================================================================
// a descriptor of the flat batch list, its fields made wave-uniform for the compiler (the address is)
__device__ __forceinline__ cx_bdesc cx_desc_uniform(const cx_bdesc& D) {
    cx_bdesc U;
    U.w = __builtin_amdgcn_readfirstlane(D.w); U.qofs = __builtin_amdgcn_readfirstlane(D.qofs); U.n = __builtin_amdgcn_readfirstlane(D.n);
    U.vbase = 0; U.cbase = 0; U.p = 0; U.j0 = 0; U.k0 = 0; U.nsteps = 0;      // (not used by the triangle stage)
    U.tbase = __builtin_amdgcn_readfirstlane(D.tbase);
    U.near = __builtin_amdgcn_readfirstlane(D.near); U.rbase = __builtin_amdgcn_readfirstlane(D.rbase);
    return U;
}
struct cx_esrc {              // wave-uniform: where the next round's cells come from
    uint32_t f, nbatches, lo, hi; // current batch, batches in the list, the wave's share (rounds)
    uint32_t at, bend;        // entries [at, bend) of the current batch are still to be handed out
    uint32_t qofs, w;         // of the current batch
    uint32_t bend_round;      // round behind the current batch
    uint32_t tbase;           // first triangle of the current batch
    cx_bdesc next;            // descriptor f + 1, requested when batch f was entered (as loaded: made uniform when it is entered)
    bool done;
};
// entries [r0 * 64, end) of batch D belong to the wave whose share of rounds is [lo, hi): a batch on the tolerance path goes as a
// whole to the wave in whose share its first round falls.  Requests the descriptor after it.
__device__ __forceinline__ void cx_esrc_enter(const cx_params& P, cx_esrc& S, const cx_bdesc& Draw) {
    const cx_bdesc D = cx_desc_uniform(Draw);
    const uint32_t nrb = (D.n + 63u) >> 6;
    const uint32_t r0 = max(S.lo, D.rbase) - D.rbase, r1 = min(S.hi, D.rbase + nrb) - D.rbase;
    S.qofs = D.qofs; S.w = D.w; S.bend_round = D.rbase + nrb; S.tbase = D.tbase;
    if (D.near) { S.at = 0u; S.bend = (r0 == 0u) ? D.n : 0u; }
    else { S.at = r0 * 64u; S.bend = min(D.n, r1 * 64u); }
    if (S.f + 1u < S.nbatches) S.next = P.flat[S.f + 1u];
}
// the batch in hand is used up (or was not ours): on to the next one of the share, if any
__device__ __forceinline__ void cx_esrc_advance(const cx_params& P, cx_esrc& S) {
    while (S.at >= S.bend && !S.done) {
        if (S.bend_round >= S.hi || S.f + 1u >= S.nbatches) { S.done = true; break; }
        S.f++;
        cx_esrc_enter(P, S, S.next);
    }
}
struct cx_eraw {              // per lane: a cell of a round as it comes from memory
    uint32_t e, w;            // queue entry, streaming wave (0xFFFFFFFF: no cell in this lane)
    uint64_t iw;              // the vertex stage's word
};
// hand out the next (up to) 64 cells and request their entries and info words.  Four rounds in five come whole out of the batch
// in hand; the others take its tail and the head of the next batch (a next batch shorter than the gap leaves lanes empty).
__device__ __forceinline__ void cx_esrc_fetch(const cx_params& P, cx_esrc& S, uint32_t lane, cx_eraw& R) {
    R.w = 0xFFFFFFFFu; R.e = 0; R.iw = 0;
    if (S.done) return;                            // wave-uniform
    uint32_t at, w;
    bool have;
    if (S.at + 64u <= S.bend) {                    // wave-uniform
        at = S.qofs + S.at + lane; w = S.w; have = true;
        S.at += 64u;
    } else {
        const uint32_t c0 = S.bend - S.at, pos0 = S.qofs + S.at, w0 = S.w;
        S.at = S.bend;
        cx_esrc_advance(P, S);
        uint32_t c1 = 0, pos1 = 0, w1 = 0;
        if (!S.done) {
            c1 = min(S.bend - S.at, 64u - c0); pos1 = S.qofs + S.at; w1 = S.w;
            S.at += c1;
        }
        have = lane < c0 + c1;
        at = (lane < c0) ? pos0 + lane : pos1 + (lane - c0);
        w = (lane < c0) ? w0 : w1;
    }
    if (have) { R.w = w; R.e = P.queue[at]; R.iw = P.info64[at]; }
}
__device__ __forceinline__ void cx_eraw_pin(cx_eraw& R) { asm volatile("" : "+v"(R.e), "+v"(R.iw) :: "memory"); }

#ifndef CX_K2E_MIN_WAVES
#define CX_K2E_MIN_WAVES 4
#endif
template <bool NEG_ORIGIN>
__global__ __launch_bounds__(256, CX_K2E_MIN_WAVES) void cx_k_emit_triangles_e(const cx_params P, const cx_task T, const uint64_t* __restrict__ hash_xy) {
    __shared__ cx_tri_lds L;
    if (P.counters[CX_CNT_TRIS] > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap) return;  // host re-runs with more room
    const uint32_t nbatches = min(P.counters[CX_CNT_BATCHES], P.fcap);
    const uint32_t rounds = P.counters[CX_CNT_ROUNDS];
    cx_tri_lds_init(L);
    __syncthreads();
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nkw = gridDim.x * 4u;     // == P.nkw
    const uint32_t share = max((rounds + nkw - 1u) / nkw, CX_S3_MIN_SHARE);
    const uint32_t lo = (blockIdx.x * 4u + wave) * share;
    if (lo >= rounds || nbatches == 0u) return;
    cx_esrc S;
    S.lo = lo; S.hi = min(rounds, lo + share); S.nbatches = nbatches; S.done = false;
    S.f = __builtin_amdgcn_readfirstlane(P.kstart[blockIdx.x * 4u + wave]);
    if (S.f >= nbatches) return;
    uint32_t run_t;      // number of the first triangle of the next round's first cell (wave-uniform)
    {
        const cx_bdesc D = P.flat[S.f];
        cx_esrc_enter(P, S, D);
        run_t = S.tbase;
        if (S.at) {   // the share starts inside this batch (never one on the tolerance path): the triangles of the cells before
            const uint64_t* __restrict__ iw = P.info64 + S.qofs;
            uint32_t t = 0;
            for (uint32_t x = lane; x < S.at; x += 64u) t += (uint32_t)(iw[x] >> 40) & 0xFu;
            run_t += cx_wave_sum(t);
        }
        if (S.at >= S.bend) {
            // the first batch of the share is on the tolerance path and belongs to the wave before: numbering is contiguous along
            // the list, so the first batch actually taken sets the count (it is taken from its first cell)
            cx_esrc_advance(P, S);
            if (S.done) return;
            run_t = S.tbase;
        }
    }
    // a round between its request and its first stage
    auto first_triangles = [&](const cx_eraw& R, uint32_t& tfirst) {
        // triangles are numbered cell after cell along the flat batch list
        const uint32_t nt = (R.w != 0xFFFFFFFFu) ? ((uint32_t)(R.iw >> 40) & 0xFu) : 0u;
        uint32_t ttot;
        const uint32_t tpre = cx_wave_prefix_small<4>(nt, ttot);
        tfirst = run_t + tpre;
        run_t += ttot;
    };
    cx_tri_qa Ab, Ac;
    cx_tri_in Ia, Ib;
    cx_eraw Rc, Rd;
    uint4 dummy = make_uint4(0, 0, 0, 0);
    auto stage1 = [&](const cx_eraw& R, cx_tri_qa& A) {
        uint32_t tfirst;
        first_triangles(R, tfirst);
        if (R.w != 0xFFFFFFFFu) cx_trie_stage1(P, T, L, R.e, R.iw, R.w, tfirst, A);
        else { A.rec = make_uint4(0, 0, 0, 0); A.geo = 0; A.w = 0; A.ij = 0; A.ck = 0; for (int c = 0; c < 6; c++) A.qa[c] = 0; }
    };
    // prologue: round 0 through both stages, round 1 through the first, round 2 requested
    {
        cx_eraw R0, R1;
        cx_esrc_fetch(P, S, lane, R0);
        cx_esrc_fetch(P, S, lane, R1);
        cx_esrc_fetch(P, S, lane, Rc);
        cx_eraw_pin(R0); cx_eraw_pin(R1);
        stage1(R0, Ac);
        stage1(R1, Ab);
        cx_triq_pin1(Ac, dummy);
        cx_triq_pin1(Ab, dummy);
        cx_triq_stage2(P, T, hash_xy, Ac, Ia);
        cx_triq_pin2(Ia);
        cx_eraw_pin(Rc);
    }
    // the loop ends when the round being expanded holds no cell: the source hands out full rounds until it is exhausted (only the
    // last one may be partial), then empty ones
    for (;;) {
        if (__ballot(Ia.rec.y != 0u) == 0ULL) break;         // (every queued cell has a sign change: its sign word is not 0)
        cx_esrc_fetch(P, S, lane, Rd);                   // entries and info words of the round three steps ahead
        stage1(Rc, Ac);                                      // queue words of the round two steps ahead
        cx_triq_stage2(P, T, hash_xy, Ab, Ib);               // info words (and hash prefixes) of the next round
        const uint32_t ttot = cx_tri_phase1<NEG_ORIGIN>(P, L, lane, wave, Ia);
#if CX_PIN_BEFORE_STORES
        cx_triq_pin1(Ac, dummy); cx_triq_pin2(Ib); cx_eraw_pin(Rd);
        cx_tri_phase2(P, L, lane, wave, ttot);
#else
        cx_tri_phase2(P, L, lane, wave, ttot);
        cx_triq_pin1(Ac, dummy); cx_triq_pin2(Ib); cx_eraw_pin(Rd);
#endif
        Ia = Ib;
        Ab = Ac;
        Rc = Rd;
    }
}

// =================================================================================================
// Fused emit: vertex records AND triangles of a batch in ONE pass over its queue entries.
//
// The staged kernels above hand vertex indices from the vertex stage to the triangle stage through memory: a sparse
// 8-byte table entry per vertex-owning cell (scattered into a table of one entry per SAMPLE) plus a 16-byte record per
// active cell, and the triangle stage gathers up to six table entries per cell.  Here the hand-over is dense: a small
// kernel (cx_k_cell_info) writes ONE u32 per queue entry -- first vertex of the cell relative to its streaming wave and
// its crossing mask -- and the queue entry of ANY active lattice cell Y is found by arithmetic: the stream kernel left,
// per plane step and streaming lane, the queue position of the lane's first active cell and the 16-bit set of its active
// cells (P.qa), so position(Y) = qa >> 16 + popc(active cells below Y).  The six neighbour cells of a voxel that can own
// one of its edges live in the same lane word, the next lane, the next wave (rows) or the next task (planes):
// cx_nb_locate.  One wave per batch; rounds of 64 cells, software-pipelined like the vertex stage: everything round s+1
// loads is issued -- and waited for -- before the stores of round s go out.
// The CPython-order quad diagonals come from a byte per lattice point (cx_k_hash_bytes, built once per grid shape):
// the slot the point's tuple hash takes in an 8-slot set and the slot it moves to when that one is taken, which is all
// a two-element set's iteration order depends on.
// Only for extractions without a wave on the tolerance path (counters[CX_CNT_NEAR] == 0): there the reference's
// np.allclose rules drop vertices and triangles per cell (the host then runs the staged kernels instead).
// =================================================================================================
struct cx_geomx {          // wave-uniform: the streaming wave tile a batch belongs to
    uint32_t w, wave, nsteps;
};
// where cell X + c lives, X = (plane step ps, streaming lane ls, row r, sample m) of the tile:
// bits 0-5 lane, 6-9 cell in the lane (4r+m), 11 next k segment, 12 next row group, 13-19 plane step, 20 next plane chunk
__device__ __forceinline__ uint32_t cx_nb_locate(const cx_geomx& GX, uint32_t ps, uint32_t ls, uint32_t r, uint32_t m, uint32_t c) {
    uint32_t mY = m + (c & 1u), rY = r + ((c >> 1) & 1u), psY = ps + (c >> 2), laneY = ls;
    uint32_t kf = 0, jf = 0, pf = 0;
    if (mY == 4u) {
        mY = 0; laneY++;
        if (laneY == 64u) { laneY = 0; kf = 1; }
    }
    if (rY == (uint32_t)CX_RJ) { rY = 0; jf = 1; }
    if (psY == GX.nsteps) { psY = 0; pf = 1; }
    return laneY | ((4u * rY + mY) << 6) | (kf << 11) | (jf << 12) | (psY << 13) | (pf << 20);
}
// streaming wave that holds the cell of a location word
__device__ __forceinline__ uint32_t cx_nb_wave(const cx_geomx& GX, const cx_task& T, uint32_t loc) {
    uint32_t wY = GX.w;
    if ((loc >> 11) & 1u) wY += 4u;                                              // next block in k
    if ((loc >> 12) & 1u) wY += (GX.wave == 3u) ? (4u * T.nks - 3u) : 1u;        // next wave / next block in j
    if ((loc >> 20) & 1u) wY += 4u * T.nks * T.njg;                              // next chunk of planes
    return wY;
}

// ---- per queue entry: first vertex of the cell relative to its wave's first vertex (24 bits) | crossing mask >> 1 << 24
__global__ __launch_bounds__(256) void cx_k_cell_info(const cx_params P, const cx_task T) {
    if (P.counters[CX_CNT_NEAR] != 0u) return;
    const uint32_t nbatches = min(P.counters[CX_CNT_BATCHES], P.fcap);
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t stride = gridDim.x * 4u;
    for (uint32_t f = blockIdx.x * 4u + wave; f < nbatches; f += stride) {
        const cx_bdesc D = P.flat[f];
        const cx_tile tile = cx_tile_of(P, T, D.w >> 2, D.w & 3u);
        cx_fast_geom G;
        G.pstart = tile.p; G.j0 = tile.j0; G.k0 = tile.k0;
        const uint32_t* __restrict__ q = P.queue + D.qofs;
        uint32_t* __restrict__ info = P.info + D.qofs;
        uint32_t vrel = D.vbase - P.wbase[D.w].v;
        for (uint32_t b0 = 0; b0 < D.n; b0 += 64u) {
            const uint32_t idx = b0 + lane;
            const bool have = idx < D.n;
            const uint32_t e = have ? q[idx] : 0u;
            uint32_t i, j, k;
            cx_decode_entry(P, G, e, i, j, k);
            const uint32_t sm = cx_entry_signs(e);
            const uint32_t vm = cx_corner_valid(P, i, j, k);
            const uint32_t s0 = (sm & 1u) ? 0xFFu : 0u;
            const uint32_t emask = have ? (((sm ^ s0) & vm) & 0xFEu) : 0u;
            uint32_t vtot;
            const uint32_t vpre = cx_wave_prefix_small<3>(__popc(emask), vtot);
            if (have) info[idx] = (vrel + vpre) | (emask << 23);
            vrel += vtot;
        }
    }
}

// ---- CPython 3.10 set order of a lattice point, one byte: bits 0-2 = slot of hash((i,j,k)) in an 8-slot table, bits 3-5 =
// the first slot of its probe sequence that differs from it (what it takes when another element sits in its slot; the same
// slot again if 16 probes never leave it).  py_set2_swapped(h1, h2) == (t2 != s1 ? t2 < s1 : alt2 < s1) with s = slot(h1),
// t2 = slot(h2), alt2 = alternative of h2: the loop there stops at the first probe of h2 that differs from s1 == t2.
__global__ void cx_k_hash_bytes(uint8_t* __restrict__ table, const uint64_t* __restrict__ hash_xy, uint32_t nrows, uint32_t n2, int32_t org2) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t row = idx / n2;
    if (row >= nrows) return;
    const uint32_t k = idx - row * n2;
    const uint64_t h = py_finish3(py_round_signed(hash_xy[row], (int32_t)k + org2));
    const uint32_t s0 = (uint32_t)h & 7u;
    uint32_t s = s0;
    uint64_t perturb = h;
    for (int guard = 0; guard < 16 && s == s0; guard++) {
        perturb >>= 5;
        s = (uint32_t)((s * 5u + 1u + perturb) & 7u);
    }
    table[idx] = (uint8_t)(s0 | (s << 3));
}

struct cx_mesh_lds {
    cx_tri_lds tri;                // triangle stage tables (per wave) + the LUT
    uint32_t vslot[4][2][448];     // per wave, double buffered: vertex o of a round -> (cell lane << 3) | direction
    uint32_t vbt[4][8];            // per wave: wbase[].v of the 8 waves a neighbour cell can live in
    uint32_t qbt[4][8];            // per wave: first queue entry of those waves
    uint32_t qab[4][8];            // per wave: offset of their plane-step tables in P.qa
    uint8_t ntri[256];
};
// A round of 64 queued cells passes three stages, one per loop iteration of the wave:
//   stage 1 (round s+2)  decode, locate the neighbour cells, request the queue words of their lanes
//   stage 2 (round s+1)  queue words -> queue positions -> request the info words; prefix sums, vertex slot table, sample
//                        loads, set-order codes
//   stage 3 (round s)    interpolation, triangle expansion, stores
// Everything an iteration loads is waited for ONCE, after the arithmetic of stage 3 and before its stores (`s_waitcnt
// vmcnt` retires loads and stores in issue order, and the number of stores varies: a load waited for behind them would
// wait for all of them).
struct cx_mstage1 {
    uint32_t e, e_next;
    uint32_t nb[6];            // neighbour cells X + c, c = 1..6: (queue position of the lane's first cell << 16) | active cells below Y
    uint32_t nbw;              // 3 bits per neighbour: which of the 8 candidate waves; bit 18+c: neighbour c is wanted
};
struct cx_mround {
    uint32_t e, lin, sm, emask, ntri, vpre, tpre, vtot, ttot;
    uint32_t vbase, tbase;
    uint32_t e2[CX_VR], sl[CX_VR];
    float f0[CX_VR], f1[CX_VR];
    uint32_t nb[6];            // info words of the neighbour cells
    uint32_t nbw;
    uint32_t hb[4];            // set-order codes of the 8 corners, (k, k+1) pairs of the 4 (i,j) columns
};
__device__ __forceinline__ void cx_mesh_stage1(const cx_params& P, const cx_task& T, const cx_fast_geom& G, const cx_geomx& GX,
                                               const uint32_t* q, uint32_t n, uint32_t b0, uint32_t lane, uint32_t e,
                                               const uint8_t* ntri_lut, const uint32_t* qab, cx_mstage1& S) {
    const uint32_t idx = b0 + lane;
    const bool have = idx < n;
    S.e = e;
    S.e_next = (idx + 64u < n) ? q[idx + 64u] : 0u;
    uint32_t i, j, k;
    cx_decode_entry(P, G, e, i, j, k);
    const uint32_t sm = cx_entry_signs(e);
    const bool voxel = have && cx_corner_valid(P, i, j, k) == 0xFFu && ntri_lut[sm] != 0;   // emits triangles
    // the neighbour cells that own a crossing edge of this voxel: the queue word of their lane.  X + c sits in the same
    // lane word unless X is in the last sample column (m == 3: next lane, or lane 0 of the next k segment), the last row
    // (r == 3: next wave) or the last plane step (next task); `qab` holds the table offsets of the 8 waves that can be.
    const uint32_t ps = (e >> 21) & 127u, ls = (e >> 15) & 63u, bit = (e >> 10) & 31u;
    const uint32_t rr = (bit * 11u) >> 6, mm = bit - CX_ROWBITS * rr;
    const uint32_t m3 = (mm == 3u) ? 1u : 0u, r3 = (rr == (uint32_t)CX_RJ - 1u) ? 1u : 0u, pl = (ps + 1u == GX.nsteps) ? 1u : 0u;
    const uint32_t kfl = m3 & ((ls == 63u) ? 1u : 0u);
    const uint32_t lane_k[2] = {ls, m3 ? ((ls + 1u) & 63u) : ls};              // lane of X + dk
    const uint32_t cm[2] = {mm, (mm + 1u) & 3u};                               // sample column of X + dk
    const uint32_t cr[2] = {4u * rr, 4u * ((rr + 1u) & 3u)};                   // 4 * row of X + dj
    const uint32_t pofs[2] = {ps << 6, pl ? 0u : ((ps + 1u) << 6)};            // plane step of X + di, times 64
    S.nbw = 0;
#pragma unroll
    for (uint32_t c = 1; c < 7; c++) {
        const uint32_t dk = c & 1u, dj = (c >> 1) & 1u, di = c >> 2;
        const uint32_t sc = ((sm >> c) & 1u) ? 0xFFu : 0u;
        uint32_t sup = 0;   // corners that are strict supersets of c: the far ends of the voxel edges corner c owns
#pragma unroll
        for (uint32_t c2 = 0; c2 < 8; c2++) sup |= ((c2 & c) == c && c2 != c) ? (1u << c2) : 0u;
        S.nb[c - 1u] = 0;
        if (voxel && ((sm ^ sc) & sup) != 0u && !(P.flags & CX_DBG_NO_LOOKUP)) {
            const uint32_t wsel = (dk ? kfl : 0u) | ((dj ? r3 : 0u) << 1) | ((di ? pl : 0u) << 2);
            const uint32_t qa = P.qa[qab[wsel] + pofs[di] + lane_k[dk]];
            // low 16 bits: the active cells below Y in its lane (their count completes the queue position)
            S.nb[c - 1u] = (qa & 0xFFFF0000u) | (qa & ((1u << (cr[dj] + cm[dk])) - 1u));
            S.nbw |= (wsel << (3u * (c - 1u))) | (1u << (18u + c));
        }
    }
}
__device__ __forceinline__ void cx_mesh_pin1(cx_mstage1& S) {
#pragma unroll
    for (uint32_t c = 0; c < 6; c++) asm volatile("" : "+v"(S.nb[c]) :: "memory");
    asm volatile("" : "+v"(S.e_next) :: "memory");
}
__device__ __forceinline__ void cx_mesh_stage2(const cx_params& P, const cx_task& T, const cx_fast_geom& G, const cx_mstage1& S,
                                               uint32_t n, uint32_t b0, uint32_t lane, uint32_t vbase, uint32_t tbase, uint32_t* slot,
                                               const uint8_t* ntri_lut, const uint32_t* qbt, cx_mround& R) {
    const float* __restrict__ A = P.grid;
    const uint32_t plane = P.n1 * P.n2;
    const uint32_t idx = b0 + lane;
    const bool have = idx < n;
    const uint32_t e = S.e;
    R.e = e;
    // the queue words are back: queue positions of the neighbour cells -> their info words
    R.nbw = S.nbw;
#pragma unroll
    for (uint32_t c = 0; c < 6; c++) {
        const uint32_t w = S.nb[c];
        uint32_t infow = 0;
        if ((S.nbw >> (19u + c)) & 1u) infow = P.info[qbt[(S.nbw >> (3u * c)) & 7u] + (w >> 16) + __popc(w & 0xFFFFu)];
        R.nb[c] = infow;
    }
    uint32_t i, j, k;
    cx_decode_entry(P, G, e, i, j, k);
    R.lin = (i * P.n1 + j) * P.n2 + k;
    R.sm = cx_entry_signs(e);
    const uint32_t vm = cx_corner_valid(P, i, j, k);
    const bool real_voxel = have && vm == 0xFFu;
    const uint32_t s0 = (R.sm & 1u) ? 0xFFu : 0u;
    R.emask = have ? (((R.sm ^ s0) & vm) & 0xFEu) : 0u;
    R.ntri = real_voxel ? (uint32_t)ntri_lut[R.sm] : 0u;
    const uint32_t nv = __popc(R.emask);
    R.vpre = cx_wave_prefix_small<3>(nv, R.vtot);
    R.tpre = cx_wave_prefix_small<4>(R.ntri, R.ttot);
    R.vbase = vbase; R.tbase = tbase;
#pragma unroll
    for (uint32_t d = 1; d < 8; d++)
        if ((R.emask >> d) & 1u) slot[R.vpre + __popc(R.emask & ((1u << d) - 1u))] = (lane << 3) | d;
    __builtin_amdgcn_wave_barrier();
    // sample loads of the first CX_VR x 64 vertices
#pragma unroll
    for (uint32_t r = 0; r < CX_VR; r++) {
        R.sl[r] = 0; R.e2[r] = 0; R.f0[r] = 0.0f; R.f1[r] = 1.0f;
        if (64u * r >= R.vtot) continue;   // wave-uniform
        const uint32_t o = 64u * r + lane;
        R.sl[r] = slot[(o < R.vtot) ? o : 0u];
        R.e2[r] = (uint32_t)__shfl((int)e, (int)(R.sl[r] >> 3));
        const uint32_t d = R.sl[r] & 7u;
        const uint32_t lin2 = cx_entry_lin(P, G, R.e2[r]);
        if (P.flags & CX_DBG_NO_VLOADS) { R.f0[r] = -1.0f; R.f1[r] = (float)lin2; continue; }
        R.f0[r] = A[lin2];
        R.f1[r] = A[lin2 + ((d & 4u) ? plane : 0u) + ((d & 2u) ? P.n2 : 0u) + (d & 1u)];
    }
    // CPython-order quad diagonals: the codes of the 8 corners, one 2-byte load per (i,j) column
    R.hb[0] = R.hb[1] = R.hb[2] = R.hb[3] = 0;
    if (P.hbytes && cx_need_hash(R.sm, 0u, R.ntri) != 0u) {   // only voxels: all 8 corners are inside the array
        const uint8_t* __restrict__ hb = P.hbytes + R.lin;
        uint16_t t0, t1, t2, t3;
        __builtin_memcpy(&t0, hb, 2); __builtin_memcpy(&t1, hb + P.n2, 2);
        __builtin_memcpy(&t2, hb + plane, 2); __builtin_memcpy(&t3, hb + plane + P.n2, 2);
        R.hb[0] = t0; R.hb[1] = t1; R.hb[2] = t2; R.hb[3] = t3;
    }
}
__device__ __forceinline__ void cx_mesh_pin2(cx_mround& R) {
#pragma unroll
    for (uint32_t r = 0; r < CX_VR; r++) asm volatile("" : "+v"(R.f0[r]), "+v"(R.f1[r]) :: "memory");
#pragma unroll
    for (uint32_t c = 0; c < 6; c++) asm volatile("" : "+v"(R.nb[c]) :: "memory");
    asm volatile("" : "+v"(R.hb[0]), "+v"(R.hb[1]), "+v"(R.hb[2]), "+v"(R.hb[3]) :: "memory");
}

// phase 1 of the triangle stage for the fused kernel: like cx_tri_phase1, with the quad diagonals from the corner codes
__device__ __forceinline__ uint32_t cx_mesh_phase1(const cx_params& P, cx_tri_lds& L, uint32_t lane, uint32_t wave, const cx_mround& R,
                                                   const uint32_t* vbt) {
    const uint32_t sm = R.sm, ntri = R.ntri;
    L.ve[wave][0][lane] = make_uint2(R.vbase + R.vpre, R.emask);
#pragma unroll
    for (uint32_t c = 1; c < 7; c++) {
        const uint32_t w = R.nb[c - 1u];
        const bool want = (R.nbw >> (18u + c)) & 1u;
        L.ve[wave][c][lane] = want ? make_uint2(vbt[(R.nbw >> (3u * (c - 1u))) & 7u] + (w & 0xFFFFFFu), (w >> 23) & 0xFEu) : make_uint2(0u, 0u);
    }
    uint32_t variants = 0;
    if (P.hbytes) {
        // code of corner c = byte (c & 1) of column c >> 1
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const uint32_t pat = ((sm >> CX_TC[t][0]) & 1u) | (((sm >> CX_TC[t][1]) & 1u) << 1) |
                                 (((sm >> CX_TC[t][2]) & 1u) << 2) | (((sm >> CX_TC[t][3]) & 1u) << 3);
            if (__popc(pat) != 2) continue;
            uint32_t l0 = 0, l1 = 0, h0 = 0, h1 = 0;
            int nl = 0, nh = 0;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t cc = CX_TC[t][m];
                const uint32_t code = (R.hb[cc >> 1] >> (8u * (cc & 1u))) & 0xFFu;
                if ((pat >> m) & 1u) { if (nl == 0) l0 = code; else l1 = code; nl++; }
                else { if (nh == 0) h0 = code; else h1 = code; nh++; }
            }
            if (cx_set2_swapped(l0, l1) != cx_set2_swapped(h0, h1)) variants |= 1u << t;
        }
    }
    L.u[wave].a.tfirst[lane] = R.tbase;   // triangle j of the round goes to tbase + j
    uint32_t pos = R.tpre;
#pragma unroll
    for (int t = 0; t < 6; t++) {
        const uint32_t pat = ((sm >> CX_TC[t][0]) & 1u) | (((sm >> CX_TC[t][1]) & 1u) << 1) |
                             (((sm >> CX_TC[t][2]) & 1u) << 2) | (((sm >> CX_TC[t][3]) & 1u) << 3);
        const uint32_t np = __popc(pat);
        const uint32_t nt = (ntri == 0u) ? 0u : ((np == 2u) ? 2u : (np & 1u));
        const uint32_t word = lane | (((uint32_t)t * 32u + pat * 2u + ((variants >> t) & 1u)) << 7);
        if (nt >= 1u) L.u[wave].a.slot[pos] = (uint16_t)word;
        if (nt == 2u) L.u[wave].a.slot[pos + 1u] = (uint16_t)(word | (1u << 6));
        pos += nt;
    }
    __builtin_amdgcn_wave_barrier();
    return R.ttot;
}

#ifndef CX_EM_MIN_WAVES
#define CX_EM_MIN_WAVES 1
#endif
__global__ __launch_bounds__(256, CX_EM_MIN_WAVES) void cx_k_emit_mesh(const cx_params P, const cx_task T) {
    __shared__ cx_mesh_lds L;
    if (P.counters[CX_CNT_NEAR] != 0u) return;   // a wave on the tolerance path: the host runs the staged kernels instead
    if (P.counters[CX_CNT_TRIS] > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap) return;   // host re-runs with more room
    const uint32_t nbatches = min(P.counters[CX_CNT_BATCHES], P.fcap);
    if (blockIdx.x * 4u >= nbatches) return;
    cx_tri_lds_init(L.tri);
    L.ntri[threadIdx.x] = cx_d_voxel_ntri[threadIdx.x];
    __syncthreads();
    const float* __restrict__ A = P.grid;
    const uint32_t plane = P.n1 * P.n2;
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nw = T.nblocks * 4u;
    const uint32_t stride = gridDim.x * 4u;
    uint32_t f = blockIdx.x * 4u + wave;
    if (f >= nbatches) return;
    cx_bdesc D = P.flat[f];
    for (;;) {   // waves are independent
        const uint32_t fn = f + stride;
        cx_bdesc Dn = D;
        if (fn < nbatches) Dn = P.flat[fn];   // next descriptor in flight while this batch is processed
        const cx_tile tile = cx_tile_of(P, T, D.w >> 2, D.w & 3u);
        cx_fast_geom G;
        G.pstart = tile.p; G.j0 = tile.j0; G.k0 = tile.k0;
        cx_geomx GX;
        GX.w = D.w; GX.wave = D.w & 3u; GX.nsteps = tile.ib - tile.p;
        // first vertex and first queue entry of the 8 streaming waves a neighbour cell can live in
        // (index: next k | next j << 1 | next chunk << 2)
        if (lane < 8u) {
            const uint32_t wY = min(cx_nb_wave(GX, T, ((lane & 1u) << 11) | (((lane >> 1) & 1u) << 12) | (((lane >> 2) & 1u) << 20)), nw - 1u);
            L.vbt[wave][lane] = P.wbase[wY].v;
            L.qbt[wave][lane] = wY * T.wcap;
            L.qab[wave][lane] = wY * (CX_SWP * 64u);
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t* __restrict__ q = P.queue + D.qofs;
        const uint32_t n = D.n;
        cx_mstage1 Sb, Sc;
        cx_mround Ra, Rb;
        uint32_t e0 = (lane < n) ? q[lane] : 0u;
        asm volatile("" : "+v"(e0) :: "memory");
        cx_mesh_stage1(P, T, G, GX, q, n, 0u, lane, e0, L.ntri, L.qab[wave], Sb);
        cx_mesh_pin1(Sb);
        cx_mesh_stage2(P, T, G, Sb, n, 0u, lane, D.vbase, D.tbase, L.vslot[wave][0], L.ntri, L.qbt[wave], Ra);
        if (64u < n) cx_mesh_stage1(P, T, G, GX, q, n, 64u, lane, Sb.e_next, L.ntri, L.qab[wave], Sc);
        cx_mesh_pin2(Ra);
        if (64u < n) { cx_mesh_pin1(Sc); Sb = Sc; }
        uint32_t par = 0;
        for (uint32_t b0 = 0; b0 < n; b0 += 64u) {
            const bool more1 = b0 + 64u < n, more2 = b0 + 128u < n;   // wave-uniform
            if (more2) cx_mesh_stage1(P, T, G, GX, q, n, b0 + 128u, lane, Sb.e_next, L.ntri, L.qab[wave], Sc);
            if (more1) cx_mesh_stage2(P, T, G, Sb, n, b0 + 64u, lane, Ra.vbase + Ra.vtot, Ra.tbase + Ra.ttot, L.vslot[wave][par ^ 1u],
                                      L.ntri, L.qbt[wave], Rb);
            // ---- stage 3 of round b0: what needs no store
            cx_vrec rec4[CX_VR];
#pragma unroll
            for (uint32_t r = 0; r < CX_VR; r++) rec4[r] = cx_vertex_record(P, G, Ra.e2[r], Ra.sl[r] & 7u, Ra.f0[r], Ra.f1[r]);
            const uint32_t ttot = cx_mesh_phase1(P, L.tri, lane, wave, Ra, L.vbt[wave]);
            if (more2) cx_mesh_pin1(Sc);
            if (more1) cx_mesh_pin2(Rb);
            // ---- the stores
            {
#pragma unroll
                for (uint32_t r = 0; r < CX_VR; r++) {
                    const uint32_t o = 64u * r + lane;
                    if (o < Ra.vtot && !(P.flags & CX_DBG_NO_VERTS)) CX_STORE_VERT(&P.verts[Ra.vbase + o], rec4[r]);
                }
                const uint32_t* slot = L.vslot[wave][par];
                for (uint32_t o0 = 64u * CX_VR; o0 < Ra.vtot; o0 += 64u) {   // more than CX_VR x 64 vertices: the rest one round at a time
                    const uint32_t o = o0 + lane;
                    const uint32_t sl = slot[(o < Ra.vtot) ? o : 0u];
                    const uint32_t e2 = (uint32_t)__shfl((int)Ra.e, (int)(sl >> 3));
                    const uint32_t d = sl & 7u;
                    const uint32_t lin2 = cx_entry_lin(P, G, e2);
                    const float f0 = A[lin2];
                    const float f1 = A[lin2 + ((d & 4u) ? plane : 0u) + ((d & 2u) ? P.n2 : 0u) + (d & 1u)];
                    const cx_vrec r4 = cx_vertex_record(P, G, e2, d, f0, f1);
                    if (o < Ra.vtot) CX_STORE_VERT(&P.verts[Ra.vbase + o], r4);
                }
            }
            cx_tri_phase2(P, L.tri, lane, wave, ttot);
            __builtin_amdgcn_wave_barrier();
            if (more1) Ra = Rb;
            if (more2) Sb = Sc;
            par ^= 1u;
        }
        if (fn >= nbatches) break;
        f = fn;
        D = Dn;
    }
}

#include "cx_tile3d.h"

// ---- launchers --------------------------------------------------------------------------------------
bool cx_fast_classify_supported(const cx_params& P) {
    return P.n2 >= 4u && ((reinterpret_cast<uintptr_t>(P.grid) & 3u) == 0u);   // a lane loads 4 samples of one row
}

bool cx_fast_classify_supported_dims(int64_t n2, const float* grid) {
    return n2 >= 4 && ((reinterpret_cast<uintptr_t>(grid) & 3u) == 0u);
}

cx_task cx_fast_task(uint32_t n0, uint32_t n1, uint32_t n2) {
    cx_task T;
    T.nks = (n2 + 255u) / 256u;
    T.njg = (n1 + 4u * CX_RJ - 1u) / (4u * CX_RJ);
    // planes per task: aim at ~3000 workgroups (12 per CU: measured best at 512^3), 2..64 planes each
    const uint32_t per_plane = T.nks * T.njg;
    const uint32_t target = cx_debug_knob("CX_TASKS", 3072u);
    uint32_t want_chunks = (target + per_plane - 1u) / per_plane;
    if (want_chunks < 1u) want_chunks = 1u;
    uint32_t ci = (n0 + want_chunks - 1u) / want_chunks;
    if (ci < 2u) ci = 2u;
    // thin slabs (one rank's share of a volume split over 8 GPUs): at least 4 planes per task -- with 2 every task streams 3 sample
    // planes for 2 planes of cells; 64 planes: 0.0446 -> 0.0426 ms per extraction two in flight
    if (ci < 4u && n0 > 8u && !cx_debug_knob("CX_TASKS", 0u)) ci = 4u;
    if (ci > CX_SWP - 1u) ci = CX_SWP - 1u;   // the sign words of ci + 1 planes are staged in LDS (and the entry format has 7 bits)
    T.ci = ci;
    T.nic = (n0 + ci - 1u) / ci;
    T.nblocks = T.nks * T.njg * T.nic;
    T.chunk = (T.nblocks + 7u) / 8u;
    T.wcap = CX_RJ * 256u * ci;
    T.bcap = T.wcap / CX_BATCH_MIN + 1u;
    T.bndcap = (256u + 2u * CX_RJ) * ci;     // per HALF tile
    T.div_nks = cx_fdiv_make(T.nks);
    T.div_njg = cx_fdiv_make(T.njg);
    return T;
}

void cx_launch_stream(const cx_params& P, const cx_task& T, hipStream_t s) {
    const bool aligned = (P.n2 % 4u == 0u) && ((reinterpret_cast<uintptr_t>(P.grid) & 15u) == 0u);
    const uint32_t ring = P.tq ? (uint32_t)(4u * 2u * (CX_RJ + 1u) * CX_PLW * sizeof(float)) : 0u;   // the staged sample planes (cx_stream_tile)
    if (aligned) hipLaunchKernelGGL(cx_k_stream<true>, dim3(T.chunk * 8u), dim3(256), ring, s, P, T);
    else hipLaunchKernelGGL(cx_k_stream<false>, dim3(T.chunk * 8u), dim3(256), ring, s, P, T);
}

void cx_launch_stream_levels(const cx_params* host_params, const cx_task& T, uint32_t nlevels, hipStream_t s) {
    const cx_params& P0 = host_params[0];
    const bool aligned = (P0.n2 % 4u == 0u) && ((reinterpret_cast<uintptr_t>(P0.grid) & 15u) == 0u);
    for (uint32_t l0 = 0; l0 < nlevels; l0 += CX_LEVELS_PER_LAUNCH) {      // more than 8 levels: another pass over the samples per 8
        const uint32_t n = std::min(CX_LEVELS_PER_LAUNCH, nlevels - l0);
        cx_params_pack pack;
        for (uint32_t k = 0; k < CX_LEVELS_PER_LAUNCH; k++) pack.p[k] = host_params[l0 + std::min(k, n - 1u)];
        if (aligned) hipLaunchKernelGGL(cx_k_stream_levels<true>, dim3(T.chunk * 8u * n), dim3(256), 0, s, pack, T, n);
        else hipLaunchKernelGGL(cx_k_stream_levels<false>, dim3(T.chunk * 8u * n), dim3(256), 0, s, pack, T, n);
    }
}
void cx_launch_scan_levels(const cx_params* device_params, const cx_task& T, uint32_t nlevels, hipStream_t s) {
    const uint32_t nw = T.nblocks * 4u;
    hipLaunchKernelGGL(cx_k_scan_list_levels, dim3((nw + 255u) / 256u, nlevels), dim3(256), 0, s, device_params, T, nw);
}

void cx_launch_scan_waves(const cx_params& P, const cx_task& T, hipStream_t s) {
    const uint32_t nw = T.nblocks * 4u;
    hipLaunchKernelGGL(cx_k_scan_list, dim3((nw + 255u) / 256u), dim3(256), 0, s, P, T, nw);
}

// one wave per batch, grid-stride: the batch count lives on the device, so launch what fills the chip
// a few times over (256 CUs) and let every wave walk the list
static uint32_t cx_batch_grid(const cx_params& P, uint32_t per_cu) {
    const uint32_t g = cx_debug_knob("CX_BGRID", 256u * per_cu);
    const uint32_t most = (P.fcap + 3u) / 4u;
    return g < most ? g : most;
}
// the vertex stage launches what is resident at once (24.8 KB of LDS per workgroup: 6 per CU) and gives every wave the same
// number of rounds; small grids: no more waves than batches could be
static uint32_t cx_vertex_grid(const cx_params& P) {
    // workgroups of the vertex stage the device holds at once, asked of the runtime (a grid beyond that starts a second
    // generation of waves when the first is two thirds through: measured +40 % kernel time with 6 per CU launched, 5 resident)
    static const uint32_t resident = [] {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cx_k_emit_vertices<true>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) return 256u * (uint32_t)per_cu;
        return (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
    }();
    const uint32_t g = cx_debug_knob("CX_BGRID", resident);
    const uint32_t most = (P.fcap + 3u) / 4u;
    return g < most ? g : most;
}
uint32_t cx_vertex_stage_waves(const cx_params& P) { return 4u * cx_vertex_grid(P); }
void cx_launch_emit_vertices(const cx_params& P, const cx_task& T, hipStream_t s) {
    if (P.tq) hipLaunchKernelGGL(cx_k_emit_vertices<true>, dim3(P.nvw / 4u), dim3(256), 0, s, P, T);
    else hipLaunchKernelGGL(cx_k_emit_vertices<false>, dim3(P.nvw / 4u), dim3(256), 0, s, P, T);
}


void cx_launch_classify_generic(const cx_params& P, hipStream_t s) {
    // contiguous cell ranges per block: multiples of 256, at least 16384, about 2048 blocks
    uint32_t cpb = (P.nsamples + 2047u) / 2048u;
    cpb = (cpb + 255u) & ~255u;
    if (cpb < 16384u) cpb = 16384u;
    const uint32_t blocks = (P.nsamples + cpb - 1u) / cpb;
    hipLaunchKernelGGL(cx_k_classify_generic, dim3(blocks), dim3(256), 0, s, P, cpb);
}

void cx_launch_emit_triangles(const cx_params& P, const uint64_t* hash_xy, hipStream_t s) {
    // one lane per record, grid-stride: launch what fills the chip a few times over
    uint32_t g = cx_debug_knob("CX_TGRID", 256u * 8u);
    const uint32_t most = (P.ccap + 255u) / 256u;
    if (g > most) g = most;
    if ((int32_t)P.org2 < 0) hipLaunchKernelGGL(cx_k_emit_triangles<true>, dim3(g ? g : 1u), dim3(256), 0, s, P, hash_xy);
    else hipLaunchKernelGGL(cx_k_emit_triangles<false>, dim3(g ? g : 1u), dim3(256), 0, s, P, hash_xy);
}

void cx_launch_emit_mesh(const cx_params& P, const cx_task& T, hipStream_t s) {
    hipLaunchKernelGGL(cx_k_cell_info, dim3(cx_batch_grid(P, 8u)), dim3(256), 0, s, P, T);
    hipLaunchKernelGGL(cx_k_emit_mesh, dim3(cx_batch_grid(P, 8u)), dim3(256), 0, s, P, T);
}

void cx_launch_hash_bytes(uint8_t* table, const uint64_t* hash_xy, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t org2, hipStream_t s) {
    const uint64_t n = (uint64_t)n0 * n1 * n2;
    hipLaunchKernelGGL(cx_k_hash_bytes, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, s, table, hash_xy, n0 * n1, n2, (int32_t)org2);
}

void cx_launch_emit_triangles_q(const cx_params& P, const cx_task& T, const uint64_t* hash_xy, hipStream_t s) {
    uint32_t g = cx_debug_knob("CX_TGRID", 256u * 8u);
    const uint32_t most = (P.ccap + 255u) / 256u;
    if (g > most) g = most;
    if ((int32_t)P.org2 < 0) hipLaunchKernelGGL(cx_k_emit_triangles_q<true>, dim3(g ? g : 1u), dim3(256), 0, s, P, T, hash_xy);
    else hipLaunchKernelGGL(cx_k_emit_triangles_q<false>, dim3(g ? g : 1u), dim3(256), 0, s, P, T, hash_xy);
}

// ---- fp32 grid coordinates of the vertex records, on request (cx_level0_download, cx_level0_device_ptrs): {x, y, z, bits(edge id)}
// with the crossing at q + t*d, in the same fp32 arithmetic the vertex stage used while its records still carried xyz
__global__ __launch_bounds__(256) void cx_k_expand_verts(const cx_vrec* __restrict__ recs, float4* __restrict__ out, uint32_t n, uint32_t n2, uint32_t plane,
                                                         cx_fdiv dplane, cx_fdiv drow) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= n) return;
    const cx_vrec r = recs[v];
    const uint32_t lin = r.x >> 3, d = r.x & 7u;
    const uint32_t i = cx_div(lin, dplane);
    const uint32_t rem = lin - i * plane;
    const uint32_t j = cx_div(rem, drow);
    const uint32_t k = rem - j * n2;
    const float t = __uint_as_float(r.y);
    const float fi = (float)i, fj = (float)j, fk = (float)k;
    float4 o;
    o.x = (d & 4u) ? fi + t : fi;
    o.y = (d & 2u) ? fj + t : fj;
    o.z = (d & 1u) ? fk + t : fk;
    o.w = __uint_as_float(r.x);
    out[v] = o;
}
void cx_launch_expand_verts(const cx_vrec* recs, float4* out, uint32_t n, uint32_t n1, uint32_t n2, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(cx_k_expand_verts, dim3((n + 255u) / 256u), dim3(256), 0, s, recs, out, n, n2, n1 * n2, cx_fdiv_make(n1 * n2), cx_fdiv_make(n2));
}

// the triangle stage's grid: what is resident at once, as for the vertex stage (its own count: it runs fewer waves per SIMD)
static uint32_t cx_triangle_grid(const cx_params& P) {
    static const uint32_t resident = [] {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cx_k_emit_triangles_e<false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) return 256u * (uint32_t)per_cu;
        return (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
    }();
    const uint32_t g = cx_debug_knob("CX_TGRID_E", resident);
    const uint32_t most = (P.fcap + 3u) / 4u;
    return g < most ? g : most;
}
uint32_t cx_triangle_stage_waves(const cx_params& P) { return 4u * cx_triangle_grid(P); }
void cx_launch_emit_triangles_e(const cx_params& P, const cx_task& T, const uint64_t* hash_xy, hipStream_t s) {
    if ((int32_t)P.org2 < 0) hipLaunchKernelGGL(cx_k_emit_triangles_e<true>, dim3(P.nkw / 4u), dim3(256), 0, s, P, T, hash_xy);
    else hipLaunchKernelGGL(cx_k_emit_triangles_e<false>, dim3(P.nkw / 4u), dim3(256), 0, s, P, T, hash_xy);
}

void cx_launch_hash_xy(uint64_t* table, uint32_t n0, uint32_t n1, uint32_t org0, uint32_t org1, hipStream_t s) {
    const uint32_t n = n0 * n1;
    hipLaunchKernelGGL(cx_k_hash_xy, dim3((n + 255u) / 256u), dim3(256), 0, s, table, n0, n1, org0, org1);
}
