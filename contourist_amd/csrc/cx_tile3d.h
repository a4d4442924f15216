// cx_tile3d.h -- the TILE EMIT path of the 3-D march (round 4).  Included by cx_march3d.hip behind the triangle-stage device
// functions it reuses (cx_vround_front, cx_tri_phase1 / cx_tri_phase2, ...).
//
// Why.  The staged pipeline hands vertex indices from its vertex stage to its triangle stage through HBM: an 8-byte info word per
// queue entry and a 16-byte record per cell written by one kernel and gathered by the next, plus the stream kernel's queue words
// (50 MB) that let the triangle stage find a neighbour's entry -- ~0.35 GB of the 1.66 GB a 512^3 extraction moves, and every
// Level-0 kernel runs at ~0.9 of what the fabric delivers: the bytes are the time.  Here ONE workgroup emits
// everything of HALF a streaming workgroup's tile (8 rows x 256 samples x ci planes: the queues of two streaming waves, two waves
// of this kernel per queue): pass A walks the queues, writes the vertex records and leaves, IN LDS, one word per queue entry (first
// vertex relative to the half tile | crossing mask) and one word per (plane step, streaming lane) that locates any cell's entry by
// arithmetic; pass B walks the queues again and expands the tetrahedra, reading the neighbours' words from LDS.  Nothing of that
// hand-over touches memory.
//
//   cell X + c, c = 1..6, owns the crossing edges of X's voxel that do not start at X (reference: the dict of interpolated pairs,
//   tetrahedral.py:164-188, 554-595).  Where it lives:
//     same half tile, same or next plane step LDS (this kernel)
//     first plane of the NEXT chunk of planes the queues of tile b + nks*njg start with their step-0 entries: pass A2 reads those
//                                            fronts too (their numbering is a prefix sum from the front) -- LDS
//     next half tile in j, next tile in k    (13 % of cells) not known to this workgroup: the voxel becomes a 16-byte BOUNDARY
//                                            RECORD and cx_k_tile_boundary expands it afterwards, reading the neighbours' words
//                                            from two small global arrays that hold them for the cells on the j- and k-faces of
//                                            every half tile (P.fj: rows j % 8 in {0, 7}; P.fk: samples k % 256 in {0, 255};
//                                            written by pass A)
// Vertex and triangle NUMBERING is the staged pipeline's (queue order per streaming wave, waves in scan order): the meshes are
// bit-identical, boundary voxels just get their triangles from the second kernel.
// Occupancy decides this kernel (first build: one workgroup of 4 waves per whole tile, 59 KB of LDS, 2 waves per SIMD: 0.49 ms at
// 512^3, 0.29 ms with every global access switched off -- latency of its own instruction chains): the triangle tables hold ONE
// packed word per (corner, cell) (cx_tri_lds_p), a workgroup holds 38 KB, four fit a CU.
// Not handled here (the host runs the staged kernels instead, cx_counts_get): an extraction with a wave on the tolerance path
// (counters[CX_CNT_NEAR]), and a half tile whose entries do not fit the LDS words (counters[CX_CNT_TILEOVF]: > P.tile_cap active
// cells, ~9 % of its cells -- white noise).
#pragma once

#define CX_TILE_PASSA_BYTES 5888u     // per wave: two slot tables of 448 words + the corner table of 64 x 9 floats (as the vertex stage)

struct cx_tile_shared {
    uint32_t cnt[2][2];      // per queue: own entries, step-0 entries of the next chunk's wave
    uint32_t bn[4];          // per wave: boundary voxels
    uint32_t vb[2];          // first vertex of this half tile / of the half tile one chunk of planes up
    uint8_t ntri[256];
    uint64_t hcol[CX_SWP][10];   // CPython tuple-hash prefixes of the half tile's (i, j) columns: planes p .. p + ci, rows j0 .. j0 + 8
};

// lane l takes x of lane l - 1; lane 0 takes `first`
__device__ __forceinline__ uint32_t cx_wave_shr1(uint32_t x, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)x, 0x138, 0xF, 0xF, false);   // wave_shr:1
}

// pass B, front half of a round of 64 queue entries: everything the triangle stage needs about each cell, the neighbours' packed
// words from LDS (rec.w / nb[].x: as cx_tri_lds_p takes them)
__device__ __forceinline__ void cx_tile_front(const cx_params& P, const cx_fast_geom& G, const uint64_t (*hcol)[10],
                                              const uint32_t* qaw, const uint32_t* info, const uint32_t* infw, const uint2* pats,
                                              const uint8_t* ntri_lut, uint32_t end, uint32_t b0, uint32_t lane, uint32_t qi,
                                              uint32_t S1n, uint32_t e, uint32_t vb0, uint32_t& run_t, uint32_t& brank,
                                              uint4* __restrict__ bout, uint32_t bcap, cx_tri_in& I) {
    const uint32_t idx = b0 + lane;
    const bool have = idx < end;
    const uint32_t ps = (e >> 21) & 127u, ls = (e >> 15) & 63u, bit = (e >> 10) & 31u;
    const uint32_t rr = (bit * 11u) >> 6, mm = bit - 6u * rr;
    uint32_t i, j, k;
    cx_decode_entry(P, G, e, i, j, k);
    const uint32_t lin = (i * P.n1 + j) * P.n2 + k;
    const uint32_t sm = have ? cx_entry_signs(e) : 0u;
    const bool real = have && cx_corner_valid(P, i, j, k) == 0xFFu;
    const uint32_t ntri = real ? (uint32_t)ntri_lut[sm] : 0u;
    uint32_t ttot;
    const uint32_t tpre = cx_wave_prefix_small<4>(ntri, ttot);
    const uint32_t tfirst = run_t + tpre;
    run_t += ttot;
    const uint32_t own = have ? infw[idx] : 0u;
    const uint32_t emask = (own >> 23) & 0xFEu;
    const bool m3 = (mm == 3u), r3 = (rr == 3u);
    const bool bnd = ntri != 0u && ((qi == 1u && r3) || (ls == 63u && m3));   // a neighbour cell in the next half tile in j or the next tile in k
    const uint32_t tetskip = real ? 0u : 0x3Fu;
    const uint64_t bm = __ballot(bnd);
    if (bnd) {
        const uint32_t at = brank + cx_mbcnt(bm);
        if (at < bcap) bout[at] = make_uint4(lin, sm | (tetskip << 8) | (ntri << 16) | (emask << 24), tfirst, vb0 + (own & 0x7FFFFFu));
    }
    brank += (uint32_t)__popcll(bm);
    const uint32_t ntri_e = bnd ? 0u : ntri;
    const uint32_t want = (ntri_e && !(P.flags & CX_DBG_NO_LOOKUP)) ? ((pats[sm].y >> 9) & 0x3Fu) : 0u;
#pragma unroll
    for (uint32_t c = 1; c < 7; c++) {
        const uint32_t dk = c & 1u, dj = (c >> 1) & 1u, di = c >> 2;
        uint32_t wd = 0u;
        if ((want >> (c - 1u)) & 1u) {
            const uint32_t wv = qi + ((dj && r3) ? 1u : 0u), st = ps + di, ln = ls + ((dk && m3) ? 1u : 0u);
            const uint32_t cell = 4u * ((rr + dj) & 3u) + ((mm + dk) & 3u);
            const uint32_t qa = qaw[(wv * S1n + st) * 64u + ln];
            wd = info[(qa >> 16) + __popc(qa & ((1u << cell) - 1u))];
        }
        I.nb[c - 1u] = make_uint2(wd, 0u);
    }
    I.rec = make_uint4(lin, sm | (tetskip << 8) | (ntri_e << 16) | (emask << 24), tfirst, own);
    I.ck = k;
    I.hb[0] = I.hb[1] = I.hb[2] = I.hb[3] = 0;
    I.hxy[0] = I.hxy[1] = I.hxy[2] = I.hxy[3] = 0;
    if ((P.flags & CX_DIAG_CPYTHON310) && cx_need_hash(sm, tetskip, ntri_e) != 0u) {
        // only voxels get here: the prefixes of the voxel's four (i, j) columns, from the half tile's copy in LDS -- pass B has no
        // global load but its queue entries, so nothing ever waits for its triangle stores
        const uint32_t jr = 4u * qi + rr;
        I.hxy[0] = hcol[ps][jr]; I.hxy[1] = hcol[ps][jr + 1u]; I.hxy[2] = hcol[ps + 1u][jr]; I.hxy[3] = hcol[ps + 1u][jr + 1u];
    }
}
#ifndef CX_TILE_MIN_WAVES
#define CX_TILE_MIN_WAVES 4
#endif
template <bool NEG_ORIGIN>
__global__ __launch_bounds__(256, CX_TILE_MIN_WAVES) void cx_k_tile_emit(const cx_params P, const cx_task T, const uint64_t* __restrict__ hash_xy) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];      // queue words [2][ci + 1][64] | info words [tile_cap + 16]
    __shared__ __attribute__((aligned(16))) unsigned char s_u[sizeof(cx_tri_lds_p) > 4u * CX_TILE_PASSA_BYTES ? sizeof(cx_tri_lds_p) : 4u * CX_TILE_PASSA_BYTES];
    __shared__ cx_tile_shared SH;
    if (P.counters[CX_CNT_NEAR] != 0u) return;                                            // the host runs the staged kernels instead
    // workgroup -> half tile: the scan kernel's lists, the class with the most queue entries first (P.torder); half tiles that
    // nothing crosses are in no list
    uint32_t ht;
    {
        const uint32_t n0 = P.counters[CX_CNT_TCLS], n1 = P.counters[CX_CNT_TCLS + 1u], n2c = P.counters[CX_CNT_TCLS + 2u];
        uint32_t i = blockIdx.x, cls = 0;
        if (i >= n0) { i -= n0; cls = 1u; if (i >= n1) { i -= n1; cls = 2u; if (i >= n2c) return; } }
        ht = P.torder[(size_t)cls * (T.nblocks * 2u) + i];
    }
    const uint32_t b = ht >> 1, h = ht & 1u;
    if (P.counters[CX_CNT_TRIS] > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap) return;    // host re-runs with more room
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qi = wave >> 1, half = wave & 1u;          // which of the two queues, which half of its rounds
    // diagnostic (cx_debug_stamps): per wave of this kernel {start, pass A done, end, its queue entries} in 10 ns ticks, behind the stream kernel's stamps
    unsigned long long* kst = P.stamps ? P.stamps + (size_t)T.nblocks * 16u + ((size_t)(2u * b + h) * 4u + wave) * 4u : nullptr;
    if (kst && lane == 0) kst[0] = __builtin_amdgcn_s_memrealtime();
    const uint32_t sw = 2u * h + qi;                          // streaming wave inside the tile
    const uint32_t w = b * 4u + sw;
    const cx_tile tile = cx_tile_of(P, T, b, sw);
    const uint32_t nsteps = (tile.p < tile.ib) ? tile.ib - tile.p : 0u;
    const uint32_t S1n = T.ci + 1u;
    uint32_t* qaw = s_dyn;
    uint32_t* info = s_dyn + 2u * S1n * 64u;
    const uint32_t nq = __builtin_amdgcn_readfirstlane(P.wsum[w].nq);
    const uint32_t wv0 = __builtin_amdgcn_readfirstlane(P.wbase[w].v), wt0 = __builtin_amdgcn_readfirstlane(P.wbase[w].t);
    const uint32_t* __restrict__ q = P.queue + (size_t)w * T.wcap;
    // the tile one chunk of planes up: its queues start with the cells of its first plane
    const uint32_t per_chunk = T.nks * T.njg;
    const bool has_next = (b / per_chunk + 1u < T.nic) && nsteps == T.ci;
    const uint32_t w2 = w + 4u * per_chunk;
    const uint32_t* __restrict__ q2 = P.queue + (size_t)(has_next ? w2 : w) * T.wcap;
    if (half == 0u) {
        uint32_t n2 = 0;
        if (has_next) {
            const uint32_t nq2 = __builtin_amdgcn_readfirstlane(P.wsum[w2].nq);
            for (uint32_t b0 = 0; b0 < nq2; b0 += 64u) {
                const uint32_t idx = b0 + lane;
                const uint32_t e = (idx < nq2) ? q2[idx] : 0xFFFFFFFFu;
                const uint64_t m = __ballot(idx < nq2 && ((e >> 21) & 127u) == 0u);
                n2 += (uint32_t)__popcll(m);
                if (m != ~0ULL) break;
            }
        }
        if (lane == 0) { SH.cnt[qi][0] = nq; SH.cnt[qi][1] = n2; }
    }
    if (threadIdx.x == 0) { SH.vb[0] = P.wbase[b * 4u + 2u * h].v; SH.vb[1] = has_next ? P.wbase[b * 4u + 2u * h + 4u * per_chunk].v : 0u; }
    SH.ntri[threadIdx.x] = cx_d_voxel_ntri[threadIdx.x];
    for (uint32_t x = threadIdx.x; x < 2u * S1n * 64u; x += 256u) qaw[x] = 0u;
    if ((P.flags & CX_DIAG_CPYTHON310) && threadIdx.x < (nsteps + 1u) * 9u) {
        const uint32_t pi = threadIdx.x / 9u, jr = threadIdx.x - 9u * pi;
        const uint32_t jh = (tile.j0 - qi * CX_RJ) + jr;        // first row of the half tile + jr
        SH.hcol[pi][jr] = hash_xy[min(tile.p + pi, P.n0 - 1u) * P.n1 + min(jh, P.n1 - 1u)];
    }
    __syncthreads();
    const uint32_t tot_own = SH.cnt[0][0] + SH.cnt[1][0], tot_next = SH.cnt[0][1] + SH.cnt[1][1];
    const uint32_t off_own = qi ? SH.cnt[0][0] : 0u;
    const uint32_t off_next = tot_own + (qi ? SH.cnt[0][1] : 0u);
    const uint32_t n2 = SH.cnt[qi][1];
    if (tot_own + tot_next > P.tile_cap) {       // workgroup-uniform
        if (threadIdx.x == 0) P.counters[CX_CNT_TILEOVF] = 1u;
        return;
    }
    const uint32_t vb0 = SH.vb[0], vb1 = SH.vb[1];
    cx_fast_geom G;
    G.pstart = tile.p; G.j0 = tile.j0; G.k0 = tile.k0;
    uint32_t* infw = info + off_own;
    uint32_t* qw = qaw + qi * S1n * 64u;
    // the wave's share of its queue: the first or the second half of the rounds of 64 entries; the second half starts from what the
    // first holds (vertices, triangles: counted from the entries, cx_skip_rounds)
    const uint32_t rounds = (nq + 63u) >> 6, hsplit = (rounds + 1u) >> 1;
    const uint32_t first = half ? 64u * hsplit : 0u, end = half ? nq : min(nq, 64u * hsplit);
    cx_run run0;
    run0.v = wv0; run0.t = wt0; run0.c = 0; run0.b = 0; run0.s = 0;
    if (half && first < end) cx_skip_rounds(P, G, q, hsplit, lane, SH.ntri, run0);
    uint32_t nbnd = 0;
    // ---- pass A: vertex records (the vertex stage's rounds, cx_emit_queue_fast), info words and queue words in LDS, face words
    if (first < end) {
        uint32_t* slot2 = reinterpret_cast<uint32_t*>(s_u + wave * CX_TILE_PASSA_BYTES);
        float* corners = reinterpret_cast<float*>(slot2 + 2u * 448u);
        cx_vround Ra, Rb;
        uint32_t e0 = (first + lane < end) ? q[first + lane] : 0u;
        uint32_t carry = half ? (q[first - 1u] >> 15) : 0xFFFFFFFFu;       // (plane step, lane) of the entry before the wave's first
        asm volatile("" : "+v"(e0), "+v"(carry) :: "memory");
        carry = __builtin_amdgcn_readfirstlane(carry);
        cx_vround_front(P, G, q, end, first, lane, e0, run0, slot2, SH.ntri, Ra);
        cx_vround_pin(Ra);
        uint32_t par = 0;
        for (uint32_t b0 = first; b0 < end; b0 += 64u) {
            const bool more = b0 + 64u < end;   // wave-uniform
            if (more) {
                cx_run nb = Ra.base;
                nb.v += Ra.vtot; nb.t += Ra.ttot; nb.c += Ra.ctot;
                cx_vround_front(P, G, q, end, b0 + 64u, lane, Ra.e_next, nb, slot2 + (par ^ 1u) * 448u, SH.ntri, Rb);
            }
#pragma unroll
            for (uint32_t c = 0; c < 8; c++) corners[lane * CX_CORNER_ROW + c] = ((c & 1u) || Ra.vk) ? Ra.f[c] : Ra.f[c + 1u];
            __builtin_amdgcn_wave_barrier();
            const uint32_t* slot = slot2 + par * 448u;
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
            // what pass B (and the boundary kernel) look up: the cell's word per queue entry, the word per (plane step, lane)
            {
                const uint32_t idx = b0 + lane;
                const bool have = idx < end;
                const uint32_t e = Ra.e;
                const uint32_t key = have ? (e >> 15) : 0xFFFFFFFEu;    // plane step << 6 | streaming lane
                const uint32_t pk = cx_wave_shr1(key, carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)key, 63);
                const uint32_t ps = (e >> 21) & 127u, ls = (e >> 15) & 63u, bit = (e >> 10) & 31u;
                const uint32_t rr = (bit * 11u) >> 6, mm = bit - 6u * rr;
                const bool jlo = (qi == 0u && rr == 0u), jhi = (qi == 1u && rr == 3u);
                const bool klo = (ls == 0u && mm == 0u), khi = (ls == 63u && mm == 3u);
                if (have) {
                    const uint32_t vfirst = Ra.base.v + Ra.vpre;
                    infw[idx] = (vfirst - vb0) | (Ra.emask << 23);
                    atomicOr(&qw[ps * 64u + ls], (1u << (4u * rr + mm)) | ((pk != key) ? ((off_own + idx) << 16) : 0u));
                    if (jlo | jhi | klo | khi) {
                        uint32_t i, j, k;
                        cx_decode_entry(P, G, e, i, j, k);
                        const uint64_t val = ((uint64_t)Ra.emask << 32) | (uint64_t)vfirst;
                        if (jlo | jhi) P.fj[((size_t)i * (4u * T.njg) + 2u * (j >> 3) + (jhi ? 1u : 0u)) * P.n2 + k] = val;
                        if (klo | khi) P.fk[((size_t)i * P.n1 + j) * (2u * T.nks) + 2u * (k >> 8) + (khi ? 1u : 0u)] = val;
                    }
                }
                nbnd += (uint32_t)__popcll(__ballot(have && Ra.ntri != 0u && (jhi || khi)));
            }
            __builtin_amdgcn_wave_barrier();
            if (more) cx_vround_pin(Rb);   // the next round's samples are waited for AFTER this round's stores went out
            if (more) Ra = Rb;
            par ^= 1u;
        }
    }
    // ---- pass A2: the first plane of the next chunk (info words with bit 31 set: relative to THAT half tile's first vertex)
    if (half == 0u && n2) {
        cx_fast_geom G2;
        G2.pstart = tile.ib; G2.j0 = tile.j0; G2.k0 = tile.k0;
        uint32_t vrel = __builtin_amdgcn_readfirstlane(P.wbase[w2].v) - vb1;
        uint32_t* inf2 = info + off_next;
        uint32_t* qw2 = qw + nsteps * 64u;
        uint32_t carry = 0xFFFFFFFFu;
        for (uint32_t b0 = 0; b0 < n2; b0 += 64u) {
            const uint32_t idx = b0 + lane;
            const bool have = idx < n2;
            const uint32_t e = have ? q2[idx] : 0u;
            uint32_t i, j, k;
            cx_decode_entry(P, G2, e, i, j, k);
            const uint32_t vm = cx_corner_valid(P, i, j, k);
            const uint32_t sm = cx_entry_signs(e);
            const uint32_t s0 = (sm & 1u) ? 0xFFu : 0u;
            const uint32_t emask = have ? (((sm ^ s0) & vm) & 0xFEu) : 0u;
            uint32_t vtot;
            const uint32_t vpre = cx_wave_prefix_small<3>(__popc(emask), vtot);
            const uint32_t key = have ? (e >> 15) : 0xFFFFFFFEu;
            const uint32_t pk = cx_wave_shr1(key, carry);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)key, 63);
            const uint32_t ls = (e >> 15) & 63u, bit = (e >> 10) & 31u;
            const uint32_t rr = (bit * 11u) >> 6, mm = bit - 6u * rr;
            if (have) {
                inf2[idx] = (vrel + vpre) | (emask << 23) | 0x80000000u;
                atomicOr(&qw2[ls], (1u << (4u * rr + mm)) | ((pk != key) ? ((off_next + idx) << 16) : 0u));
            }
            vrel += vtot;
        }
    }
    if (lane == 0) SH.bn[wave] = nbnd;
    if (kst && lane == 0) { kst[1] = __builtin_amdgcn_s_memrealtime(); kst[3] = (first < end) ? end - first : 0u; }
    __syncthreads();                       // pass A of all four waves is complete: LDS words, and the pass-A tables are dead
    cx_tri_lds_p& L = *reinterpret_cast<cx_tri_lds_p*>(s_u);
    cx_tri_lds_init(L);
    if (threadIdx.x == 0) { L.vb[0] = vb0; L.vb[1] = vb1; }
    __syncthreads();
    uint32_t bnd_off = 0, bnd_tot = 0;
#pragma unroll
    for (uint32_t ww = 0; ww < 4u; ww++) {
        if (ww < wave) bnd_off += SH.bn[ww];
        bnd_tot += SH.bn[ww];
    }
    if (threadIdx.x == 0) P.bndn[2u * b + h] = min(bnd_tot, T.bndcap);
    // ---- pass B: the tetrahedra of every voxel whose neighbour cells this workgroup knows
    if (first < end) {
        uint4* __restrict__ bout = P.bnd + (size_t)(2u * b + h) * T.bndcap + bnd_off;
        const uint32_t bcap = (bnd_off < T.bndcap) ? T.bndcap - bnd_off : 0u;
        uint32_t run_t = run0.t, brank = 0;
        // the queue entries of four rounds are requested together and waited for once: a round itself loads nothing from memory, so
        // its triangle stores are never waited for (`s_waitcnt vmcnt` retires loads and stores in issue order: a round that loaded
        // anything after the previous round's stores would wait for those stores' round trip)
        cx_tri_in Ia;
        for (uint32_t g0 = first; g0 < end; g0 += 256u) {       // wave-uniform
            uint32_t eg0 = (g0 + lane < end) ? q[g0 + lane] : 0u, eg1 = (g0 + 64u + lane < end) ? q[g0 + 64u + lane] : 0u;
            uint32_t eg2 = (g0 + 128u + lane < end) ? q[g0 + 128u + lane] : 0u, eg3 = (g0 + 192u + lane < end) ? q[g0 + 192u + lane] : 0u;
            asm volatile("" : "+v"(eg0), "+v"(eg1), "+v"(eg2), "+v"(eg3) :: "memory");
            const uint32_t gend = min(end, g0 + 256u);
            for (uint32_t b0 = g0; b0 < gend; b0 += 64u) {
                cx_tile_front(P, G, SH.hcol, qaw, info, infw, L.pats, SH.ntri, end, b0, lane, qi, S1n, eg0, vb0, run_t, brank, bout, bcap, Ia);
                const uint32_t ttot = cx_tri_phase1<NEG_ORIGIN>(P, L, lane, wave, Ia);
                cx_tri_phase2(P, L, lane, wave, ttot);
                eg0 = eg1; eg1 = eg2; eg2 = eg3;
            }
        }
    }
    if (kst && lane == 0) kst[2] = __builtin_amdgcn_s_memrealtime();
}

// ---- the voxels on the high j / k faces of the half tiles: their neighbours' words from the face arrays
__device__ __forceinline__ void cx_tile_fetch_faces(const cx_params& P, const cx_task& T, const uint64_t* __restrict__ hash_xy, const uint2* pats,
                                                    const uint4& rec, cx_tri_in& I) {
    const uint32_t plane = P.n1 * P.n2;
    I.rec = rec;
    const uint32_t lin = rec.x, sm = rec.y & 0xFFu, tetskip = (rec.y >> 8) & 0x3Fu, ntri = (rec.y >> 16) & 0xFFu;
    const uint32_t ci = cx_div(lin, P.div_plane);
    const uint32_t r = lin - ci * plane;
    const uint32_t cj = cx_div(r, P.div_row);
    const uint32_t ck = r - cj * P.n2;
    const bool jface = (cj & 7u) == 7u;     // on the high j face of its half tile: every neighbour is in a face row (else: in a face column)
    const uint32_t want = (ntri && !(P.flags & CX_DBG_NO_LOOKUP)) ? ((pats[sm].y >> 9) & 0x3Fu) : 0u;
#pragma unroll
    for (uint32_t c = 1; c < 7; c++) {
        uint2 pr = make_uint2(0u, 0u);
        if ((want >> (c - 1u)) & 1u) {
            const uint32_t yi = ci + (c >> 2), yj = cj + ((c >> 1) & 1u), yk = ck + (c & 1u);
            uint64_t e;
            if (jface) e = P.fj[((size_t)yi * (4u * T.njg) + 2u * (yj >> 3) + (((yj & 7u) == 7u) ? 1u : 0u)) * P.n2 + yk];
            else e = P.fk[((size_t)yi * P.n1 + yj) * (2u * T.nks) + 2u * (yk >> 8) + (((yk & 255u) == 255u) ? 1u : 0u)];
            pr = make_uint2((uint32_t)e, (uint32_t)(e >> 32));
        }
        I.nb[c - 1u] = pr;
    }
    I.ck = ck;
    I.hb[0] = I.hb[1] = I.hb[2] = I.hb[3] = 0;
    I.hxy[0] = I.hxy[1] = I.hxy[2] = I.hxy[3] = 0;
    if ((P.flags & CX_DIAG_CPYTHON310) && cx_need_hash(sm, tetskip, ntri) != 0u) {
        const uint32_t i1 = min(ci + 1u, P.n0 - 1u), j1 = min(cj + 1u, P.n1 - 1u);
        I.hxy[0] = hash_xy[ci * P.n1 + cj]; I.hxy[1] = hash_xy[ci * P.n1 + j1];
        I.hxy[2] = hash_xy[i1 * P.n1 + cj]; I.hxy[3] = hash_xy[i1 * P.n1 + j1];
    }
}
template <bool NEG_ORIGIN>
__global__ __launch_bounds__(256) void cx_k_tile_boundary(const cx_params P, const cx_task T, const uint64_t* __restrict__ hash_xy) {
    __shared__ cx_tri_lds L;
    if (P.counters[CX_CNT_NEAR] != 0u || P.counters[CX_CNT_TILEOVF] != 0u) return;
    if (P.counters[CX_CNT_TRIS] > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap) return;
    const uint32_t b = blockIdx.x;             // half tile
    const uint32_t n = min(P.bndn[b], T.bndcap);
    if (n == 0u) return;                       // whole block
    cx_tri_lds_init(L);
    __syncthreads();
    const uint32_t lane = cx_lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4* __restrict__ recs = P.bnd + (size_t)b * T.bndcap;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    for (uint32_t b0 = wave * 64u; b0 < n; b0 += 256u) {     // wave-uniform
        const uint32_t idx = b0 + lane;
        const uint4 rec = (idx < n) ? recs[idx] : zero;
        cx_tri_in I;
        cx_tile_fetch_faces(P, T, hash_xy, L.pats, rec, I);
        const uint32_t ttot = cx_tri_phase1<NEG_ORIGIN>(P, L, lane, wave, I);
        cx_tri_phase2(P, L, lane, wave, ttot);
    }
}

uint32_t cx_tile_cap_default() { return 2048u; }
static uint32_t cx_tile_lds_bytes(const cx_task& T, uint32_t tile_cap) { return (2u * (T.ci + 1u) * 64u + tile_cap + 16u) * (uint32_t)sizeof(uint32_t); }
void cx_launch_tile_emit(const cx_params& P, const cx_task& T, const uint64_t* hash_xy, hipStream_t s) {
    const uint32_t lds = cx_tile_lds_bytes(T, P.tile_cap);
    if ((int32_t)P.org2 < 0) hipLaunchKernelGGL(cx_k_tile_emit<true>, dim3(T.nblocks * 2u), dim3(256), lds, s, P, T, hash_xy);
    else hipLaunchKernelGGL(cx_k_tile_emit<false>, dim3(T.nblocks * 2u), dim3(256), lds, s, P, T, hash_xy);
    if ((int32_t)P.org2 < 0) hipLaunchKernelGGL(cx_k_tile_boundary<true>, dim3(T.nblocks * 2u), dim3(256), 0, s, P, T, hash_xy);
    else hipLaunchKernelGGL(cx_k_tile_boundary<false>, dim3(T.nblocks * 2u), dim3(256), 0, s, P, T, hash_xy);
}
