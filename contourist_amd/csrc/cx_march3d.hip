// cx_march3d.hip -- Level-0 kernels of the 3-D marching-tetrahedra voxel march for gfx950 (wave64).
//
// Reference semantics restated here (paths relative to the reference checkout, contourist/...):
//   border_voxel                      tetrahedral.py:383-394
//   enumerate_tetrahedron_triangles   tetrahedral.py:561-595
//   contour_pair_interpolation        tetrahedral.py:471-487
// Data layout / kernel plan: DESIGN.md.
//
// Lattice "cell" q = lattice point (i,j,k) seen as the lower corner of the voxel [q, q+1].
// A cell OWNS the 7 lattice edges q -> q+d, d = 4di+2dj+dk in 1..7, and (when all 8 corners are
// inside the array) the 6 Kuhn tetrahedra of its voxel.  Vertex id of an edge = (lin(q) << 3) | d.
#include "cx_common.h"

__device__ __constant__ uint8_t cx_d_tet_corners[6][4] = CX_TET_CORNERS_INIT;
__device__ __constant__ uint64_t cx_d_tet_tris[6][16][2] = CX_TET_TRIS_INIT;
__device__ __constant__ uint8_t cx_d_voxel_ntri[256] = CX_VOXEL_NTRI_INIT;

// corner masks of the 6 tetrahedra: bit c set <=> cube corner c is a vertex of tet t
#define CX_TETMASK(t) (uint32_t)((1u << cx_d_tet_corners[t][0]) | (1u << cx_d_tet_corners[t][1]) | \
                                  (1u << cx_d_tet_corners[t][2]) | (1u << cx_d_tet_corners[t][3]))

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// exclusive prefix sum over the wave of a small per-lane count (< 2^NBITS), plus the wave total,
// from NBITS ballots (no LDS, no cross-lane data movement).
template <int NBITS>
__device__ __forceinline__ uint32_t wave_prefix_small(uint32_t x, uint32_t& total) {
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int b = 0; b < NBITS; b++) {
        uint64_t m = __ballot((x >> b) & 1u);
        pre += mbcnt(m) << b;
        tot += (uint32_t)__popcll(m) << b;
    }
    total = tot;
    return pre;
}

// pattern of tet t inside a voxel sign mask: bit m set <=> tet vertex m is low
__device__ __forceinline__ uint32_t tet_pattern(uint32_t sm, int t) {
    return ((sm >> cx_d_tet_corners[t][0]) & 1u) | (((sm >> cx_d_tet_corners[t][1]) & 1u) << 1) |
           (((sm >> cx_d_tet_corners[t][2]) & 1u) << 2) | (((sm >> cx_d_tet_corners[t][3]) & 1u) << 3);
}

__device__ __forceinline__ uint32_t tet_ntri(uint32_t pattern) {
    uint32_t n = __popc(pattern);
    return (n == 2) ? 2u : ((n == 1 || n == 3) ? 1u : 0u);
}

// np.allclose(value, f) for one sample (border_voxel, tetrahedral.py:391): |v-f| <= 1e-8 + 1e-5|f|
__device__ __forceinline__ bool near_b(double f, double v) { return fabs(v - f) <= 1e-8 + 1e-5 * fabs(f); }

// Is the crossing lattice edge (q, q+d) used by at least one emitted triangle?  Only reached when
// both end points are within the reference's np.allclose tolerances of the isovalue, where the
// reference may skip whole voxels (border_voxel) or single tetrahedra (tetrahedral.py:576).
__device__ __noinline__ bool edge_used_slow(const cx_params& P, uint32_t i, uint32_t j, uint32_t k, uint32_t d) {
    const float* A = P.grid;
    const uint32_t n1 = P.n1, n2 = P.n2;
    for (uint32_t o = 0; o < 8; o++) {
        if (o & d) continue;  // q is corner o of voxel p = q - o, q+d is corner o|d
        uint32_t oi = (o >> 2) & 1u, oj = (o >> 1) & 1u, ok = o & 1u;
        if (i < oi || j < oj || k < ok) continue;
        uint32_t pi = i - oi, pj = j - oj, pk = k - ok;
        if (pi + 1 >= P.n0 || pj + 1 >= P.n1 || pk + 1 >= P.n2) continue;
        uint32_t near_a = 0, all_b = 1;
        for (uint32_t c = 0; c < 8; c++) {
            double f = (double)A[((size_t)(pi + ((c >> 2) & 1u)) * n1 + (pj + ((c >> 1) & 1u))) * n2 + (pk + (c & 1u))];
            if (fabs(f - P.value) <= P.tol_value) near_a |= 1u << c;
            if (!near_b(f, P.value)) all_b = 0;
        }
        if (all_b) continue;  // not a border voxel: never enumerated
        uint32_t c1 = o, c2 = o | d;
        for (int t = 0; t < 6; t++) {
            uint32_t tm = CX_TETMASK(t);
            if (((tm >> c1) & 1u) && ((tm >> c2) & 1u) && (near_a & tm) != tm) return true;
        }
    }
    return false;
}

// =================================================================================================
// K1 (generic shapes): one lane per lattice cell, 64 consecutive linear indices per wave.
// Classifies the cell, interpolates the edge crossings it owns, reserves output space with one
// atomic per wave and counter, writes vertex records, the per-cell crossing mask, per-row vertex
// bases and one record per active cell for the triangle kernel.
// =================================================================================================
__global__ __launch_bounds__(256) void cx_k_classify_generic(const cx_params P) {
    const uint32_t lane = lane_id();
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t wave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * waves_per_block;
    const uint32_t N = P.nsamples;
    const uint32_t plane = P.n1 * P.n2;
    const float* __restrict__ A = P.grid;

    for (uint32_t base = wave * 64u; base < N; base += nwaves * 64u) {
        const uint32_t lin = base + lane;
        const bool in = lin < N;
        const uint32_t linc = in ? lin : (N - 1);
        const uint32_t i = cx_div(linc, P.div_plane);
        const uint32_t r = linc - i * plane;
        const uint32_t j = cx_div(r, P.div_row);
        const uint32_t k = r - j * P.n2;
        // corner validity and clamped offsets
        const bool vi = (i + 1 < P.n0), vj = (j + 1 < P.n1), vk = (k + 1 < P.n2);
        const uint32_t oi = vi ? plane : 0u, oj = vj ? P.n2 : 0u, ok = vk ? 1u : 0u;
        float f[8];
        f[0] = A[linc];
        f[1] = A[linc + ok];
        f[2] = A[linc + oj];
        f[3] = A[linc + oj + ok];
        f[4] = A[linc + oi];
        f[5] = A[linc + oi + ok];
        f[6] = A[linc + oi + oj];
        f[7] = A[linc + oi + oj + ok];
        uint32_t vm = 1u | (vk ? 2u : 0u) | (vj ? 4u : 0u) | ((vj && vk) ? 8u : 0u);
        vm |= vi ? (vm << 4) : 0u;
        uint32_t sm = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) sm |= (f[c] < P.vcmp) ? (1u << c) : 0u;
        const uint32_t smv = sm & vm;
        const bool active = in && (smv != 0u) && (smv != vm);
        const uint64_t act = __ballot(active);
        if (act == 0) continue;  // wave-uniform: nothing crosses in these 64 cells

        uint32_t emask = 0, ntri = 0, tetskip = 0, border = 0;
        if (active) {
            // crossing mask of the 7 owned edges: corner d valid and on the other side than corner 0
            const uint32_t s0 = (sm & 1u) ? 0xFFu : 0u;
            emask = ((sm ^ s0) & vm) & 0xFEu;
            // tolerance masks in float64, exactly as the reference evaluates them
            uint32_t near_a = 0, nb = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const double fc = (double)f[c];
                near_a |= (fabs(fc - P.value) <= P.tol_value) ? (1u << c) : 0u;
                nb |= near_b(fc, P.value) ? (1u << c) : 0u;
            }
            const bool real_voxel = (vm == 0xFFu);
            if (real_voxel) {
                if (nb == 0xFFu) {
                    tetskip = 0x3Fu;  // border_voxel() false: np.allclose(value, function_values)
                } else {
                    border = 1;
                    if (near_a == 0) {
                        ntri = cx_d_voxel_ntri[sm];
                    } else {
                        for (int t = 0; t < 6; t++) {
                            const uint32_t tm = CX_TETMASK(t);
                            if ((near_a & tm) == tm) tetskip |= 1u << t;
                            else ntri += tet_ntri(tet_pattern(sm, t));
                        }
                    }
                }
            } else {
                tetskip = 0x3Fu;  // no voxel here (upper array boundary): the cell only owns edges
            }
            // drop owned crossings that no emitted triangle uses (tolerance skips around them)
            if (emask && ((near_a & 1u) || (nb & 1u))) {
                for (uint32_t d = 1; d < 8; d++) {
                    if (!((emask >> d) & 1u)) continue;
                    const bool suspicious = (((near_a >> d) & near_a & 1u) | ((nb >> d) & nb & 1u)) != 0u;
                    if (suspicious && !edge_used_slow(P, i, j, k, d)) emask &= ~(1u << d);
                }
            }
        }
        const uint32_t nv = __popc(emask);
        const bool rec = active && (nv != 0u || ntri != 0u);
        uint32_t vtot, ttot;
        const uint32_t vpre = wave_prefix_small<3>(nv, vtot);
        const uint32_t tpre = wave_prefix_small<4>(ntri, ttot);
        const uint64_t recm = __ballot(rec);
        const uint32_t cpre = mbcnt(recm);
        const uint32_t ctot = (uint32_t)__popcll(recm);
        const uint32_t btot = (uint32_t)__popcll(__ballot(border != 0));
        uint32_t vbase = 0, tbase = 0, cbase = 0;
        if (lane == 0) {
            if (vtot) vbase = atomicAdd(&P.counters[CX_CNT_VERTS], vtot);
            if (ttot) tbase = atomicAdd(&P.counters[CX_CNT_TRIS], ttot);
            if (ctot) cbase = atomicAdd(&P.counters[CX_CNT_CELLS], ctot);
            if (btot) atomicAdd(&P.counters[CX_CNT_BORDER], btot);
        }
        vbase = __builtin_amdgcn_readfirstlane(vbase);
        tbase = __builtin_amdgcn_readfirstlane(tbase);
        cbase = __builtin_amdgcn_readfirstlane(cbase);
        const uint32_t vfirst = vbase + vpre;

        // per-cell crossing masks and per-row bases for every 8-cell row that owns a vertex
        const uint64_t owners = __ballot(nv != 0u);
        const uint32_t rowbits = (uint32_t)(owners >> (lane & ~7u)) & 0xFFu;
        if (in && rowbits) {
            P.emask8[lin] = (uint8_t)emask;
            if ((lane & 7u) == 0u) P.rowbase[lin >> 3] = vfirst;
        }
        if (nv && vbase + vtot <= P.vcap) {
            const float fi = (float)i, fj = (float)j, fk = (float)k;
            uint32_t slot = vfirst;
#pragma unroll
            for (uint32_t d = 1; d < 8; d++) {
                if ((emask >> d) & 1u) {
                    // fraction from the owning lattice point: (v - f(q)) / (f(q+d) - f(q)); the
                    // reference interpolates from the low end with ratio=(v-flow)/(fhigh-flow),
                    // or 0.5 when |fhigh-flow| <= 1e-8 -- identical in exact arithmetic.
                    const double den = (double)f[d] - (double)f[0];
                    float t = 0.5f;
                    if (fabs(den) > 1e-8) t = (float)(P.value - (double)f[0]) / (float)den;
                    float4 rec4;
                    rec4.x = (d & 4u) ? fi + t : fi;
                    rec4.y = (d & 2u) ? fj + t : fj;
                    rec4.z = (d & 1u) ? fk + t : fk;
                    rec4.w = __uint_as_float((lin << 3) | d);
                    P.verts[slot++] = rec4;
                }
            }
        }
        if (rec && cbase + ctot <= P.ccap) {
            uint4 c4;
            c4.x = lin;
            c4.y = sm | (tetskip << 8) | (ntri << 16) | (emask << 24);
            c4.z = tbase + tpre;
            c4.w = vfirst;
            P.cells[cbase + cpre] = c4;
        }
    }
}

// ---- CPython 3.10 tuple hash + 8-slot set order, used only with
// CX_DIAG_CPYTHON310 to reproduce the quad diagonal the reference picks (tetrahedral.py:592-595).
__device__ __forceinline__ uint64_t py_tuplehash3(uint32_t x, uint32_t y, uint32_t z) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5;
    acc += (uint64_t)x * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += (uint64_t)y * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += (uint64_t)z * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += 3ULL ^ (P5 ^ 3527539ULL);
    if (acc == ~0ULL) acc = 1546275796ULL;
    return acc;
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

// vertex index of edge ref e = (c1 << 3) | d, given per-corner first-vertex index and crossing mask
struct corner_lut {
    uint32_t vfirst[8];
    uint32_t emask[8];
};

// =================================================================================================
// K2: one lane per active-cell record; expands the 6 tetrahedra into index triples.
// =================================================================================================
__global__ __launch_bounds__(256) void cx_k_emit_triangles(const cx_params P) {
    const uint32_t ncells = min(P.counters[CX_CNT_CELLS], P.ccap);
    const uint32_t ntris_total = P.counters[CX_CNT_TRIS];
    if (ntris_total > P.tcap || P.counters[CX_CNT_VERTS] > P.vcap) return;  // host will re-run with more room
    const uint32_t plane = P.n1 * P.n2;
    const uint64_t* __restrict__ emask64 = reinterpret_cast<const uint64_t*>(P.emask8);
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < ncells; idx += gridDim.x * blockDim.x) {
        const uint4 c4 = P.cells[idx];
        const uint32_t ntri = (c4.y >> 16) & 0xFFu;
        if (ntri == 0) continue;
        const uint32_t lin = c4.x;
        const uint32_t sm = c4.y & 0xFFu, tetskip = (c4.y >> 8) & 0x3Fu;
        // first-vertex index and crossing mask of the 7 corners that can own an edge of this voxel
        uint32_t vfirst[8], em[8];
#pragma unroll
        for (uint32_t c = 0; c < 7; c++) {
            const uint32_t lc = lin + ((c & 4u) ? plane : 0u) + ((c & 2u) ? P.n2 : 0u) + (c & 1u);
            // does corner c own a crossing edge of this voxel?  (some corner c2 > c, c subset of c2, other side)
            const uint32_t sc = ((sm >> c) & 1u) ? 0xFFu : 0u;
            uint32_t sup = 0;  // corners that are strict supersets of c
#pragma unroll
            for (uint32_t c2 = 0; c2 < 8; c2++) sup |= ((c2 & c) == c && c2 != c) ? (1u << c2) : 0u;
            if (((sm ^ sc) & sup) == 0u) { vfirst[c] = 0; em[c] = 0; continue; }
            const uint64_t m64 = emask64[lc >> 3];
            const uint32_t sh = (lc & 7u) * 8u;
            em[c] = (uint32_t)(m64 >> sh) & 0xFFu;
            vfirst[c] = P.rowbase[lc >> 3] + (uint32_t)__popcll(m64 & ((1ULL << sh) - 1ULL));
        }
        vfirst[7] = 0; em[7] = 0;
        uint32_t ci = 0, cj = 0, ck = 0;
        const bool emulate = (P.flags & CX_DIAG_CPYTHON310) != 0u;
        if (emulate) {
            ci = cx_div(lin, P.div_plane);
            const uint32_t r = lin - ci * plane;
            cj = cx_div(r, P.div_row);
            ck = r - cj * P.n2;
        }
        int32_t* out = P.tris + (size_t)c4.z * 3u;
        for (int t = 0; t < 6; t++) {
            if ((tetskip >> t) & 1u) continue;
            const uint32_t pat = tet_pattern(sm, t);
            uint32_t variant = 0;
            if (emulate && __popc(pat) == 2) {
                // low set and high set, each in insertion (tet vertex) order
                uint64_t hl[2], hh[2];
                int nl = 0, nh = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const uint32_t c = cx_d_tet_corners[t][m];
                    const uint64_t h = py_tuplehash3(ci + ((c >> 2) & 1u), cj + ((c >> 1) & 1u), ck + (c & 1u));
                    if ((pat >> m) & 1u) { if (nl < 2) hl[nl] = h; nl++; }
                    else { if (nh < 2) hh[nh] = h; nh++; }
                }
                variant = (py_set2_swapped(hl[0], hl[1]) != py_set2_swapped(hh[0], hh[1])) ? 1u : 0u;
            }
            const uint64_t e = cx_d_tet_tris[t][pat][variant];
            const uint32_t n = (uint32_t)(e >> 36) & 3u;
            for (uint32_t q = 0; q < n; q++) {
                const uint32_t tri = (uint32_t)(e >> (18u * q)) & 0x3FFFFu;
#pragma unroll
                for (uint32_t s = 0; s < 3; s++) {
                    const uint32_t ref = (tri >> (6u * s)) & 0x3Fu;
                    const uint32_t c1 = ref >> 3, d = ref & 7u;
                    // runtime-indexed small arrays: select with a compare chain to stay in registers
                    uint32_t vf = 0, m = 0;
#pragma unroll
                    for (uint32_t c = 0; c < 7; c++) { vf = (c1 == c) ? vfirst[c] : vf; m = (c1 == c) ? em[c] : m; }
                    *out++ = (int32_t)(vf + __popc(m & ((1u << d) - 1u)));
                }
            }
        }
    }
}

void cx_launch_classify_generic(const cx_params& P, hipStream_t s) {
    const uint32_t chunks = (P.nsamples + 63u) / 64u;          // 64-cell wave chunks
    uint32_t blocks = (chunks + 3u) / 4u;
    const uint32_t cap = 256u * 16u;                          // 16 blocks per CU, grid-stride beyond
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(cx_k_classify_generic, dim3(blocks), dim3(256), 0, s, P);
}

void cx_launch_emit_triangles(const cx_params& P, hipStream_t s) {
    hipLaunchKernelGGL(cx_k_emit_triangles, dim3(256u * 8u), dim3(256), 0, s, P);
}
