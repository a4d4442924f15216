// cx_cell.h -- per-cell device logic of the marching-tetrahedra march, shared by the classify kernels.
//
// Reference semantics restated (paths relative to the reference checkout, contourist/...):
//   border_voxel                      tetrahedral.py:383-394
//   enumerate_tetrahedron_triangles   tetrahedral.py:561-595
//   contour_pair_interpolation        tetrahedral.py:471-487
#pragma once
#include "cx_common.h"

// corner masks of the 6 tetrahedra: bit c set <=> cube corner c is a vertex of tet t
#define CX_TETMASK(t) (uint32_t)((1u << cx_d_tet_corners[t][0]) | (1u << cx_d_tet_corners[t][1]) | \
                                  (1u << cx_d_tet_corners[t][2]) | (1u << cx_d_tet_corners[t][3]))

__device__ __forceinline__ uint32_t cx_lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t cx_mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// exclusive prefix sum over the wave of a small per-lane count (< 2^NBITS), plus the wave total,
// from NBITS ballots (no LDS, no cross-lane data movement).
template <int NBITS>
__device__ __forceinline__ uint32_t cx_wave_prefix_small(uint32_t x, uint32_t& total) {
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int b = 0; b < NBITS; b++) {
        const uint64_t m = __ballot((x >> b) & 1u);
        pre += cx_mbcnt(m) << b;
        tot += (uint32_t)__popcll(m) << b;
    }
    total = tot;
    return pre;
}

// pattern of tet t inside a voxel sign mask: bit m set <=> tet vertex m is low
__device__ __forceinline__ uint32_t cx_tet_pattern(uint32_t sm, int t) {
    return ((sm >> cx_d_tet_corners[t][0]) & 1u) | (((sm >> cx_d_tet_corners[t][1]) & 1u) << 1) |
           (((sm >> cx_d_tet_corners[t][2]) & 1u) << 2) | (((sm >> cx_d_tet_corners[t][3]) & 1u) << 3);
}

// tet vertex m of tet t -> cube corner as compile-time constants (same data as cx_d_tet_corners)
__device__ constexpr uint8_t CX_TCC[6][4] = CX_TET_CORNERS_INIT;
// triangles a voxel with sign mask sm emits when no tolerance skip applies: per tetrahedron the number
// of low corners n -> n odd: 1, n == 2: 2   (pure ALU; a per-lane table lookup would be a memory access)
__device__ __forceinline__ uint32_t cx_voxel_ntri(uint32_t sm) {
    uint32_t nt = 0;
#pragma unroll
    for (int t = 0; t < 6; t++) {
        const uint32_t n = ((sm >> CX_TCC[t][0]) & 1u) + ((sm >> CX_TCC[t][1]) & 1u) + ((sm >> CX_TCC[t][2]) & 1u) +
                           ((sm >> CX_TCC[t][3]) & 1u);
        nt += (n == 2u) ? 2u : (n & 1u);
    }
    return nt;
}

__device__ __forceinline__ uint32_t cx_tet_ntri(uint32_t pattern) {
    const uint32_t n = __popc(pattern);
    return (n == 2) ? 2u : ((n == 1 || n == 3) ? 1u : 0u);
}

// np.allclose(value, f) for one sample (border_voxel, tetrahedral.py:391): |v-f| <= 1e-8 + 1e-5|f|
__device__ __forceinline__ bool cx_near_b(double f, double v) { return fabs(v - f) <= 1e-8 + 1e-5 * fabs(f); }

// Is the crossing lattice edge (q, q+d) used by at least one emitted triangle?  Only reached when
// both end points are within the reference's np.allclose tolerances of the isovalue, where the
// reference may skip whole voxels (border_voxel) or single tetrahedra (tetrahedral.py:576).
static __device__ __forceinline__ bool cx_edge_used_slow(const float* A, uint32_t n0, uint32_t n1, uint32_t n2,
                                                      double value, double tol_value, uint32_t i, uint32_t j,
                                                      uint32_t k, uint32_t d) {
    for (uint32_t o = 0; o < 8; o++) {
        if (o & d) continue;  // q is corner o of voxel p = q - o, q+d is corner o|d
        const uint32_t oi = (o >> 2) & 1u, oj = (o >> 1) & 1u, ok = o & 1u;
        if (i < oi || j < oj || k < ok) continue;
        const uint32_t pi = i - oi, pj = j - oj, pk = k - ok;
        if (pi + 1 >= n0 || pj + 1 >= n1 || pk + 1 >= n2) continue;
        uint32_t near_a = 0, all_b = 1;
        for (uint32_t c = 0; c < 8; c++) {
            const double f = (double)A[((size_t)(pi + ((c >> 2) & 1u)) * n1 + (pj + ((c >> 1) & 1u))) * n2 + (pk + (c & 1u))];
            if (fabs(f - value) <= tol_value) near_a |= 1u << c;
            if (!cx_near_b(f, value)) all_b = 0;
        }
        if (all_b) continue;  // not a border voxel: never enumerated
        const uint32_t c1 = o, c2 = o | d;
        for (int t = 0; t < 6; t++) {
            const uint32_t tm = CX_TETMASK(t);
            if (((tm >> c1) & 1u) && ((tm >> c2) & 1u) && (near_a & tm) != tm) return true;
        }
    }
    return false;
}

struct cx_cell_info {
    uint32_t sm;       // bit c: f(corner c) < value   (clamped corners repeat their source)
    uint32_t emask;    // bit d (1..7): owned edge q->q+d crosses and is used by an emitted triangle
    uint32_t ntri;     // triangles this voxel emits
    uint32_t tetskip;  // bit t: tetrahedron t emits nothing because of the reference's tolerances
    uint32_t border;   // 1 if this is a voxel with a sign change that border_voxel() accepts
};

// load the 8 corners of cell (i,j,k), out-of-array corners clamped onto the array; vm = validity mask.
// The (k, k+1) pair of each of the 4 (i,j) rows is ONE 8-byte load (4-byte aligned): half the
// address-processing work of 8 scalar loads in a phase that is bound by scattered accesses.
struct __attribute__((packed, aligned(4))) cx_f2 {
    float x, y;
};
__device__ __forceinline__ uint32_t cx_load_corners(const cx_params& P, uint32_t lin, uint32_t i, uint32_t j,
                                                    uint32_t k, float f[8]) {
    const float* __restrict__ A = P.grid;
    const uint32_t plane = P.n1 * P.n2;
    const bool vi = (i + 1 < P.n0), vj = (j + 1 < P.n1), vk = (k + 1 < P.n2);
    const uint32_t oi = vi ? plane : 0u, oj = vj ? P.n2 : 0u;
    // at the array edge in k read the pair (k-1, k) instead and repeat k
    const uint32_t base = vk ? lin : lin - 1u;
    const cx_f2 p0 = *reinterpret_cast<const cx_f2*>(A + base);
    const cx_f2 p1 = *reinterpret_cast<const cx_f2*>(A + base + oj);
    const cx_f2 p2 = *reinterpret_cast<const cx_f2*>(A + base + oi);
    const cx_f2 p3 = *reinterpret_cast<const cx_f2*>(A + base + oi + oj);
    f[0] = vk ? p0.x : p0.y; f[1] = p0.y;
    f[2] = vk ? p1.x : p1.y; f[3] = p1.y;
    f[4] = vk ? p2.x : p2.y; f[5] = p2.y;
    f[6] = vk ? p3.x : p3.y; f[7] = p3.y;
    uint32_t vm = 1u | (vk ? 2u : 0u) | (vj ? 4u : 0u) | ((vj && vk) ? 8u : 0u);
    vm |= vi ? (vm << 4) : 0u;
    return vm;
}

__device__ __forceinline__ uint32_t cx_sign_mask(const cx_params& P, const float f[8]) {
    uint32_t sm = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) sm |= (f[c] < P.vcmp) ? (1u << c) : 0u;
    return sm;
}

// classification of one ACTIVE cell (sign change among its valid corners)
__device__ __forceinline__ cx_cell_info cx_classify_cell(const cx_params& P, const float f[8], uint32_t vm,
                                                         uint32_t sm, uint32_t i, uint32_t j, uint32_t k) {
    cx_cell_info R;
    R.sm = sm;
    R.ntri = 0;
    R.tetskip = 0;
    R.border = 0;
    // crossing mask of the 7 owned edges: corner d valid and on the other side than corner 0
    const uint32_t s0 = (sm & 1u) ? 0xFFu : 0u;
    R.emask = ((sm ^ s0) & vm) & 0xFEu;
    // cheap fp32 screen for the reference's np.allclose tolerances (a superset of both float64 tests)
    float dmin = fabsf(f[0] - P.vcmp);
#pragma unroll
    for (int c = 1; c < 8; c++) dmin = fminf(dmin, fabsf(f[c] - P.vcmp));
    const bool any_near = dmin <= P.near_abs;
    const bool real_voxel = (vm == 0xFFu);
    if (!any_near) {
        if (real_voxel) {
            R.border = 1;
            R.ntri = cx_voxel_ntri(sm);
        } else {
            R.tetskip = 0x3Fu;  // no voxel here (upper array boundary): the cell only owns edges
        }
        return R;
    }
    // tolerance masks in float64, exactly as the reference evaluates them
    uint32_t near_a = 0, nb = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const double fc = (double)f[c];
        near_a |= (fabs(fc - P.value) <= P.tol_value) ? (1u << c) : 0u;
        nb |= cx_near_b(fc, P.value) ? (1u << c) : 0u;
    }
    if (real_voxel) {
        if (nb == 0xFFu) {
            R.tetskip = 0x3Fu;  // border_voxel() false: np.allclose(value, function_values)
        } else {
            R.border = 1;
            for (int t = 0; t < 6; t++) {
                const uint32_t tm = CX_TETMASK(t);
                if ((near_a & tm) == tm) R.tetskip |= 1u << t;
                else R.ntri += cx_tet_ntri(cx_tet_pattern(sm, t));
            }
        }
    } else {
        R.tetskip = 0x3Fu;
    }
    // drop owned crossings that no emitted triangle uses (tolerance skips around them)
    if (R.emask && ((near_a & 1u) || (nb & 1u))) {
        for (uint32_t d = 1; d < 8; d++) {
            if (!((R.emask >> d) & 1u)) continue;
            const bool suspicious = (((near_a >> d) & near_a & 1u) | ((nb >> d) & nb & 1u)) != 0u;
            if (suspicious && !cx_edge_used_slow(P.grid, P.n0, P.n1, P.n2, P.value, P.tol_value, i, j, k, d))
                R.emask &= ~(1u << d);
        }
    }
    return R;
}

// vertex records of one cell: calls sink(r, record) for the r-th owned crossing (ascending direction d)
// THE fraction of a crossing along its lattice edge, (v - f(q)) / (f(q+d) - f(q)), in fp32: reciprocal and product (v_rcp_f32 is
// good to one unit in the last place, so the fraction is good to ~1.5: 2e-7 of a voxel, a fifth of the 1e-6 the coordinates are
// held to).  ONE definition for the stream kernel, the vertex stage's gather path, the per-cell path and the fused kernel, so that
// whichever of them interpolates a crossing writes the same bits.  (Until round 4: __fdividef, which hipcc expands to the IEEE
// division sequence -- ~11 instructions where this takes 2; Level 1 recomputes every point in float64 from the grid anyway.)
__device__ __forceinline__ float cx_fraction(float num, float den) { return num * __builtin_amdgcn_rcpf(den); }

template <typename Sink>
__device__ __forceinline__ void cx_emit_vertices(const cx_params& P, const float f[8], uint32_t emask, uint32_t lin,
                                                 uint32_t i, uint32_t j, uint32_t k, Sink sink) {
    (void)i; (void)j; (void)k;
    // v - f(q) with the isovalue carried as two floats: exact to fp32 rounding of the result
    const float num = (P.vhi - f[0]) + P.vlo;
    uint32_t r = 0;
#pragma unroll
    for (uint32_t d = 1; d < 8; d++) {
        if ((emask >> d) & 1u) {
            // fraction from the owning lattice point: (v - f(q)) / (f(q+d) - f(q)); the reference
            // interpolates from the low end with ratio=(v-flow)/(fhigh-flow), or 0.5 when
            // |fhigh-flow| <= 1e-8 (tetrahedral.py:483-487) -- identical in exact arithmetic.
            const float den = f[d] - f[0];
            float t = cx_fraction(num, den);
            if (fabsf(den) <= 1.001e-8f) {   // rare: decide the reference's |den| <= 1e-8 test in float64
                const double dd = (double)f[d] - (double)f[0];
                t = (fabs(dd) <= 1e-8) ? 0.5f : (float)((P.value - (double)f[0]) / dd);
            }
            sink(r++, make_uint2((lin << 3) | d, __float_as_uint(t)));
        }
    }
}
