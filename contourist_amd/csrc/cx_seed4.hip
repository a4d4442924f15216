// cx_seed4.hip -- seeded selection of level-set components for the 4-D march (C ABI: cx_select_seeded4d).
//
// The 4-D contour maker of the reference inherits its search from the 3-D one (contourist/tetrahedral.py) and runs it
// with the 80 neighbour offsets of pentatopes.py:32-39:
//   find_initial_voxels  tetrahedral.py:396-441  each end point pair is bisected until adjacent; for both points: the
//                                                point's own hyper-voxel if it is a border voxel, else the first border
//                                                voxel among its 80 neighbours (OFFSETS4D order); shared `visited` set
//   expand_voxels        :443-463                breadth-first over the 80 neighbours, border voxels inside the grid only
//   border_voxel         :383-394                min <= value <= max over the 16 corners and not np.allclose(value, corners)
// Here, as in cx_seed.hip: the dense march has produced every hyper-voxel with a sign change (cell records); they are
// grouped by 80-connectivity with a lock-free union-find, the groups that contain a seed voxel are kept and the
// tetrahedra of all other hyper-voxels are masked out for cx_postprocess4d.  A tetrahedron belongs to the hyper-voxel
// whose corner 0 is the componentwise minimum of the owners of its four edges (every tetrahedron touches all five
// corners of its pentatope, every pentatope contains corner 0 of its hypercube).
// Deviations: hyper-voxels whose corners only touch the isovalue (no strict sign change) do not bridge groups; seed
// points whose hyper-voxel is not inside the ARRAY are skipped (the array may carry a rim of samples around the reference's
// grid: cx_select_seeded4d_ex takes the grid's box, and seed voxels in the rim are kept as the reference keeps them).
#include <algorithm>
#include <cstring>
#include <string>

#include "cx_state4.h"

extern "C" int cx_select_seeded4d_ex(cx_ctx* ctx, const int32_t* endpoints_ijkl, int64_t n, const int32_t* range_lo_hi, uint32_t flags, int64_t* out_counts);

#define CXS4_HIP(ctx, call)                                                                      \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return (e__ == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;                      \
        }                                                                                        \
    } while (0)

#define CXS4_NONE 0xFFFFFFFFu

struct cxs4_grid {
    const float* A;
    uint32_t n[4];
    double value;
    int lo[4], hi[4];   // in_range box of the breadth-first growth: lo <= hyper-voxel < hi (default 0 .. n-1)
};
__device__ __forceinline__ bool cxs4_in_range(const cxs4_grid& G, const int p[4]) {
    for (int a = 0; a < 4; a++)
        if (p[a] < G.lo[a] || p[a] >= G.hi[a]) return false;
    return true;
}
__device__ __forceinline__ uint32_t cxs4_lin(const cxs4_grid& G, const int p[4]) {
    return (((uint32_t)p[0] * G.n[1] + (uint32_t)p[1]) * G.n[2] + (uint32_t)p[2]) * G.n[3] + (uint32_t)p[3];
}
__device__ __forceinline__ void cxs4_unravel(const cxs4_grid& G, uint32_t lin, int p[4]) {
    p[3] = (int)(lin % G.n[3]); lin /= G.n[3];
    p[2] = (int)(lin % G.n[2]); lin /= G.n[2];
    p[1] = (int)(lin % G.n[1]);
    p[0] = (int)(lin / G.n[1]);
}
__device__ __forceinline__ bool cxs4_voxel_inside(const cxs4_grid& G, const int p[4]) {
    for (int a = 0; a < 4; a++)
        if (p[a] < 0 || p[a] + 1 >= (int)G.n[a]) return false;
    return true;
}
// a record of a real hyper-voxel with a strict sign change among its 16 corners
__device__ __forceinline__ bool cxs4_is_voxel_record(const cxs4_grid& G, const uint4& c) {
    const uint32_t sm = c.y & 0xFFFFu;
    if (sm == 0u || sm == 0xFFFFu) return false;
    int p[4];
    cxs4_unravel(G, c.x, p);
    return cxs4_voxel_inside(G, p);
}
__device__ __forceinline__ uint32_t cxs4_find(uint32_t* parent, uint32_t x) {
    for (;;) {
        const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        const uint32_t g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != p) atomicCAS(&parent[x], p, g);   // path halving
        x = p;
    }
}
__device__ __forceinline__ void cxs4_union(uint32_t* parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = cxs4_find(parent, a);
        b = cxs4_find(parent, b);
        if (a == b) return;
        const uint32_t win = min(a, b), lose = max(a, b);
        if (atomicCAS(&parent[lose], lose, win) == lose) return;
    }
}

__global__ void cxs4_k_map(const uint4* cells, uint32_t ncells, uint32_t* vmap, uint32_t* parent, cxs4_grid G) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ncells) return;
    parent[r] = r;
    const uint4 c = cells[r];
    if (cxs4_is_voxel_record(G, c)) vmap[c.x] = r;
}
// record index of the hyper-voxel at p, or CXS4_NONE (the map is not cleared: entries validate themselves)
__device__ __forceinline__ uint32_t cxs4_lookup(const cxs4_grid& G, const uint4* cells, uint32_t ncells, const uint32_t* vmap, const int p[4]) {
    if (!cxs4_voxel_inside(G, p)) return CXS4_NONE;
    const uint32_t lin = cxs4_lin(G, p);
    const uint32_t r = vmap[lin];
    if (r >= ncells) return CXS4_NONE;
    const uint4 c = cells[r];
    return (c.x == lin && cxs4_is_voxel_record(G, c)) ? r : CXS4_NONE;
}
// The unions in two steps, as cx_seed.hip does them for the 3-D march: a workgroup unites CXS4_UB consecutive records among themselves in
// an LDS forest and writes it into the global parent words with plain stores (cxs4_k_union_block); only pairs that straddle two blocks
// go through the device-scope union-find (cxs4_k_union_far).  (One step: 2.7 ms for the 0.87 M hyper-voxels of config 4.)
#define CXS4_UB 1024u
__device__ __forceinline__ uint32_t cxs4_lfind(uint32_t* lp, uint32_t x) {
    for (;;) {
        const uint32_t p = lp[x];
        if (p == x) return x;
        const uint32_t g = lp[p];
        if (g != p) atomicCAS(&lp[x], p, g);   // path halving
        x = p;
    }
}
// f(record of the neighbour) for the 40 "forward" neighbours of hyper-voxel p that are surface hyper-voxels inside the range (the
// other 40 are reached from the other side)
template <typename F>
__device__ __forceinline__ void cxs4_forward_neighbours(const cxs4_grid& G, const uint4* cells, uint32_t ncells, const uint32_t* vmap, const int p[4], F f) {
    for (int code = 41; code < 81; code++) {   // offsets in lexicographic order, (0,0,0,0) is code 40
        int o[4], x = code;
        o[3] = x % 3 - 1; x /= 3;
        o[2] = x % 3 - 1; x /= 3;
        o[1] = x % 3 - 1; x /= 3;
        o[0] = x - 1;
        const int q[4] = {p[0] + o[0], p[1] + o[1], p[2] + o[2], p[3] + o[3]};
        if (!cxs4_in_range(G, q)) continue;   // in_range (tetrahedral.py:465-469)
        const uint32_t other = cxs4_lookup(G, cells, ncells, vmap, q);
        if (other != CXS4_NONE) f(other);
    }
}
__global__ __launch_bounds__(256) void cxs4_k_union_block(const uint4* cells, uint32_t ncells, const uint32_t* vmap, uint32_t* parent, cxs4_grid G) {
    __shared__ uint32_t lp[CXS4_UB];
    const uint32_t b0 = blockIdx.x * CXS4_UB;
    for (uint32_t x = threadIdx.x; x < CXS4_UB; x += 256u) lp[x] = x;
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < CXS4_UB; x += 256u) {
        const uint32_t r = b0 + x;
        if (r >= ncells) continue;
        const uint4 c = cells[r];
        if (!cxs4_is_voxel_record(G, c)) continue;
        int p[4];
        cxs4_unravel(G, c.x, p);
        if (!cxs4_in_range(G, p)) continue;   // only in-range hyper-voxels grow (seed voxels outside the box: cxs4_k_mark)
        cxs4_forward_neighbours(G, cells, ncells, vmap, p, [&](uint32_t other) {
            const uint32_t o = other - b0;
            if (o >= CXS4_UB) return;                             // another block's record: cxs4_k_union_far
            uint32_t a = x, b = o;
            for (;;) {
                a = cxs4_lfind(lp, a);
                b = cxs4_lfind(lp, b);
                if (a == b) break;
                const uint32_t win = min(a, b), lose = max(a, b);
                if (atomicCAS(&lp[lose], lose, win) == lose) break;
            }
        });
    }
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < CXS4_UB; x += 256u) {
        if (b0 + x >= ncells) continue;
        const uint32_t root = cxs4_lfind(lp, x);
        if (root != x) parent[b0 + x] = b0 + root;      // (nobody else touches these words in this kernel); roots = smallest ids
    }
}
__global__ void cxs4_k_union_far(const uint4* cells, uint32_t ncells, const uint32_t* vmap, uint32_t* parent, cxs4_grid G) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ncells) return;
    const uint4 c = cells[r];
    if (!cxs4_is_voxel_record(G, c)) return;
    int p[4];
    cxs4_unravel(G, c.x, p);
    if (!cxs4_in_range(G, p)) return;
    const uint32_t b0 = (r / CXS4_UB) * CXS4_UB;
    cxs4_forward_neighbours(G, cells, ncells, vmap, p, [&](uint32_t other) {
        if (other - b0 >= CXS4_UB) cxs4_union(parent, r, other);  // (pairs inside one block are united already)
    });
}
__global__ void cxs4_k_flatten(uint32_t* parent, uint32_t n) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) parent[r] = cxs4_find(parent, r);
}

__device__ bool cxs4_border_voxel(const cxs4_grid& G, const int p[4]) {
    if (!cxs4_voxel_inside(G, p)) return false;
    double lo = 1e300, hi = -1e300;
    bool allclose = true;
    for (int c = 0; c < 16; c++) {
        const int q[4] = {p[0] + ((c >> 3) & 1), p[1] + ((c >> 2) & 1), p[2] + ((c >> 1) & 1), p[3] + (c & 1)};
        const double f = (double)G.A[cxs4_lin(G, q)];
        lo = fmin(lo, f); hi = fmax(hi, f);
        if (!(fabs(G.value - f) <= 1e-8 + 1e-5 * fabs(f))) allclose = false;
    }
    if (allclose) return false;
    return lo <= G.value && hi >= G.value;
}
__device__ bool cxs4_visit(unsigned long long* table, unsigned long long mask, const int p[4]) {   // true: newly added
    const unsigned long long key = ((unsigned long long)(p[0] + 4) << 48) | ((unsigned long long)(p[1] + 4) << 32) |
                                   ((unsigned long long)(p[2] + 4) << 16) | (unsigned long long)(p[3] + 4);
    unsigned long long h = (key * 0x9E3779B97F4A7C15ULL) >> 20;
    for (;;) {
        const unsigned long long cur = table[h & mask];
        if (cur == key + 1ULL) return false;
        if (cur == 0ULL) { table[h & mask] = key + 1ULL; return true; }
        h++;
    }
}
// seeds, sequentially, as the reference runs them (one thread; end point lists are short).
// out[0] = number of seed voxels, out[1] = end point pairs that do not straddle the isovalue or lie outside the grid
__global__ void cxs4_k_seeds(cxs4_grid G, const int32_t* ep, uint32_t n, unsigned long long* visited, unsigned long long vmask,
                             uint32_t* seeds, uint32_t* out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t ns = 0, bad = 0;
    for (uint32_t s = 0; s < n; s++) {
        int lowp[4], highp[4];
        bool okp = true;
        for (int a = 0; a < 4; a++) {
            lowp[a] = ep[s * 8 + a];
            highp[a] = ep[s * 8 + 4 + a];
            if (lowp[a] < 0 || highp[a] < 0 || lowp[a] >= (int)G.n[a] || highp[a] >= (int)G.n[a]) okp = false;
        }
        if (!okp) { bad++; continue; }
        double lowv = (double)G.A[cxs4_lin(G, lowp)], highv = (double)G.A[cxs4_lin(G, highp)];
        if (lowv > G.value || highv < G.value) {
            for (int a = 0; a < 4; a++) { const int t = lowp[a]; lowp[a] = highp[a]; highp[a] = t; }
            const double t = lowv; lowv = highv; highv = t;
        }
        if (!(lowv <= G.value && highv >= G.value)) { bad++; continue; }   // the reference asserts here (:412-414)
        for (;;) {
            bool far = false;
            for (int a = 0; a < 4; a++) far = far || abs(lowp[a] - highp[a]) > 1;
            if (!far) break;
            int mid[4];
            for (int a = 0; a < 4; a++) mid[a] = (lowp[a] + highp[a]) / 2;   // both non-negative: Python floor division
            if ((double)G.A[cxs4_lin(G, mid)] < G.value) { for (int a = 0; a < 4; a++) lowp[a] = mid[a]; }
            else { for (int a = 0; a < 4; a++) highp[a] = mid[a]; }
        }
        for (int which = 0; which < 2; which++) {
            const int* p = which ? highp : lowp;
            if (!cxs4_visit(visited, vmask, p)) continue;
            if (cxs4_border_voxel(G, p)) { seeds[ns++] = cxs4_lin(G, p); continue; }
            bool found = false;
            for (int code = 0; code < 81 && !found; code++) {
                if (code == 40) continue;
                int o[4], x = code;
                o[3] = x % 3 - 1; x /= 3;
                o[2] = x % 3 - 1; x /= 3;
                o[1] = x % 3 - 1; x /= 3;
                o[0] = x - 1;
                const int q[4] = {p[0] + o[0], p[1] + o[1], p[2] + o[2], p[3] + o[3]};
                bool neg = false;
                for (int a = 0; a < 4; a++) neg = neg || q[a] < 0 || q[a] >= (int)G.n[a];
                if (neg) continue;   // not inside the array: cannot be a border voxel here, and not worth a table entry
                if (!cxs4_visit(visited, vmask, q)) continue;
                if (cxs4_border_voxel(G, q)) { seeds[ns++] = cxs4_lin(G, q); found = true; }
            }
        }
    }
    out[0] = ns;
    out[1] = bad;
}
// many end point pairs (the exhaustive search on a large open surface, the coarse crossing search): one thread per pair, no
// shared `visited` set -- each end point yields its own hyper-voxel or its first border neighbour (OFFSETS4D order).  (The
// reference's shared set only changes which of several adjacent candidates gets picked when pairs collide.)
// Slots 2s, 2s+1; CXS4_NONE = none.
__global__ void cxs4_k_seeds_parallel(cxs4_grid G, const int32_t* ep, uint32_t n, uint32_t* seeds, uint32_t* out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    seeds[2 * s] = seeds[2 * s + 1] = CXS4_NONE;
    if (s == 0) out[0] = 2u * n;   // slots to look at
    int lowp[4], highp[4];
    for (int a = 0; a < 4; a++) {
        lowp[a] = ep[s * 8 + a];
        highp[a] = ep[s * 8 + 4 + a];
        if (lowp[a] < 0 || highp[a] < 0 || lowp[a] >= (int)G.n[a] || highp[a] >= (int)G.n[a]) { atomicAdd(&out[1], 1u); return; }
    }
    double lowv = (double)G.A[cxs4_lin(G, lowp)], highv = (double)G.A[cxs4_lin(G, highp)];
    if (lowv > G.value || highv < G.value) {
        for (int a = 0; a < 4; a++) { const int t = lowp[a]; lowp[a] = highp[a]; highp[a] = t; }
        const double t = lowv; lowv = highv; highv = t;
    }
    if (!(lowv <= G.value && highv >= G.value)) { atomicAdd(&out[1], 1u); return; }
    for (;;) {
        bool far = false;
        for (int a = 0; a < 4; a++) far = far || abs(lowp[a] - highp[a]) > 1;
        if (!far) break;
        int mid[4];
        for (int a = 0; a < 4; a++) mid[a] = (lowp[a] + highp[a]) / 2;
        if ((double)G.A[cxs4_lin(G, mid)] < G.value) { for (int a = 0; a < 4; a++) lowp[a] = mid[a]; }
        else { for (int a = 0; a < 4; a++) highp[a] = mid[a]; }
    }
    for (int which = 0; which < 2; which++) {
        const int* p = which ? highp : lowp;
        if (cxs4_border_voxel(G, p)) { seeds[2 * s + which] = cxs4_lin(G, p); continue; }
        bool found = false;
        for (int code = 0; code < 81 && !found; code++) {
            if (code == 40) continue;
            int o[4], x = code;
            o[3] = x % 3 - 1; x /= 3;
            o[2] = x % 3 - 1; x /= 3;
            o[1] = x % 3 - 1; x /= 3;
            o[0] = x - 1;
            const int q[4] = {p[0] + o[0], p[1] + o[1], p[2] + o[2], p[3] + o[3]};
            bool neg = false;
            for (int a = 0; a < 4; a++) neg = neg || q[a] < 0 || q[a] >= (int)G.n[a];
            if (neg) continue;
            if (cxs4_border_voxel(G, q)) { seeds[2 * s + which] = cxs4_lin(G, q); found = true; }
        }
    }
}
// flag[] = groups reached; seedkeep[] = seed voxels themselves: a seed voxel outside the in_range box (the reference does not
// range-check the voxels it starts from, tetrahedral.py:396-441) is kept and grows one step into the box, as the reference's
// first expand_voxels round does
__global__ void cxs4_k_mark(const uint4* cells, uint32_t ncells, const uint32_t* vmap, const uint32_t* parent, const uint32_t* seeds,
                            const uint32_t* nseeds, uint8_t* flag, uint8_t* seedkeep, cxs4_grid G) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseeds[0] || seeds[s] == CXS4_NONE) return;
    int p[4];
    cxs4_unravel(G, seeds[s], p);
    const uint32_t r = cxs4_lookup(G, cells, ncells, vmap, p);
    if (cxs4_in_range(G, p)) {
        if (r != CXS4_NONE) flag[parent[r]] = 1;   // (a border voxel without a strict sign change has no tetrahedra and does not grow here)
        return;
    }
    if (r != CXS4_NONE) seedkeep[r] = 1;
    for (int code = 0; code < 81; code++) {
        if (code == 40) continue;
        int o[4], x = code;
        o[3] = x % 3 - 1; x /= 3;
        o[2] = x % 3 - 1; x /= 3;
        o[1] = x % 3 - 1; x /= 3;
        o[0] = x - 1;
        const int q[4] = {p[0] + o[0], p[1] + o[1], p[2] + o[2], p[3] + o[3]};
        if (!cxs4_in_range(G, q)) continue;
        const uint32_t r2 = cxs4_lookup(G, cells, ncells, vmap, q);
        if (r2 != CXS4_NONE) flag[parent[r2]] = 1;
    }
}
#define CXS4_PARTIAL0 32u       // out[32 + 32 p]: partial sums p = 0 .. CXS4_PARTIALS - 1 of the tetrahedra kept
#define CXS4_PARTIALS 128u
#define CXS4_OUT_WORDS (CXS4_PARTIAL0 + 32u * CXS4_PARTIALS)
// keep[t] = 1 for the tetrahedra of the hyper-voxels in flagged groups; out[2] groups kept, out[3] tetrahedra kept
__global__ void cxs4_k_keep(const uint4* cells, uint32_t ncells, const uint32_t* vmap, const uint32_t* parent, const uint8_t* flag,
                            const uint8_t* seedkeep, const int32_t* tets, const uint32_t* vkeys, uint32_t nt, uint8_t* keep, uint32_t* out, cxs4_grid G, int all_in_range) {
    // (the kept tetrahedra are counted per wave, then per workgroup, and added to one of 128 partial sums in cache lines of their own: one
    // device-scope add per kept tetrahedron on ONE address -- 28 M on config 4, executed one after the other -- took 5.0 of this
    // selection's 8.1 ms)
    __shared__ uint32_t s_kept;
    if (threadIdx.x == 0) s_kept = 0u;
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint8_t k = 0;
    if (t < nt) {
        int b[4] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF};
        for (int q = 0; q < 4; q++) {
            int p[4];
            cxs4_unravel(G, vkeys[(uint32_t)tets[(size_t)t * 4 + q]] >> 4, p);
            for (int a = 0; a < 4; a++) b[a] = min(b[a], p[a]);
        }
        const uint32_t r = cxs4_lookup(G, cells, ncells, vmap, b);
        k = (r != CXS4_NONE && ((cxs4_in_range(G, b) && (all_in_range || flag[parent[r]] != 0)) || seedkeep[r] != 0)) ? 1 : 0;
        keep[t] = k;
    }
    const uint32_t nk = (uint32_t)__popcll(__ballot(k != 0));
    if ((threadIdx.x & 63u) == 0u && nk) atomicAdd(&s_kept, nk);
    __syncthreads();
    if (threadIdx.x == 0 && s_kept) atomicAdd(&out[CXS4_PARTIAL0 + 32u * (blockIdx.x & (CXS4_PARTIALS - 1u))], s_kept);
}
__global__ void cxs4_k_keep_sum(uint32_t* out) {
    __shared__ uint32_t s;
    if (threadIdx.x == 0) s = 0u;
    __syncthreads();
    const uint32_t v = out[CXS4_PARTIAL0 + 32u * threadIdx.x];
    if (v) atomicAdd(&s, v);
    __syncthreads();
    if (threadIdx.x == 0) out[3] += s;
}
__global__ void cxs4_k_count_groups(const uint32_t* parent, uint32_t ncells, const uint8_t* flag, uint32_t* out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < ncells && parent[r] == r && flag[r]) atomicAdd(&out[2], 1u);
}

extern "C" int cx_select_seeded4d(cx_ctx* ctx, const int32_t* endpoints_ijkl, int64_t n, int64_t* out_counts) {
    return cx_select_seeded4d_ex(ctx, endpoints_ijkl, n, nullptr, 0u, out_counts);
}
extern "C" int cx_select_seeded4d_ex(cx_ctx* ctx, const int32_t* endpoints_ijkl, int64_t n, const int32_t* range_lo_hi, uint32_t flags,
                                     int64_t* out_counts) {
    const int all_in_range = (flags & CX_SEED_ALL_IN_RANGE) ? 1 : 0;
    if (!ctx || (n > 0 && !endpoints_ijkl) || n < 0) return CX_ERR_INVALID;
    cx_state4* S4 = ctx->s4;
    if (!S4 || !S4->extracted) { ctx->err = "cx_select_seeded4d: no valid 4-D extraction"; return CX_ERR_STATE; }
    CXS4_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t ncells = (uint32_t)S4->counts.n_cells, nt = (uint32_t)S4->counts.n_triangles;
    cxs4_grid G;
    G.A = S4->grid;
    size_t nsamples = 1;
    for (int a = 0; a < 4; a++) {
        G.n[a] = (uint32_t)S4->n[a]; nsamples *= (size_t)S4->n[a];
        G.lo[a] = range_lo_hi ? std::max(range_lo_hi[a], 0) : 0;
        G.hi[a] = range_lo_hi ? std::min(range_lo_hi[4 + a], (int)S4->n[a] - 1) : (int)S4->n[a] - 1;
    }
    G.value = S4->value;
    {
        const int rcg = cx_grow(ctx, S4->tet_keep, S4->keep_cap, (size_t)nt + 64);
        if (rcg) return rcg;
    }
    S4->keep_valid = false;
    uint32_t *vmap = nullptr, *parent = nullptr, *seeds = nullptr, *out = nullptr;
    uint8_t* flag = nullptr;
    int32_t* ep = nullptr;
    unsigned long long* visited = nullptr;
    unsigned long long vsize = 1024;
    // beyond this many pairs (or on request): one thread per pair, no shared visited set (cx_seeded_mode tells which ran)
    const int64_t sequential_max = (flags & CX_SEED_PARALLEL) ? -1 : 16384;
    while (n <= sequential_max && vsize < (unsigned long long)n * 164ULL * 4ULL) vsize <<= 1;
    ctx->seed_mode = (n <= sequential_max) ? 0 : 1;
    int rc = CX_OK;
    uint32_t host_out[4] = {0, 0, 0, 0};
    do {
        hipError_t e;
#define CXS4_TRY(call) if ((e = (call)) != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e); rc = (e == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP; break; }
        // (scratch kept in the context between calls, shared with cx_select_seeded3d_ex: cx_grow only ever grows)
#define CXS4_GRAB(slot, ptr, bytes) { if ((rc = cx_grow(ctx, ctx->seed_buf[slot], ctx->seed_cap[slot], (size_t)(bytes)))) break; ptr = reinterpret_cast<decltype(ptr)>(ctx->seed_buf[slot]); }
        CXS4_GRAB(0, vmap, (nsamples + 64) * sizeof(uint32_t));
        CXS4_GRAB(1, parent, ((size_t)ncells + 64) * sizeof(uint32_t));
        CXS4_GRAB(3, flag, 2 * ((size_t)ncells + 64));   // flag | seedkeep
        CXS4_GRAB(4, seeds, ((size_t)n * 2 + 64) * sizeof(uint32_t));
        CXS4_GRAB(5, out, CXS4_OUT_WORDS * sizeof(uint32_t));
        CXS4_GRAB(6, ep, ((size_t)n * 8 + 8) * sizeof(int32_t));
        CXS4_GRAB(7, visited, vsize * sizeof(unsigned long long));
#undef CXS4_GRAB
        CXS4_TRY(hipMemsetAsync(flag, 0, 2 * ((size_t)ncells + 64), st));
        CXS4_TRY(hipMemsetAsync(out, 0, CXS4_OUT_WORDS * sizeof(uint32_t), st));
        CXS4_TRY(hipMemsetAsync(visited, 0, vsize * sizeof(unsigned long long), st));
        CXS4_TRY(hipMemsetAsync(S4->tet_keep, 0, (size_t)nt + 64, st));
        if (n) CXS4_TRY(hipMemcpyAsync(ep, endpoints_ijkl, (size_t)n * 8 * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (ncells && nt) {
            const uint32_t blocks = (ncells + 255u) / 256u;
            hipLaunchKernelGGL(cxs4_k_map, dim3(blocks), dim3(256), 0, st, S4->cells, ncells, vmap, parent, G);
            hipLaunchKernelGGL(cxs4_k_union_block, dim3((ncells + CXS4_UB - 1u) / CXS4_UB), dim3(256), 0, st, S4->cells, ncells, vmap, parent, G);
            hipLaunchKernelGGL(cxs4_k_union_far, dim3(blocks), dim3(256), 0, st, S4->cells, ncells, vmap, parent, G);
            hipLaunchKernelGGL(cxs4_k_flatten, dim3(blocks), dim3(256), 0, st, parent, ncells);
            if (n <= sequential_max)   // sequential, with the reference's shared visited set
                hipLaunchKernelGGL(cxs4_k_seeds, dim3(1), dim3(64), 0, st, G, ep, (uint32_t)n, visited, vsize - 1ULL, seeds, out);
            else
                hipLaunchKernelGGL(cxs4_k_seeds_parallel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, G, ep, (uint32_t)n, seeds, out);
            hipLaunchKernelGGL(cxs4_k_mark, dim3((uint32_t)((2 * n + 255) / 256) + 1u), dim3(256), 0, st, S4->cells, ncells, vmap, parent, seeds, out, flag, flag + ncells + 64, G);
            hipLaunchKernelGGL(cxs4_k_keep, dim3((nt + 255u) / 256u), dim3(256), 0, st, S4->cells, ncells, vmap, parent, flag, flag + ncells + 64, S4->tets, S4->vkeys, nt,
                               S4->tet_keep, out, G, all_in_range);
            hipLaunchKernelGGL(cxs4_k_keep_sum, dim3(1), dim3(CXS4_PARTIALS), 0, st, out);
            hipLaunchKernelGGL(cxs4_k_count_groups, dim3(blocks), dim3(256), 0, st, parent, ncells, flag, out);
        }
        CXS4_TRY(hipGetLastError());
        CXS4_TRY(hipMemcpyAsync(host_out, out, sizeof(host_out), hipMemcpyDeviceToHost, st));
        CXS4_TRY(hipStreamSynchronize(st));
#undef CXS4_TRY
    } while (0);
    if (rc) return rc;
    if (out_counts) {
        out_counts[0] = host_out[0]; out_counts[1] = host_out[2]; out_counts[2] = host_out[3]; out_counts[3] = host_out[1];
    }
    if (host_out[1]) { ctx->err = "cx_select_seeded4d: an end point pair does not straddle the isovalue (or lies outside the grid)"; return CX_ERR_INVALID; }
    S4->keep_valid = true;
    S4->post_valid = false;
    return CX_OK;
}

extern "C" int cx_seeded4d_mask_download(cx_ctx* ctx, uint8_t* tet_keep) {
    if (!ctx || !tet_keep) return CX_ERR_INVALID;
    cx_state4* S4 = ctx->s4;
    if (!S4 || !S4->extracted || !S4->keep_valid) { ctx->err = "cx_seeded4d_mask_download: run cx_select_seeded4d first"; return CX_ERR_STATE; }
    CXS4_HIP(ctx, hipSetDevice(ctx->device));
    if (S4->counts.n_triangles)
        CXS4_HIP(ctx, hipMemcpyAsync(tet_keep, S4->tet_keep, (size_t)S4->counts.n_triangles, hipMemcpyDeviceToHost, ctx->stream));
    CXS4_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CX_OK;
}
