// cx_seed.hip -- seeded selection of surface components for the 3-D march (C ABI: cx_select_seeded3d).
//
// Reference semantics restated (contourist/tetrahedral.py):
//   find_initial_voxels  :396-441  each end point pair is bisected until adjacent; for both points: the point's
//                                  own voxel if it is a border voxel, else the first border voxel among its 26
//                                  neighbours (OFFSETS order, :41-47); a shared `visited` set skips repeats
//   expand_voxels        :443-463  breadth-first over the 26 neighbours, border voxels inside the grid only
//   border_voxel         :383-394  min <= value <= max over the 8 corners and not np.allclose(value, corners)
// Here: the dense march has already produced every surface voxel (cell records); the voxels are grouped by
// 26-connectivity with a lock-free union-find over the records, the groups that contain a seed voxel are
// kept, and the triangles of all other voxels are masked out for the Level-1 post-pass.
// Deviations (documented in DESIGN.md): voxels without triangles (corners equal to the isovalue, min <= v <= max
// without a strict sign change) do not bridge groups; seed points whose voxel is not inside the array are skipped.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>

#include "cx_ctx.h"

#define CXS_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return (e__ == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;                      \
        }                                                                                        \
    } while (0)

struct cxs_grid {
    const float* A;
    uint32_t n0, n1, n2;
    double value;
    int lo[3], hi[3];   // in_range box of the breadth-first growth: lo <= voxel < hi (default 0 .. n-1)
};
__device__ __forceinline__ bool cxs_in_range(const cxs_grid& G, int i, int j, int k) {
    return i >= G.lo[0] && j >= G.lo[1] && k >= G.lo[2] && i < G.hi[0] && j < G.hi[1] && k < G.hi[2];
}

__device__ __forceinline__ bool cxs_is_voxel_record(const uint4& c) {
    return ((c.y >> 16) & 0xFFu) != 0u;   // has triangles (only real voxels do)
}
__device__ __forceinline__ uint32_t cxs_find(uint32_t* parent, uint32_t x) {
    for (;;) {
        const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        const uint32_t g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != p) atomicCAS(&parent[x], p, g);   // path halving
        x = p;
    }
}
__device__ __forceinline__ void cxs_union(uint32_t* parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = cxs_find(parent, a);
        b = cxs_find(parent, b);
        if (a == b) return;
        const uint32_t win = min(a, b), lose = max(a, b);
        if (atomicCAS(&parent[lose], lose, win) == lose) return;
    }
}

// abits: one bit per lattice point, set for the surface voxels (cleared by the host before): the union kernel asks it first -- a
// surface voxel has four or five of its 13 forward neighbours on the surface, and a bit that is clear costs a read of a 17 MB bitmap
// that stays in the caches instead of a random read of the 537 MB map plus one of the record it names (4.3 ms of the 5.8 ms a selection
// on the 512^3 bench field took)
__global__ void cxs_k_map(const uint4* cells, uint32_t ncells, uint32_t* vmap, uint32_t* parent, uint32_t* abits) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ncells) return;
    parent[r] = r;
    const uint4 c = cells[r];
    if (cxs_is_voxel_record(c)) {
        vmap[c.x] = r;
        atomicOr(&abits[c.x >> 5], 1u << (c.x & 31u));
    }
}
// record index of the surface voxel at linear index lin, or 0xFFFFFFFF (the map is not cleared: entries validate themselves)
__device__ __forceinline__ uint32_t cxs_lookup(const uint4* cells, uint32_t ncells, const uint32_t* vmap, uint32_t lin) {
    const uint32_t r = vmap[lin];
    if (r >= ncells) return 0xFFFFFFFFu;
    const uint4 c = cells[r];
    return (c.x == lin && cxs_is_voxel_record(c)) ? r : 0xFFFFFFFFu;
}
// The unions in two steps (as the Level-1 edge linking of cx_post.hip): records follow the march, so most of a voxel's surface neighbours
// are records of the same few hundred -- a workgroup unites CXS_UB consecutive records among themselves in an LDS forest and writes the
// forest into the global parent words with plain stores (cxs_k_union_block); only pairs that straddle two blocks go through the
// device-scope union-find (cxs_k_union_far).  (One step, every pair through device-scope atomics: 4.2 ms of the 5.8 ms a selection on
// the 512^3 bench field took -- the look-ups themselves, map and records, were not what it cost: a bitmap in front of them changed nothing.)
#define CXS_UB 1024u
__device__ __forceinline__ uint32_t cxs_lfind(uint32_t* lp, uint32_t x) {
    for (;;) {
        const uint32_t p = lp[x];
        if (p == x) return x;
        const uint32_t g = lp[p];
        if (g != p) atomicCAS(&lp[x], p, g);   // path halving
        x = p;
    }
}
// calls f(neighbour's linear index) for the 13 "forward" neighbours of voxel (i, j, k) inside the range whose bit is set (the other
// 13 are reached from the other side)
template <typename F>
__device__ __forceinline__ void cxs_forward_neighbours(const cxs_grid& G, const uint32_t* abits, uint32_t i, uint32_t j, uint32_t k, F f) {
    for (int di = 0; di <= 1; di++)
        for (int dj = -1; dj <= 1; dj++)
            for (int dk = -1; dk <= 1; dk++) {
                if (di == 0 && (dj < 0 || (dj == 0 && dk <= 0))) continue;
                const int ni = (int)i + di, nj = (int)j + dj, nk = (int)k + dk;
                if (!cxs_in_range(G, ni, nj, nk)) continue;   // in_range (:465-469)
                const uint32_t nl = ((uint32_t)ni * G.n1 + (uint32_t)nj) * G.n2 + (uint32_t)nk;
                if ((abits[nl >> 5] >> (nl & 31u)) & 1u) f(nl);      // (a set bit: cxs_k_map wrote the map entry in this very call)
            }
}
__global__ __launch_bounds__(256) void cxs_k_union_block(const uint4* cells, uint32_t ncells, const uint32_t* vmap, uint32_t* parent, cxs_grid G,
                                                         const uint32_t* abits) {
    __shared__ uint32_t lp[CXS_UB];
    const uint32_t b0 = blockIdx.x * CXS_UB;
    for (uint32_t x = threadIdx.x; x < CXS_UB; x += 256u) lp[x] = x;
    __syncthreads();
    const uint32_t plane = G.n1 * G.n2;
    for (uint32_t x = threadIdx.x; x < CXS_UB; x += 256u) {
        const uint32_t r = b0 + x;
        if (r >= ncells) continue;
        const uint4 c = cells[r];
        if (!cxs_is_voxel_record(c)) continue;
        const uint32_t i = c.x / plane, rem = c.x - i * plane, j = rem / G.n2, k = rem - j * G.n2;
        if (!cxs_in_range(G, (int)i, (int)j, (int)k)) continue;   // only in-range voxels grow (seed voxels outside the box: cxs_k_mark)
        cxs_forward_neighbours(G, abits, i, j, k, [&](uint32_t nl) {
            const uint32_t o = vmap[nl] - b0;
            if (o >= CXS_UB) return;                              // another block's record: cxs_k_union_far
            uint32_t a = x, b = o;
            for (;;) {
                a = cxs_lfind(lp, a);
                b = cxs_lfind(lp, b);
                if (a == b) break;
                const uint32_t win = min(a, b), lose = max(a, b);
                if (atomicCAS(&lp[lose], lose, win) == lose) break;
            }
        });
    }
    __syncthreads();
    // the block's forest into the global parent words (nobody else touches them in this kernel); roots = smallest ids
    for (uint32_t x = threadIdx.x; x < CXS_UB; x += 256u) {
        if (b0 + x >= ncells) continue;
        const uint32_t root = cxs_lfind(lp, x);
        if (root != x) parent[b0 + x] = b0 + root;
    }
}
__global__ void cxs_k_union_far(const uint4* cells, uint32_t ncells, const uint32_t* vmap, uint32_t* parent, cxs_grid G, const uint32_t* abits) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ncells) return;
    const uint4 c = cells[r];
    if (!cxs_is_voxel_record(c)) return;
    const uint32_t plane = G.n1 * G.n2;
    const uint32_t i = c.x / plane, rem = c.x - i * plane, j = rem / G.n2, k = rem - j * G.n2;
    if (!cxs_in_range(G, (int)i, (int)j, (int)k)) return;
    const uint32_t b0 = (r / CXS_UB) * CXS_UB;
    cxs_forward_neighbours(G, abits, i, j, k, [&](uint32_t nl) {
        const uint32_t o = vmap[nl];
        if (o - b0 >= CXS_UB) cxs_union(parent, r, o);            // (pairs inside one block are united already)
    });
}
__global__ void cxs_k_flatten(uint32_t* parent, uint32_t n) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) parent[r] = cxs_find(parent, r);
}

// ---- seeds: sequential, as the reference runs them (one thread; end point lists are short)
__device__ bool cxs_border_voxel(const cxs_grid& G, int i, int j, int k, bool& inside) {
    inside = i >= 0 && j >= 0 && k >= 0 && i + 1 < (int)G.n0 && j + 1 < (int)G.n1 && k + 1 < (int)G.n2;
    if (!inside) return false;
    double lo = 1e300, hi = -1e300;
    bool allclose = true;
    for (int c = 0; c < 8; c++) {
        const double f = (double)G.A[((size_t)(i + ((c >> 2) & 1)) * G.n1 + (size_t)(j + ((c >> 1) & 1))) * G.n2 + (size_t)(k + (c & 1))];
        lo = fmin(lo, f); hi = fmax(hi, f);
        if (!(fabs(G.value - f) <= 1e-8 + 1e-5 * fabs(f))) allclose = false;
    }
    if (allclose) return false;
    return lo <= G.value && hi >= G.value;
}
__device__ bool cxs_visit(unsigned long long* table, unsigned long long mask, long long i, long long j, long long k) {   // true: newly added
    const unsigned long long key = ((unsigned long long)(i + 4) << 42) | ((unsigned long long)(j + 4) << 21) | (unsigned long long)(k + 4);
    unsigned long long h = (key * 0x9E3779B97F4A7C15ULL) >> 20;
    for (;;) {
        const unsigned long long cur = table[h & mask];
        if (cur == key + 1ULL) return false;
        if (cur == 0ULL) { table[h & mask] = key + 1ULL; return true; }
        h++;
    }
}
__device__ __forceinline__ double cxs_f(const cxs_grid& G, const int p[3]) {
    return (double)G.A[((size_t)p[0] * G.n1 + (size_t)p[1]) * G.n2 + (size_t)p[2]];
}
// out[0] = number of seed voxels, out[1] = number of end point pairs that do not straddle the isovalue (error)
__global__ void cxs_k_seeds(cxs_grid G, const int32_t* ep, uint32_t n, unsigned long long* visited, unsigned long long vmask,
                            uint32_t* seeds, uint32_t* out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t ns = 0, bad = 0;
    for (uint32_t s = 0; s < n; s++) {
        int lowp[3] = {ep[s * 6 + 0], ep[s * 6 + 1], ep[s * 6 + 2]}, highp[3] = {ep[s * 6 + 3], ep[s * 6 + 4], ep[s * 6 + 5]};
        bool okp = true;
        for (int a = 0; a < 3; a++) {
            const int lim = (int)(a == 0 ? G.n0 : (a == 1 ? G.n1 : G.n2));
            if (lowp[a] < 0 || highp[a] < 0 || lowp[a] >= lim || highp[a] >= lim) okp = false;
        }
        if (!okp) { bad++; continue; }
        double lowv = cxs_f(G, lowp), highv = cxs_f(G, highp);
        if (lowv > G.value || highv < G.value) {
            for (int a = 0; a < 3; a++) { const int t = lowp[a]; lowp[a] = highp[a]; highp[a] = t; }
            const double t = lowv; lowv = highv; highv = t;
        }
        if (!(lowv <= G.value && highv >= G.value)) { bad++; continue; }   // the reference asserts here (:412-414)
        while (abs(lowp[0] - highp[0]) > 1 || abs(lowp[1] - highp[1]) > 1 || abs(lowp[2] - highp[2]) > 1) {
            int mid[3];
            for (int a = 0; a < 3; a++) {   // Python floor division
                const int sum = lowp[a] + highp[a];
                mid[a] = (sum >= 0) ? sum / 2 : -((-sum + 1) / 2);
            }
            if (cxs_f(G, mid) < G.value) { for (int a = 0; a < 3; a++) lowp[a] = mid[a]; }
            else { for (int a = 0; a < 3; a++) highp[a] = mid[a]; }
        }
        for (int which = 0; which < 2; which++) {
            const int* p = which ? highp : lowp;
            if (!cxs_visit(visited, vmask, p[0], p[1], p[2])) continue;
            bool inside;
            if (cxs_border_voxel(G, p[0], p[1], p[2], inside)) {
                seeds[ns++] = ((uint32_t)p[0] * G.n1 + (uint32_t)p[1]) * G.n2 + (uint32_t)p[2];
                continue;
            }
            bool found = false;
            for (int di = -1; di <= 1 && !found; di++)
                for (int dj = -1; dj <= 1 && !found; dj++)
                    for (int dk = -1; dk <= 1 && !found; dk++) {
                        if (di == 0 && dj == 0 && dk == 0) continue;
                        const int q0 = p[0] + di, q1 = p[1] + dj, q2 = p[2] + dk;
                        if (!cxs_visit(visited, vmask, q0, q1, q2)) continue;
                        if (cxs_border_voxel(G, q0, q1, q2, inside)) {
                            seeds[ns++] = ((uint32_t)q0 * G.n1 + (uint32_t)q1) * G.n2 + (uint32_t)q2;
                            found = true;
                        }
                    }
        }
    }
    out[0] = ns;
    out[1] = bad;
}
// flag[] = groups reached; seedkeep[] = seed voxels themselves (a seed voxel outside the in_range box is kept and
// grows one step into the box, as the reference's first expand_voxels round does)
// many end point pairs (coarse crossing search, skip > 1): one thread per pair, no shared `visited` set -- each end
// point yields its own voxel or its first border neighbour.  (The reference's shared set only changes which of
// several adjacent candidate voxels gets picked when pairs collide.)  Slots 2s, 2s+1; 0xFFFFFFFF = none.
__global__ void cxs_k_seeds_parallel(cxs_grid G, const int32_t* ep, uint32_t n, uint32_t* seeds, uint32_t* out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    seeds[2 * s] = seeds[2 * s + 1] = 0xFFFFFFFFu;
    int lowp[3] = {ep[s * 6 + 0], ep[s * 6 + 1], ep[s * 6 + 2]}, highp[3] = {ep[s * 6 + 3], ep[s * 6 + 4], ep[s * 6 + 5]};
    for (int a = 0; a < 3; a++) {
        const int lim = (int)(a == 0 ? G.n0 : (a == 1 ? G.n1 : G.n2));
        if (lowp[a] < 0 || highp[a] < 0 || lowp[a] >= lim || highp[a] >= lim) { atomicAdd(&out[1], 1u); return; }
    }
    double lowv = cxs_f(G, lowp), highv = cxs_f(G, highp);
    if (lowv > G.value || highv < G.value) {
        for (int a = 0; a < 3; a++) { const int t = lowp[a]; lowp[a] = highp[a]; highp[a] = t; }
        const double t = lowv; lowv = highv; highv = t;
    }
    if (!(lowv <= G.value && highv >= G.value)) { atomicAdd(&out[1], 1u); return; }
    while (abs(lowp[0] - highp[0]) > 1 || abs(lowp[1] - highp[1]) > 1 || abs(lowp[2] - highp[2]) > 1) {
        int mid[3];
        for (int a = 0; a < 3; a++) {
            const int sum = lowp[a] + highp[a];
            mid[a] = (sum >= 0) ? sum / 2 : -((-sum + 1) / 2);
        }
        if (cxs_f(G, mid) < G.value) { for (int a = 0; a < 3; a++) lowp[a] = mid[a]; }
        else { for (int a = 0; a < 3; a++) highp[a] = mid[a]; }
    }
    for (int which = 0; which < 2; which++) {
        const int* p = which ? highp : lowp;
        bool inside;
        if (cxs_border_voxel(G, p[0], p[1], p[2], inside)) {
            seeds[2 * s + which] = ((uint32_t)p[0] * G.n1 + (uint32_t)p[1]) * G.n2 + (uint32_t)p[2];
            continue;
        }
        bool found = false;
        for (int di = -1; di <= 1 && !found; di++)
            for (int dj = -1; dj <= 1 && !found; dj++)
                for (int dk = -1; dk <= 1 && !found; dk++) {
                    if (di == 0 && dj == 0 && dk == 0) continue;
                    if (cxs_border_voxel(G, p[0] + di, p[1] + dj, p[2] + dk, inside)) {
                        seeds[2 * s + which] = ((uint32_t)(p[0] + di) * G.n1 + (uint32_t)(p[1] + dj)) * G.n2 + (uint32_t)(p[2] + dk);
                        found = true;
                    }
                }
    }
    if (s == 0) out[0] = 2u * n;   // slots to look at
}
__global__ void cxs_k_mark(const uint4* cells, uint32_t ncells, const uint32_t* vmap, const uint32_t* parent, const uint32_t* seeds,
                           const uint32_t* nseeds, uint8_t* flag, uint8_t* seedkeep, cxs_grid G) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseeds[0]) return;
    const uint32_t lin = seeds[s];
    if (lin == 0xFFFFFFFFu) return;
    const uint32_t plane = G.n1 * G.n2;
    const int i = (int)(lin / plane), j = (int)((lin % plane) / G.n2), k = (int)(lin % G.n2);
    const uint32_t r = cxs_lookup(cells, ncells, vmap, lin);
    if (r != 0xFFFFFFFFu) {
        seedkeep[r] = 1;
        if (cxs_in_range(G, i, j, k)) { flag[parent[r]] = 1; return; }
    } else if (cxs_in_range(G, i, j, k)) {
        return;   // a border voxel without triangles (corners equal to the isovalue): does not grow here
    }
    for (int di = -1; di <= 1; di++)
        for (int dj = -1; dj <= 1; dj++)
            for (int dk = -1; dk <= 1; dk++) {
                if ((di | dj | dk) == 0 || !cxs_in_range(G, i + di, j + dj, k + dk)) continue;
                const uint32_t o = cxs_lookup(cells, ncells, vmap, ((uint32_t)(i + di) * G.n1 + (uint32_t)(j + dj)) * G.n2 + (uint32_t)(k + dk));
                if (o != 0xFFFFFFFFu) flag[parent[o]] = 1;
            }
}
// Triangles are numbered record after record, so the triangles of a workgroup's 256 records are ONE range of the triangle array: every
// record's thread decides for its voxel and leaves the decision per triangle in LDS, then the workgroup walks the range with one thread
// per TRIANGLE -- consecutive flag bytes, consecutive index triples (one thread per record looping over its up to 12 triangles wrote a
// byte and read a triple of its own per round: 0.76 ms of a selection's 2.0 on the 512^3 bench field).
#define CXS_KEEP_TRIS (256u * 12u)
#define CXS_PARTIAL0 32u        // out[32 + 32 p], out[33 + 32 p]: partial sums p = 0 .. CXS_PARTIALS - 1 of (groups kept, triangles kept)
#define CXS_PARTIALS 128u
#define CXS_OUT_WORDS (CXS_PARTIAL0 + 32u * CXS_PARTIALS)
__global__ void cxs_k_keep_sum(uint32_t* out) {
    uint32_t g = out[CXS_PARTIAL0 + 32u * threadIdx.x], t = out[CXS_PARTIAL0 + 32u * threadIdx.x + 1u];
    __shared__ uint32_t sg, st;
    if (threadIdx.x == 0) { sg = 0; st = 0; }
    __syncthreads();
    if (g) atomicAdd(&sg, g);
    if (t) atomicAdd(&st, t);
    __syncthreads();
    if (threadIdx.x == 0) { out[2] += sg; out[3] += st; }
}
__global__ __launch_bounds__(256) void cxs_k_keep(const uint4* cells, uint32_t ncells, const uint32_t* parent, const uint8_t* flag, const uint8_t* seedkeep,
                                                  uint8_t* tri_keep, const int32_t* tris, uint8_t* vkeep, uint32_t* out, cxs_grid G, int all_in_range) {
    __shared__ uint8_t lk[CXS_KEEP_TRIS];
    __shared__ uint32_t s_first, s_end, s_groups, s_tris;
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x == 0) { s_first = 0xFFFFFFFFu; s_end = 0u; s_groups = 0u; s_tris = 0u; }
    __syncthreads();
    uint4 c = make_uint4(0, 0, 0, 0);
    uint32_t ntri = 0;
    bool keep = false;
    if (r < ncells) {
        c = cells[r];
        ntri = (c.y >> 16) & 0xFFu;
        if (ntri) {
            const uint32_t plane = G.n1 * G.n2;
            const bool inr = cxs_in_range(G, (int)(c.x / plane), (int)((c.x % plane) / G.n2), (int)(c.x % G.n2));
            keep = (inr && (all_in_range || flag[parent[r]] != 0)) || seedkeep[r] != 0;
            // (counted per workgroup and added to one of 128 partial sums, each in a cache line of its own: one device-scope add per
            // kept voxel on ONE address -- same-address atomics execute one after the other -- was most of this kernel's 0.75 ms)
            if (keep && parent[r] == r) atomicAdd(&s_groups, 1u);   // groups kept
            if (keep) atomicAdd(&s_tris, ntri);
            atomicMin(&s_first, c.z);
            atomicMax(&s_end, c.z + ntri);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && (s_groups | s_tris)) {
        uint32_t* part = out + CXS_PARTIAL0 + 32u * (blockIdx.x & (CXS_PARTIALS - 1u));
        if (s_groups) atomicAdd(part, s_groups);
        if (s_tris) atomicAdd(part + 1, s_tris);
    }
    const uint32_t first = s_first, end = s_end;
    if (first >= end) return;                                      // no record of this workgroup has a triangle
    const bool fits = end - first <= CXS_KEEP_TRIS;               // (always, for records numbered in order; checked because LDS is finite)
    if (fits) {
        for (uint32_t x = threadIdx.x; x < end - first; x += 256u) lk[x] = 2;      // 2 = not a triangle of this workgroup's records
        __syncthreads();
        for (uint32_t t = 0; t < ntri; t++) lk[c.z - first + t] = keep ? 1 : 0;
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < end - first; x += 256u) {
            const uint8_t k = lk[x];
            if (k == 2) continue;
            tri_keep[first + x] = k;
            if (k) {
                const size_t at = (size_t)(first + x) * 3;
                const int32_t a = tris[at], b = tris[at + 1], d = tris[at + 2];
                vkeep[a] = 1; vkeep[b] = 1; vkeep[d] = 1;
            }
        }
    } else {
        for (uint32_t t = 0; t < ntri; t++) {
            tri_keep[c.z + t] = keep ? 1 : 0;
            if (keep)
                for (int s = 0; s < 3; s++) vkeep[tris[(size_t)(c.z + t) * 3 + s]] = 1;
        }
    }
}

extern "C" int cx_select_seeded3d_ex(cx_ctx* ctx, const int32_t* endpoints_ijk, int64_t n, const int32_t* range_lo_hi, uint32_t flags,
                                     int64_t* out_counts);
extern "C" int cx_select_seeded3d(cx_ctx* ctx, const int32_t* endpoints_ijk, int64_t n, const int32_t* range_lo_hi, int64_t* out_counts) {
    return cx_select_seeded3d_ex(ctx, endpoints_ijk, n, range_lo_hi, 0u, out_counts);
}
extern "C" int cx_select_seeded3d_ex(cx_ctx* ctx, const int32_t* endpoints_ijk, int64_t n, const int32_t* range_lo_hi, uint32_t flags,
                                     int64_t* out_counts) {
    if (!ctx || (n > 0 && !endpoints_ijk) || n < 0) return CX_ERR_INVALID;
    const int all_in_range = (flags & CX_SEED_ALL_IN_RANGE) ? 1 : 0;
    if (!ctx->extracted) { ctx->err = "cx_select_seeded3d: no valid extraction"; return CX_ERR_STATE; }
    CXS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    {
        const int rcr = cx_ensure_cell_records(ctx);   // the fused emit kernel leaves none behind
        if (rcr) return rcr;
    }
    const uint32_t ncells = (uint32_t)ctx->counts.n_cells, nt = (uint32_t)ctx->counts.n_triangles, nv = (uint32_t)ctx->counts.n_vertices;
    const cx_params& P = ctx->last;
    cxs_grid G;
    G.A = P.grid; G.n0 = P.n0; G.n1 = P.n1; G.n2 = P.n2; G.value = P.value;
    const int dims[3] = {(int)P.n0, (int)P.n1, (int)P.n2};
    for (int a = 0; a < 3; a++) {
        G.lo[a] = range_lo_hi ? std::max(range_lo_hi[a], 0) : 0;
        G.hi[a] = range_lo_hi ? std::min(range_lo_hi[3 + a], dims[a] - 1) : dims[a] - 1;
    }
    // persistent masks
    {
        const int rcg = cx_grow(ctx, ctx->tri_keep, ctx->keep_cap, (size_t)nt + (size_t)nv + 64);
        if (rcg) return rcg;
    }
    uint8_t* tri_keep = ctx->tri_keep;
    uint8_t* vkeep = ctx->tri_keep + nt;
    ctx->keep_valid = false;
    // scratch
    uint32_t *vmap = nullptr, *parent = nullptr, *seeds = nullptr, *out = nullptr, *abits = nullptr;
    uint8_t* flag = nullptr;
    int32_t* ep = nullptr;
    unsigned long long* visited = nullptr;
    unsigned long long vsize = 1024;
    // beyond this many pairs (or on request): one thread per pair, no shared visited set
    const int64_t CXS_SEQUENTIAL_MAX = (flags & CX_SEED_PARALLEL) ? -1 : 65536;
    while (n <= CXS_SEQUENTIAL_MAX && vsize < (unsigned long long)n * 54ULL * 4ULL) vsize <<= 1;
    ctx->seed_mode = (n <= CXS_SEQUENTIAL_MAX) ? 0 : 1;
    int rc = CX_OK;
    uint32_t host_out[4] = {0, 0, 0, 0};
    do {
        hipError_t e;
#define CXS_TRY(call) if ((e = (call)) != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e); rc = (e == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP; break; }
        // (scratch kept in the context between calls: cx_grow only ever grows)
#define CXS_GRAB(slot, ptr, bytes) { if ((rc = cx_grow(ctx, ctx->seed_buf[slot], ctx->seed_cap[slot], (size_t)(bytes)))) break; ptr = reinterpret_cast<decltype(ptr)>(ctx->seed_buf[slot]); }
        CXS_GRAB(0, vmap, ((size_t)P.nsamples + 64) * sizeof(uint32_t));
        CXS_GRAB(1, parent, ((size_t)ncells + 64) * sizeof(uint32_t));
        CXS_GRAB(2, abits, ((size_t)P.nsamples / 32 + 64) * sizeof(uint32_t));
        CXS_TRY(hipMemsetAsync(abits, 0, ((size_t)P.nsamples / 32 + 64) * sizeof(uint32_t), st));
        CXS_GRAB(3, flag, 2 * ((size_t)ncells + 64));
        CXS_GRAB(4, seeds, ((size_t)n * 2 + 64) * sizeof(uint32_t));
        CXS_GRAB(5, out, CXS_OUT_WORDS * sizeof(uint32_t));
        CXS_GRAB(6, ep, ((size_t)n * 6 + 8) * sizeof(int32_t));
        CXS_GRAB(7, visited, vsize * sizeof(unsigned long long));
#undef CXS_GRAB
        CXS_TRY(hipMemsetAsync(flag, 0, 2 * ((size_t)ncells + 64), st));
        CXS_TRY(hipMemsetAsync(out, 0, CXS_OUT_WORDS * sizeof(uint32_t), st));
        CXS_TRY(hipMemsetAsync(visited, 0, vsize * sizeof(unsigned long long), st));
        CXS_TRY(hipMemsetAsync(tri_keep, 0, (size_t)nt + (size_t)nv + 64, st));
        if (n) CXS_TRY(hipMemcpyAsync(ep, endpoints_ijk, (size_t)n * 6 * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (ncells) {
            const uint32_t blocks = (ncells + 255u) / 256u;
            hipLaunchKernelGGL(cxs_k_map, dim3(blocks), dim3(256), 0, st, ctx->cells, ncells, vmap, parent, abits);
            hipLaunchKernelGGL(cxs_k_union_block, dim3((ncells + CXS_UB - 1u) / CXS_UB), dim3(256), 0, st, ctx->cells, ncells, vmap, parent, G, (const uint32_t*)abits);
            hipLaunchKernelGGL(cxs_k_union_far, dim3(blocks), dim3(256), 0, st, ctx->cells, ncells, vmap, parent, G, (const uint32_t*)abits);
            hipLaunchKernelGGL(cxs_k_flatten, dim3(blocks), dim3(256), 0, st, parent, ncells);
            if (n <= CXS_SEQUENTIAL_MAX)   // sequential, with the reference's shared visited set
                hipLaunchKernelGGL(cxs_k_seeds, dim3(1), dim3(64), 0, st, G, ep, (uint32_t)n, visited, vsize - 1ULL, seeds, out);
            else
                hipLaunchKernelGGL(cxs_k_seeds_parallel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, G, ep, (uint32_t)n, seeds, out);
            hipLaunchKernelGGL(cxs_k_mark, dim3((uint32_t)((2 * n + 255) / 256) + 1u), dim3(256), 0, st, ctx->cells, ncells, vmap, parent, seeds, out, flag, flag + ncells + 64, G);
            hipLaunchKernelGGL(cxs_k_keep, dim3(blocks), dim3(256), 0, st, ctx->cells, ncells, parent, flag, flag + ncells + 64, tri_keep, ctx->tris, vkeep, out, G, all_in_range);
            hipLaunchKernelGGL(cxs_k_keep_sum, dim3(1), dim3(CXS_PARTIALS), 0, st, out);
        }
        CXS_TRY(hipGetLastError());
        CXS_TRY(hipMemcpyAsync(host_out, out, sizeof(host_out), hipMemcpyDeviceToHost, st));
        CXS_TRY(hipStreamSynchronize(st));
#undef CXS_TRY
    } while (0);
    if (rc) return rc;
    if (out_counts) {
        out_counts[0] = host_out[0]; out_counts[1] = host_out[2]; out_counts[2] = host_out[3]; out_counts[3] = host_out[1];
    }
    if (host_out[1]) { ctx->err = "cx_select_seeded3d: an end point pair does not straddle the isovalue (or lies outside the grid)"; return CX_ERR_INVALID; }
    ctx->keep_valid = true;
    ctx->post_valid = false;
    return CX_OK;
}

// host copies of the masks of the last cx_select_seeded3d (all ones when there is no selection): for callers that take
// the Level-0 mesh to the host before the post-pass (cx_postprocess3d_mesh)
extern "C" int cx_seeded_masks_download(cx_ctx* ctx, uint8_t* tri_keep, uint8_t* vert_keep) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->extracted) { ctx->err = "cx_seeded_masks_download: no valid extraction"; return CX_ERR_STATE; }
    CXS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nt = (size_t)ctx->counts.n_triangles, nv = (size_t)ctx->counts.n_vertices;
    if (!ctx->keep_valid) {
        if (tri_keep) memset(tri_keep, 1, nt);
        if (vert_keep) memset(vert_keep, 1, nv);
        return CX_OK;
    }
    if (tri_keep && nt) CXS_HIP(ctx, hipMemcpyAsync(tri_keep, ctx->tri_keep, nt, hipMemcpyDeviceToHost, ctx->stream));
    if (vert_keep && nv) CXS_HIP(ctx, hipMemcpyAsync(vert_keep, ctx->tri_keep + nt, nv, hipMemcpyDeviceToHost, ctx->stream));
    CXS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CX_OK;
}

extern "C" int cx_seeded_mode(cx_ctx* ctx, int* mode) {
    if (!ctx || !mode) return CX_ERR_INVALID;
    *mode = ctx->seed_mode;
    return CX_OK;
}
