// cx_post.hip -- Level-1 mesh post-passes on the device (surface-sized work).
//
// Reference semantics restated (paths relative to the reference checkout, contourist/...):
//   quantize_interpolations   tetrahedral.py:190-215      weld by bucket trunc(p * int(10000/corner))
//   remove_tiny_simplices     tetrahedral.py:353-375      drop tiny triangles, move their vertices together
//   extract_surface_geometry  tetrahedral.py:604-621      compact used vertices
//   clean_triangles           surface_geometry.py:14-50   drop zero-area triangles, merge their coincident vertices
//   orient_triangles          surface_geometry.py:52-140  per component: max-x start rule + edge flood fill
// Where the reference's result depends on Python set/dict order, a canonical choice is made (the
// member with the smallest priority -- edge id in the pipeline, index for a caller's mesh --
// represents a weld bucket / merge group; of triangles that become the same vertex set the one
// with the smallest sorted priority triple survives).  oracle/postpass.py restates the same choices.
//
// Building blocks: open-addressing hash tables (64-bit keys, atomicCAS), lock-free union-find
// (root = smallest priority; a parity bit per link carries the relative winding for the
// orientation flood fill), pointer jumping, a three-kernel exclusive scan.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "cx_ctx.h"
#include "cx_state4.h"

#define CXP_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return (e__ == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;                      \
        }                                                                                        \
    } while (0)

typedef unsigned long long u64;
#define CXP_EMPTY 0xFFFFFFFFFFFFFFFFULL
#define CXP_NONE 0xFFFFFFFFu

struct cxp_dev {
    void* p = nullptr;
    size_t bytes = 0;
};

struct cx_post_state {
    cxp_dev pts, prio, rep, tri, alive, parent, parent2, tkeys, tvals, flags, scan, blocksums, pts_out, tri_out, comp, misc;
    cxp_dev keys_out, keys_tmp, told, cls, bnd;   // edge ids of the output vertices; sharded Level 1 (cx_postprocess3d_shard_*)
    cxp_dev ever;                                 // vertices something was ever merged into: u8[nv] before | u8[nv2] after the compaction
    bool keys_valid = false;
    struct {
        bool open = false;            // between cx_postprocess3d_shard_begin and _finish
        uint32_t nv2 = 0, nt2 = 0;    // mesh of own + first-halo-layer triangles the labels refer to
        uint32_t nt_in = 0;           // triangles the post-pass started from (layout of S->cls)
        uint32_t n1 = 0, n4 = 0, ncand = 0;   // own triangles next to the lower neighbour, copies of the upper neighbour's, open components
    } shard;
    cxp_dev mpairs, msegs, mtris, mmid, mtime, mnext;   // morph triangles (4-D)
    // The morph triangles and their segments are kept SORTED by the bin of their start time (cxp_morph_sort_by_start, the last step of
    // cx_morph_triangles): the triangles that exist at a time t are then a window of ids, and so are their segments.
    cxp_dev msegs2, mtris2, mtime2;                     // the other halves of the double buffers the sort writes into
    cxp_dev meflags, metflag, menew, mecnt, medesc, me_pts, me_tri;   // cx_morph_eval_many: segment flag bytes (ALL zero between two calls) / triangle flag bytes / new point ids / block counts of the windows, per-time descriptors, outputs
    void* me_pinned = nullptr;                          // 48 KB of pinned host memory: descriptors up (36 KB), totals back (12 KB) without staging copies
    bool meflags_clean = false;                         // the segment flag bytes are all zero (the kernels that consume a flag clear it)
    uint32_t mbin_t[257] = {0}, mbin_s[257] = {0};      // first triangle / segment of every start-time bin (CXP_SB_BINS + 1 entries)
    double mt_lo = 0.0, mt_inv_width = 0.0;             // the bins: bin(x) = (x - mt_lo) * mt_inv_width, clamped (cxp_sb_bin)
    double mt_maxdur = 0.0, ms_maxdur = 0.0;            // longest life of a triangle / a segment
    bool msorted = false;
    std::vector<int64_t> me_off;                        // last cx_morph_eval_many: per time {first point, points, first triangle, triangles}
    int64_t nv_out = 0, nt_out = 0;
    int64_t ms_out = 0, mt_out = 0;
};

static int cxp_reserve(cx_ctx* ctx, cxp_dev& d, size_t bytes) {
    if (d.bytes >= bytes) return CX_OK;
    if (d.p) (void)hipFree(d.p);
    d.p = nullptr; d.bytes = 0;
    CXP_HIP(ctx, hipMalloc(&d.p, bytes));
    d.bytes = bytes;
    return CX_OK;
}

void cx_post_free(cx_ctx* ctx) {
    if (!ctx->post) return;
    cx_post_state* S = ctx->post;
    cxp_dev* all[] = {&S->pts, &S->prio, &S->rep, &S->tri, &S->alive, &S->parent, &S->parent2, &S->tkeys, &S->tvals,
                      &S->flags, &S->scan, &S->blocksums, &S->pts_out, &S->tri_out, &S->comp, &S->misc,
                      &S->keys_out, &S->keys_tmp, &S->told, &S->cls, &S->bnd, &S->ever,
                      &S->mpairs, &S->msegs, &S->mtris, &S->mmid, &S->mtime, &S->mnext, &S->msegs2, &S->mtris2, &S->mtime2,
                      &S->meflags, &S->metflag, &S->menew, &S->mecnt, &S->medesc, &S->me_pts, &S->me_tri};
    for (cxp_dev* d : all)
        if (d->p) (void)hipFree(d->p);
    if (S->me_pinned) (void)hipHostFree(S->me_pinned);
    delete S;
    ctx->post = nullptr;
}

// ---- device helpers --------------------------------------------------------------------------------
__device__ __forceinline__ u64 cxp_mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
// total order on doubles as unsigned integers
__device__ __forceinline__ u64 cxp_orderable(double x) {
    u64 b = (u64)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}

// monotonic maximum with a plain read first: once the running maximum is established almost every caller
// sees that its value cannot raise it and skips the atomic (same-address atomics serialise at ~88/us)
__device__ __forceinline__ void cxp_max64(u64* addr, u64 v) {
    if (__hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) return;
    atomicMax(addr, v);
}
__device__ __forceinline__ void cxp_max32(uint32_t* addr, uint32_t v) {
    if (__hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v) return;
    atomicMax(addr, v);
}
// the same for a whole wave: when all its lanes share one key (component) they reduce among themselves and issue
// ONE atomic; a wave that straddles components falls back to one call per lane.  (With values that grow along
// the array -- vertices ordered by x -- the plain read alone does not help: every caller raises the maximum.)
// All lanes of the wave must call it; `active` masks the idle ones.
__device__ __forceinline__ void cxp_wave_max64(u64* table, uint32_t key, u64 v, bool active) {
    const uint64_t act = __ballot(active);
    if (act == 0ULL) return;
    const uint32_t k = (uint32_t)__shfl((int)key, __ffsll((long long)act) - 1);
    if (__ballot(active && key != k) != 0ULL) {   // wave-uniform
        if (active) cxp_max64(&table[key], v);
        return;
    }
    u64 m = active ? v : 0ULL;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)m, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(m >> 32), o);
        const u64 other = ((u64)hi << 32) | lo;
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63u) == 0u) cxp_max64(&table[k], m);
}

// union-find over 32-bit ids; parent word = (parity << 32) | parent id.  Root = smallest priority.
__device__ __forceinline__ uint32_t cxp_find(const u64* parent, uint32_t x, uint32_t& parity) {
    uint32_t par = 0;
    for (;;) {
        const u64 w = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t p = (uint32_t)w;
        if (p == x) break;
        par ^= (uint32_t)(w >> 32) & 1u;
        x = p;
    }
    parity = par;
    return x;
}
// link the sets of a and b; rel = parity between a and b (0: same winding class).
// Every access to the parent words is a device-scope atomic (loads included): the L2s of the 8 XCDs are not
// coherent with each other inside a kernel, and a version with plain loads and plain path-halving stores showed a
// rare wrong winding of one component (stale lines mixing with memory-side compare-and-swaps).
__device__ __forceinline__ void cxp_union(u64* parent, const uint32_t* prio, uint32_t a, uint32_t b, uint32_t rel) {
    for (;;) {
        uint32_t pa, pb;
        uint32_t ra = cxp_find(parent, a, pa), rb = cxp_find(parent, b, pb);
        if (ra == rb) return;
        const uint32_t ka = prio ? prio[ra] : ra, kb = prio ? prio[rb] : rb;
        const bool a_wins = (ka < kb) || (ka == kb && ra < rb);
        const uint32_t win = a_wins ? ra : rb, lose = a_wins ? rb : ra;
        const u64 expect = (u64)lose;                                  // still a root, parity 0
        const u64 desired = ((u64)((pa ^ pb ^ rel) & 1u) << 32) | (u64)win;
        if (atomicCAS(&parent[lose], expect, desired) == expect) return;
    }
}

// The same for forests without parities and priorities (connectivity only: the march's own meshes), with path HALVING: on the way
// up every node is pointed at its grandparent -- with a device-scope store, which is executed where the compare-and-swaps are, so
// the two cannot disagree (see above) -- and only ever at an ancestor.  Most unions of a large component find both sides in one
// tree already; without the halving each of them walks the whole chain again (roots are the smallest ids, not the flattest trees).
__device__ __forceinline__ uint32_t cxp_find0(u64* parent, uint32_t x) {
    for (;;) {
        const uint32_t p = (uint32_t)__hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        const uint32_t g = (uint32_t)__hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g == p) return p;
        __hip_atomic_store(&parent[x], (u64)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = g;
    }
}
__device__ __forceinline__ void cxp_union0(u64* parent, uint32_t a, uint32_t b) {
    for (;;) {
        const uint32_t ra = cxp_find0(parent, a), rb = cxp_find0(parent, b);
        if (ra == rb) return;
        const uint32_t win = min(ra, rb), lose = max(ra, rb);
        if (atomicCAS(&parent[lose], (u64)lose, (u64)win) == (u64)lose) return;
        a = ra; b = rb;       // somebody else moved the loser: go on from the roots seen so far
    }
}

__global__ void cxp_k_iota64(u64* a, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (u64)i;
}
__global__ void cxp_k_fill64(u64* a, size_t n, u64 v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = v;
}
// pointer jumping with parity until every node points at its root
__global__ void cxp_k_jump(u64* parent, uint32_t n, uint32_t* changed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 w = parent[i];
    uint32_t p = (uint32_t)w;
    if (p == i) return;
    bool moved = false;
#pragma unroll 1
    for (int hop = 0; hop < 3; hop++) {          // up to three hops per launch: fewer launches and host round trips
        const u64 wp = parent[p];
        const uint32_t pp = (uint32_t)wp;
        if (pp == p) break;
        w = ((((w >> 32) ^ (wp >> 32)) & 1ULL) << 32) | (u64)pp;
        p = pp;
        moved = true;
    }
    if (moved) {
        parent[i] = w;
        *changed = 1u;
    }
}

// ---- exclusive scan of u32 (n up to 2^32) ----------------------------------------------------------
#define CXP_SCAN_BLOCK 1024u
__global__ __launch_bounds__(256) void cxp_k_scan_blocks(const uint32_t* in, uint32_t* out, uint32_t* sums, uint32_t n) {
    __shared__ uint32_t s[256];
    const uint32_t base = blockIdx.x * CXP_SCAN_BLOCK + threadIdx.x * 4u;
    uint32_t v[4], t = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        t += v[k];
    }
    s[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint32_t x = (threadIdx.x >= o) ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += x;
        __syncthreads();
    }
    uint32_t run = s[threadIdx.x] - t;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) sums[blockIdx.x] = s[255];
}
// one workgroup turns the block sums into exclusive offsets: 16 consecutive sums per thread, 16384 per pass
#define CXP_SUMS_PER_THREAD 16u
__device__ __forceinline__ void cxp_scan_sums_body(uint32_t* sums, uint32_t nb, uint32_t* total, unsigned long long* total64);
// two arrays in one launch (cx_morph_eval: the block counts of the segments and of the triangles), a workgroup each
__global__ __launch_bounds__(1024) void cxp_k_scan_sums2(uint32_t* sums_a, uint32_t na, uint32_t* total_a, uint32_t* sums_b, uint32_t nb, uint32_t* total_b) {
    if (blockIdx.x == 0) cxp_scan_sums_body(sums_a, na, total_a, nullptr);
    else cxp_scan_sums_body(sums_b, nb, total_b, nullptr);
}
__global__ __launch_bounds__(1024) void cxp_k_scan_sums(uint32_t* sums, uint32_t nb, uint32_t* total, unsigned long long* total64) {
    cxp_scan_sums_body(sums, nb, total, total64);
}
__device__ __forceinline__ void cxp_scan_sums_body(uint32_t* sums, uint32_t nb, uint32_t* total, unsigned long long* total64) {
    __shared__ uint32_t s[1024];
    uint32_t carry = 0;
    unsigned long long wide = 0;   // the same total without the wrap at 2^32 (each pass adds less than 2^32)
    for (uint32_t base = 0; base < nb; base += 1024u * CXP_SUMS_PER_THREAD) {
        const uint32_t i0 = base + threadIdx.x * CXP_SUMS_PER_THREAD;
        uint32_t v[CXP_SUMS_PER_THREAD], t = 0;
#pragma unroll
        for (uint32_t k = 0; k < CXP_SUMS_PER_THREAD; k++) {
            v[k] = (i0 + k < nb) ? sums[i0 + k] : 0u;
            t += v[k];
        }
        s[threadIdx.x] = t;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            const uint32_t x = (threadIdx.x >= o) ? s[threadIdx.x - o] : 0u;
            __syncthreads();
            s[threadIdx.x] += x;
            __syncthreads();
        }
        uint32_t run = carry + s[threadIdx.x] - t;
#pragma unroll
        for (uint32_t k = 0; k < CXP_SUMS_PER_THREAD; k++) {
            if (i0 + k < nb) sums[i0 + k] = run;
            run += v[k];
        }
        const uint32_t passtot = s[1023];
        __syncthreads();
        carry += passtot;
        wide += (unsigned long long)passtot;
    }
    if (threadIdx.x == 0) {
        *total = carry;
        if (total64) *total64 = wide;
    }
}
__global__ void cxp_k_scan_add(uint32_t* out, const uint32_t* sums, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += sums[i / CXP_SCAN_BLOCK];
}

// ---- step kernels -------------------------------------------------------------------------------------
// float64 vertex coordinates exactly as the reference computes them (tetrahedral.py:471-487)
struct cxp_origin3 {
    double o[3];   // lattice coordinates of sample (0,0,0) in the whole volume (cx_set_origin), or zeros
};
__global__ void cxp_k_vertices_f64(const float* __restrict__ A, const double* __restrict__ A64, uint32_t n1, uint32_t n2, cx_fdiv dplane, cx_fdiv drow,
                                   double value, const cx_vrec* __restrict__ verts, uint32_t nv, double* pts, uint32_t* prio, cxp_origin3 org) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const uint32_t key = verts[v].x;
    const uint32_t lin = key >> 3, d = key & 7u;
    const uint32_t plane = n1 * n2;
    const uint32_t i = cx_div(lin, dplane);
    const uint32_t r = lin - i * plane;
    const uint32_t j = cx_div(r, drow);
    const uint32_t k = r - j * n2;
    const uint32_t lin2 = lin + ((d & 4u) ? plane : 0u) + ((d & 2u) ? n2 : 0u) + (d & 1u);
    const double f0 = A64 ? A64[lin] : (double)A[lin], f1 = A64 ? A64[lin2] : (double)A[lin2];
    const bool owner_low = !(f0 > f1);             // reference swaps when flow > fhigh
    const double flow = owner_low ? f0 : f1, fhigh = owner_low ? f1 : f0;
    double ratio = 0.5;
    const double den = 1.0 * (fhigh - flow);
    if (!(fabs(den) <= 1e-8)) ratio = (value - flow) / den;
    const double q[3] = {(double)i + org.o[0], (double)j + org.o[1], (double)k + org.o[2]};   // integers: exact
    const uint32_t db[3] = {(d >> 2) & 1u, (d >> 1) & 1u, d & 1u};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double low = owner_low ? q[a] : q[a] + (double)db[a];
        const double high = owner_low ? q[a] + (double)db[a] : q[a];
        pts[(size_t)v * 3 + a] = low + ratio * (high - low);
    }
    prio[v] = key;
}

__global__ void cxp_k_iota_prio(uint32_t* prio, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) prio[i] = i;
}

struct cxp_weld_params {
    double ex[3];
    double thr;   // > 0: vertices are crossings of the march (priority = edge id); those further than thr from both ends of their
                  // lattice edge cannot share a bucket with another crossing (see cxp_weld_alone)
};
// Crossings sit on edges of the Kuhn lattice (one per edge).  Two of them in one weld bucket are less than 1/ex apart in every
// axis.  Edges without a common end point stay at least 1/3 apart (max-norm; checked over all pairs of edge directions), edges
// with a common end point V run apart from it along some axis -- so one of the two crossings is within 1/ex of V and the other
// within 2/ex.  A crossing at least 2/ex from both ends of its edge is therefore ALONE in its bucket (ex >= 16): it is its own
// representative and does not have to go through the table (at 512^3: 4 of 5 vertices).
__device__ __forceinline__ bool cxp_weld_alone(const double* p, uint32_t key, const cxp_weld_params& W) {
    if (!(W.thr > 0.0)) return false;
    const uint32_t d = key & 7u;
    const double x = (d & 4u) ? p[0] : ((d & 2u) ? p[1] : p[2]);   // an axis the edge moves along
    const double u = x - floor(x);
    return u > W.thr && u < 1.0 - W.thr;
}
// weld bucket of a point: trunc(p * expander) per axis as the reference's astype(int) does it (tetrahedral.py:192-196),
// i.e. TOWARDS ZERO: on an array with a rim around the reference's grid the coordinates are the reference's own and
// may be negative, and (-1/ex, 0) shares bucket 0 with [0, 1/ex).  Biased by 2^20 per axis so the fields stay unsigned.
__device__ __forceinline__ u64 cxp_weld_key(const double* p, const cxp_weld_params& W) {
    const u64 q0 = (u64)((long long)(p[0] * W.ex[0]) + (1LL << 20));
    const u64 q1 = (u64)((long long)(p[1] * W.ex[1]) + (1LL << 20));
    const u64 q2 = (u64)((long long)(p[2] * W.ex[2]) + (1LL << 20));
    return (q0 << 42) | (q1 << 21) | q2;
}

// weld: bucket -> vertex with the LARGEST priority (= largest edge key).  The members of a bucket are crossings on the
// upward edges of one lattice point (truncation puts everything in [V, V + 1/expander)^3 together), so the largest key
// is the most diagonal edge -- the one the reference's "last inserted wins" (tetrahedral.py:198-203) picks most often,
// because an edge shared by fewer voxels is first met later (probe on the corner = 511 fixture: 56 of 112 buckets
// agree, against 15 of 112 for the smallest key; tiny-collapse counts then fall inside the reference's own band)
__global__ void cxp_k_weld_insert(const double* pts, const uint32_t* prio, uint32_t nv, cxp_weld_params W, u64* tkeys, u64* tvals,
                                  u64 mask, const uint8_t* vkeep) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    if (vkeep && !vkeep[v]) return;   // seeded selection: vertices of dropped components do not exist
    if (cxp_weld_alone(pts + (size_t)v * 3, prio[v], W)) return;
    const u64 key = cxp_weld_key(pts + (size_t)v * 3, W);
    u64 slot = cxp_mix(key) & mask;
    for (;;) {
        const u64 cur = atomicCAS(&tkeys[slot], CXP_EMPTY, key);
        if (cur == CXP_EMPTY || cur == key) break;
        slot = (slot + 1) & mask;
    }
    atomicMax(&tvals[slot], ((u64)prio[v] << 32) | (u64)v);
}
__global__ void cxp_k_weld_lookup(const double* pts, uint32_t nv, cxp_weld_params W, const u64* tkeys, const u64* tvals, u64 mask,
                                  uint32_t* rep, const uint8_t* vkeep, const uint32_t* prio) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    if (vkeep && !vkeep[v]) { rep[v] = v; return; }
    if (cxp_weld_alone(pts + (size_t)v * 3, prio[v], W)) { rep[v] = v; return; }
    const u64 key = cxp_weld_key(pts + (size_t)v * 3, W);
    u64 slot = cxp_mix(key) & mask;
    while (tkeys[slot] != key) slot = (slot + 1) & mask;
    rep[v] = (uint32_t)tvals[slot];
}

// remap triangles through `map` (optionally a union-find: map == nullptr), kill those with < 3 distinct vertices.
// involved (optional): flags the vertices something was merged INTO.  Two different triangles of the march can only become the
// same vertex set through a merge, and then both contain such a vertex: the dedupe that follows only has to look at triangles
// with a flagged vertex (a few per cent of a mesh) instead of putting all of them through its table of compare-and-swaps.
// ever (optional): the same flags, never cleared between the stages -- an edge of the march's mesh can only come to lie on more than
// two triangles through such a vertex (cxp_k_edges_block).
__global__ void cxp_k_remap(int32_t* tri, uint8_t* alive, uint32_t nt, const uint32_t* map, const u64* parent, uint8_t* involved = nullptr,
                            uint8_t* ever = nullptr) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
    // the three indices, then the three look-ups, then the stores: written as one loop -- load, look up, store -- every load waited
    // for the store before it (they may alias as far as the compiler knows): seven round trips one after the other per triangle
    const uint32_t x[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
    uint32_t v[3];
    if (map) {
        v[0] = map[x[0]]; v[1] = map[x[1]]; v[2] = map[x[2]];
    } else {
        uint32_t par;
#pragma unroll
        for (int s = 0; s < 3; s++) v[s] = cxp_find(parent, x[s], par);
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        if (v[s] != x[s]) {
            tri[(size_t)t * 3 + s] = (int32_t)v[s];
            if (involved) involved[v[s]] = 1;
            if (ever) ever[v[s]] = 1;
        }
    }
    if (v[0] == v[1] || v[0] == v[2] || v[1] == v[2]) alive[t] = 0;
}

// priority of a triangle = its sorted triple of ORIGINAL vertex priorities (compared lexicographically)
struct cxp_tri3 {
    uint32_t a, b, c;
};
__device__ __forceinline__ cxp_tri3 cxp_sorted3(uint32_t x, uint32_t y, uint32_t z) {
    if (x > y) { const uint32_t t = x; x = y; y = t; }
    if (y > z) { const uint32_t t = y; y = z; z = t; }
    if (x > y) { const uint32_t t = x; x = y; y = t; }
    cxp_tri3 r = {x, y, z};
    return r;
}
__device__ __forceinline__ bool cxp_less3(const cxp_tri3& p, const cxp_tri3& q) {
    if (p.a != q.a) return p.a < q.a;
    if (p.b != q.b) return p.b < q.b;
    return p.c < q.c;
}

// dedupe triangles that are the same vertex set: table slot holds the id of the current winner
__global__ void cxp_k_dedupe_insert(const int32_t* tri, const uint32_t* tprio3, const uint8_t* alive, uint32_t nt, u64* table,
                                    u64 mask, const uint8_t* involved = nullptr) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
    if (involved && !(involved[tri[(size_t)t * 3]] | involved[tri[(size_t)t * 3 + 1]] | involved[tri[(size_t)t * 3 + 2]])) return;   // cannot have a twin
    const cxp_tri3 me = cxp_sorted3((uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]);
    const cxp_tri3 mp = {tprio3[(size_t)t * 3], tprio3[(size_t)t * 3 + 1], tprio3[(size_t)t * 3 + 2]};
    u64 slot = cxp_mix(((u64)me.a << 40) ^ ((u64)me.b << 20) ^ (u64)me.c ^ ((u64)me.c << 50)) & mask;
    for (;;) {
        u64 cur = atomicCAS(&table[slot], CXP_EMPTY, (u64)t);
        if (cur == CXP_EMPTY) return;
        for (;;) {   // slot occupied by triangle `cur`: same vertex set?
            const uint32_t o = (uint32_t)cur;
            const cxp_tri3 ot = cxp_sorted3((uint32_t)tri[(size_t)o * 3], (uint32_t)tri[(size_t)o * 3 + 1], (uint32_t)tri[(size_t)o * 3 + 2]);
            if (ot.a != me.a || ot.b != me.b || ot.c != me.c) break;   // different set: probe on
            const cxp_tri3 op = {tprio3[(size_t)o * 3], tprio3[(size_t)o * 3 + 1], tprio3[(size_t)o * 3 + 2]};
            if (!cxp_less3(mp, op)) return;                            // the resident wins
            const u64 seen = atomicCAS(&table[slot], cur, (u64)t);
            if (seen == cur) return;                                   // replaced it
            cur = seen;                                                // someone else got in: compare again
        }
        slot = (slot + 1) & mask;
    }
}
__global__ void cxp_k_dedupe_resolve(const int32_t* tri, uint8_t* alive, uint32_t nt, const u64* table, u64 mask, const uint8_t* involved = nullptr) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
    if (involved && !(involved[tri[(size_t)t * 3]] | involved[tri[(size_t)t * 3 + 1]] | involved[tri[(size_t)t * 3 + 2]])) return;   // was not inserted
    const cxp_tri3 me = cxp_sorted3((uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]);
    u64 slot = cxp_mix(((u64)me.a << 40) ^ ((u64)me.b << 20) ^ (u64)me.c ^ ((u64)me.c << 50)) & mask;
    for (;;) {
        const uint32_t o = (uint32_t)table[slot];
        if (o == t) return;
        const cxp_tri3 ot = cxp_sorted3((uint32_t)tri[(size_t)o * 3], (uint32_t)tri[(size_t)o * 3 + 1], (uint32_t)tri[(size_t)o * 3 + 2]);
        if (ot.a == me.a && ot.b == me.b && ot.c == me.c) { alive[t] = 0; return; }
        slot = (slot + 1) & mask;
    }
}

// original priority triple of every triangle (sorted), taken before any remap
__global__ void cxp_k_tri_prio(const int32_t* tri, const uint32_t* prio, uint32_t nt, uint32_t* tprio3) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const cxp_tri3 p = cxp_sorted3(prio[tri[(size_t)t * 3]], prio[tri[(size_t)t * 3 + 1]], prio[tri[(size_t)t * 3 + 2]]);
    tprio3[(size_t)t * 3] = p.a; tprio3[(size_t)t * 3 + 1] = p.b; tprio3[(size_t)t * 3 + 2] = p.c;
}

__global__ __launch_bounds__(256) void cxp_k_count_alive(const uint8_t* alive, uint32_t nt, uint32_t* counter) {
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t n = 0;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += gridDim.x * blockDim.x) n += alive[t] ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += (uint32_t)__shfl_xor((int)n, o);
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(counter, s_n);
}

// tiny triangles (tetrahedral.py:360-365): bbox * (1/corner) < epsilon on every axis
__global__ void cxp_k_tiny(const int32_t* tri, uint8_t* alive, uint32_t nt, const double* pts, double ic0, double ic1, double ic2,
                           double eps, u64* parent, const uint32_t* prio, uint8_t* moved) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
    const uint32_t v0 = tri[(size_t)t * 3], v1 = tri[(size_t)t * 3 + 1], v2 = tri[(size_t)t * 3 + 2];
    const double ic[3] = {ic0, ic1, ic2};
    double worst = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double x0 = pts[(size_t)v0 * 3 + a], x1 = pts[(size_t)v1 * 3 + a], x2 = pts[(size_t)v2 * 3 + a];
        const double d = (fmax(x0, fmax(x1, x2)) - fmin(x0, fmin(x1, x2))) * ic[a];
        worst = fmax(worst, d);
    }
    if (worst < eps) {
        alive[t] = 0;
        cxp_union(parent, prio, v0, v1, 0);
        cxp_union(parent, prio, v0, v2, 0);
        moved[v0] = moved[v1] = moved[v2] = 1;
    }
}
// members of a tiny group take the coordinates of the group's root (tetrahedral.py:368-370, canonical root)
__global__ void cxp_k_move(double* pts, const double* pts_src, const u64* parent, const uint8_t* moved, uint32_t nv) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv || !moved[v]) return;
    uint32_t par;
    const uint32_t r = cxp_find(parent, v, par);
#pragma unroll
    for (int a = 0; a < 3; a++) pts[(size_t)v * 3 + a] = pts_src[(size_t)r * 3 + a];
}

// zero-area triangles (surface_geometry.py:33-43): drop, and merge their np.allclose vertex pairs
__global__ void cxp_k_degenerate(const int32_t* tri, uint8_t* alive, uint32_t nt, const double* pts, u64* parent, const uint32_t* prio) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
    const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
    double P[3][3];
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int a = 0; a < 3; a++) P[s][a] = pts[(size_t)v[s] * 3 + a];
    const double ux = P[0][0] - P[2][0], uy = P[0][1] - P[2][1], uz = P[0][2] - P[2][2];   // A - C
    const double wx = P[1][0] - P[2][0], wy = P[1][1] - P[2][1], wz = P[1][2] - P[2][2];   // B - C
    const double cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
    if (fabs(cx) <= 1e-8 && fabs(cy) <= 1e-8 && fabs(cz) <= 1e-8) {
        alive[t] = 0;
        const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int i = pr[e][0], j = pr[e][1];
            bool close = true;
#pragma unroll
            for (int a = 0; a < 3; a++) close = close && (fabs(P[i][a] - P[j][a]) <= 1e-8 + 1e-5 * fabs(P[j][a]));
            if (close) cxp_union(parent, prio, v[i], v[j], 0);
        }
    }
}

__global__ void cxp_k_alive_u32(const uint8_t* alive, uint32_t nt, uint32_t* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nt) out[t] = alive[t] ? 1u : 0u;
}
// ---- ordered compaction without materialised scans (Level 1 of the 3-D path, cx_morph_eval) -------------------------------------
// (keys: the priority = edge id of every surviving vertex travels with it; told: where a surviving triangle came from)
// Flags are bytes; a workgroup covers CXP_SCAN_BLOCK consecutive elements: first pass counts its flags (cxp_k_me_count), one
// workgroup turns the block counts into offsets (cxp_k_scan_sums2), the consumer kernels redo the scan INSIDE their block while they
// write.  (Until round 3: a full exclusive scan per array -- three kernels that read and write 4 bytes per element twice.)
__device__ __forceinline__ uint32_t cxp_block_excl4(const uint32_t v[4], uint32_t* s, uint32_t& block_total) {
    // exclusive prefix of this thread's 4 consecutive elements within the workgroup of 256 threads (s: 256 words of LDS)
    const uint32_t t = v[0] + v[1] + v[2] + v[3];
    s[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint32_t x = (threadIdx.x >= o) ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += x;
        __syncthreads();
    }
    block_total = s[255];
    return s[threadIdx.x] - t;
}
__global__ __launch_bounds__(256) void cxp_k_me_count(const uint8_t* flags, uint32_t n, uint32_t* count) {
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * CXP_SCAN_BLOCK + threadIdx.x * 4u;
    uint32_t c = 0;
    if (base + 3u < n) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(flags + base);   // 4 flags (0 / 1 each); base is a multiple of 4
        c = __popc(w & 0x01010101u);
    } else {
        for (uint32_t k = 0; k < 4 && base + k < n; k++) c += flags[base + k] ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0) count[blockIdx.x] = s_n;
}
__global__ void cxp_k_mark_used8(const int32_t* tri, const uint8_t* alive, uint32_t nt, uint8_t* used) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
    const uint32_t a = (uint32_t)tri[(size_t)t * 3], b = (uint32_t)tri[(size_t)t * 3 + 1], c = (uint32_t)tri[(size_t)t * 3 + 2];   // all loads before the first store
    used[a] = 1; used[b] = 1; used[c] = 1;
}
// vertices in use -> their new ids (vnew, for the triangles) and their data in the compacted arrays
__global__ __launch_bounds__(256) void cxp_k_compact_pts_fused(const double* pts, const uint8_t* used, const uint32_t* voff, uint32_t nv, uint32_t* vnew,
                                                               double* out, const uint32_t* keys, uint32_t* keys_out, const uint8_t* ever, uint8_t* ever_out) {
    __shared__ uint32_t s[256];
    const uint32_t base = blockIdx.x * CXP_SCAN_BLOCK + threadIdx.x * 4u;
    uint32_t f[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) f[k] = (base + k < nv && used[base + k]) ? 1u : 0u;
    uint32_t tot;
    uint32_t id = voff[blockIdx.x] + cxp_block_excl4(f, s, tot);
    if (tot == 0u) return;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (!f[k]) continue;
        const uint32_t v = base + k;
        const double x = pts[(size_t)v * 3], y = pts[(size_t)v * 3 + 1], z = pts[(size_t)v * 3 + 2];
        const uint32_t key = keys_out ? keys[v] : 0u;
        const uint8_t e = ever_out ? ever[v] : (uint8_t)0;
        vnew[v] = id;
        out[(size_t)id * 3] = x; out[(size_t)id * 3 + 1] = y; out[(size_t)id * 3 + 2] = z;
        if (keys_out) keys_out[id] = key;
        if (ever_out) ever_out[id] = e;
        id++;
    }
}
__global__ __launch_bounds__(256) void cxp_k_compact_tri_fused(const int32_t* tri, const uint8_t* alive, const uint32_t* toff, const uint32_t* vnew, uint32_t nt,
                                                               int32_t* out, uint32_t* told) {
    __shared__ uint32_t s[256];
    const uint32_t base = blockIdx.x * CXP_SCAN_BLOCK + threadIdx.x * 4u;
    uint32_t f[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) f[k] = (base + k < nt && alive[base + k]) ? 1u : 0u;
    uint32_t tot;
    uint32_t id = toff[blockIdx.x] + cxp_block_excl4(f, s, tot);
    if (tot == 0u) return;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (!f[k]) continue;
        const uint32_t t = base + k;
        const uint32_t a = (uint32_t)tri[(size_t)t * 3], b = (uint32_t)tri[(size_t)t * 3 + 1], c = (uint32_t)tri[(size_t)t * 3 + 2];
        const uint32_t na = vnew[a], nb = vnew[b], nc = vnew[c];      // three look-ups in flight, then the stores
        out[(size_t)id * 3] = (int32_t)na; out[(size_t)id * 3 + 1] = (int32_t)nb; out[(size_t)id * 3 + 2] = (int32_t)nc;
        if (told) told[id] = t;
        id++;
    }
}

// ---- orientation ---------------------------------------------------------------------------------------
// edge table: key = (min vertex << 32) | max vertex, value = first triangle that inserted the edge.
// Every later triangle on that edge is linked to the first with the parity of their relative winding
// (surface_geometry.py:110-138: neighbours must run along the shared edge in opposite directions).
__device__ __forceinline__ uint32_t cxp_edge_dir(const int32_t* tri, uint32_t t, uint32_t lo, uint32_t hi) {
    // 1 if triangle t runs lo -> hi along its boundary cycle, 0 if hi -> lo
    const uint32_t a = tri[(size_t)t * 3], b = tri[(size_t)t * 3 + 1], c = tri[(size_t)t * 3 + 2];
    return ((a == lo && b == hi) || (b == lo && c == hi) || (c == lo && a == hi)) ? 1u : 0u;
}
// edge table: slot s = (key at tab[2s], first triangle at tab[2s+1]) -- one cache line per probe.
// Slots are placed by the SMALLER vertex id (lo * mult + a few bits of the larger one): vertex ids follow the
// order of the march, so the triangles of a wave probe neighbouring lines instead of random ones.
// Device-scope atomics are executed at the memory side of the fabric (64 bytes of write traffic each, ~21 G/s for
// the whole chip: PMC TCC_ATOMIC / WRITE_SIZE), so they are what this stage is bound by: ONE read-modify-write per
// edge visit claims the key, the claimant publishes its triangle with a plain store (kernel 1); after the kernel
// boundary every other visitor finds the slot again with plain loads and links itself to the claimant (kernel 2).
// (mult == 0: plain hashing -- the small first table of cxp_k_edges_block, where the edges that arrive are the ones around merged
// vertices, clustered in space: rows per vertex ran into each other there, chains of hundreds of probes)
__device__ __forceinline__ u64 cxp_edge_slot(uint32_t lo, uint32_t hi, u64 mask, u64 mult) {
    if (mult == 0) return cxp_mix(((u64)lo << 32) | (u64)hi) & mask;
    return ((u64)lo * mult + (cxp_mix((u64)hi) % mult)) & mask;
}
__global__ void cxp_k_edges_claim(const int32_t* tri, uint32_t nt, u64* tab, u64 mask, u64 mult) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
    for (int e = 0; e < 3; e++) {
        const uint32_t p = v[e], q = v[(e + 1) % 3];
        const uint32_t lo = min(p, q), hi = max(p, q);
        const u64 key = ((u64)lo << 32) | (u64)hi;
        u64 slot = cxp_edge_slot(lo, hi, mask, mult);
        for (;;) {
            // plain read first: the second visitor of an edge usually finds the key already there and needs no
            // read-modify-write (a stale EMPTY only costs the CAS it would have done anyway)
            u64 cur = __hip_atomic_load(&tab[2 * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == CXP_EMPTY) cur = atomicCAS(&tab[2 * slot], CXP_EMPTY, key);
            if (cur == CXP_EMPTY) { __hip_atomic_store(&tab[2 * slot + 1], (u64)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            if (cur == key) break;
            slot = (slot + 1) & mask;
        }
    }
}
// coherent == 0 (windings given by the caller, cx_surface_geometry): every visitor of an edge is united with the
// claimant with the parity of their relative winding, as the reference propagates it (surface_geometry.py:110-138).
// coherent != 0 (meshes of the march, which winds every triangle from low to high): two triangles that run along a
// shared edge in the same direction only occur where the weld has pinched sheets together along an edge shared by 3+
// triangles -- exactly the links that contradict each other (such an edge cannot have all of its triangles pairwise
// opposite).  United with their parities in one racing kernel, a contradictory union won somewhere and flipped a
// whole subtree: up to 4 % of the area of a 512^3 spherical shell came out wound the wrong way, differently from run
// to run.  So on these meshes a link only connects and never flips (parity 0; the neighbour's winding is not even
// read): both sides keep the march's winding -- a patch that hangs on a pinched edge still joins the component, as in
// the reference's traversal -- and the component is then turned as a whole by the max-x rule.
__global__ void cxp_k_edges_link(const int32_t* tri, uint32_t nt, const u64* tab, u64 mask, u64 mult, u64* parent, int coherent) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const uint32_t p = v[e], q = v[(e + 1) % 3];
        const uint32_t lo = min(p, q), hi = max(p, q);
        const u64 key = ((u64)lo << 32) | (u64)hi;
        u64 slot = cxp_edge_slot(lo, hi, mask, mult);
        while (tab[2 * slot] != key) slot = (slot + 1) & mask;   // every key was inserted by the claim kernel
        const uint32_t o = (uint32_t)tab[2 * slot + 1];
        if (o == t) continue;
        uint32_t rel = 0;
        if (!coherent) rel = (cxp_edge_dir(tri, t, lo, hi) == cxp_edge_dir(tri, o, lo, hi)) ? 1u : 0u;   // same direction = parity 1
        cxp_union(parent, nullptr, t, o, rel);
    }
}
// The same linking for the meshes of the march (coherent != 0: connectivity only) in two steps.  Triangle ids follow the march, so
// most edges join two triangles a few hundred ids apart: a workgroup takes CXP_LINK_BLOCK consecutive triangles, unites those
// whose partner lies in the same block in LDS (ds atomics instead of ~4 device-scope atomics per union, which are executed at
// the memory side of the fabric and bound the one-step kernel), writes the block's forest into the global parent words with
// plain stores -- nobody else touches them in this kernel -- and leaves the partners outside the block in `others` for the
// second kernel, which unites them with the global routine as before.  The roots are the smallest ids either way.
#ifndef CXP_LINK_BLOCK
#define CXP_LINK_BLOCK 2048u
#endif
#define CXP_LINK_PER_THREAD (CXP_LINK_BLOCK / 256u)
__device__ __forceinline__ uint32_t cxp_lfind(uint32_t* lp, uint32_t x) {
    for (;;) {
        const uint32_t p = ((volatile uint32_t*)lp)[x];
        if (p == x) return x;
        const uint32_t g = ((volatile uint32_t*)lp)[p];
        if (g != p) ((volatile uint32_t*)lp)[x] = g;   // path halving: only ever replaces a parent by an ancestor
        x = p;
    }
}
__global__ __launch_bounds__(256) void cxp_k_edges_link_local(const int32_t* tri, uint32_t nt, const u64* tab, u64 mask, u64 mult, u64* parent,
                                                              uint32_t* others) {
    __shared__ uint32_t lp[CXP_LINK_BLOCK];
    const uint32_t b0 = blockIdx.x * CXP_LINK_BLOCK;
    for (uint32_t x = threadIdx.x; x < CXP_LINK_BLOCK; x += 256u) lp[x] = x;
    __syncthreads();
#pragma unroll 1
    for (uint32_t i = 0; i < CXP_LINK_PER_THREAD; i++) {
        const uint32_t t = b0 + i * 256u + threadIdx.x;
        if (t >= nt) continue;
        const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const uint32_t p = v[e], q = v[(e + 1) % 3];
            const uint32_t lo = min(p, q), hi = max(p, q);
            const u64 key = ((u64)lo << 32) | (u64)hi;
            u64 slot = cxp_edge_slot(lo, hi, mask, mult);
            while (tab[2 * slot] != key) slot = (slot + 1) & mask;   // every key was inserted by the claim kernel
            const uint32_t o = (uint32_t)tab[2 * slot + 1];
            uint32_t far = CXP_NONE;
            if (o != t) {
                if (o - b0 < CXP_LINK_BLOCK) {          // partner in this block (o >= b0 by unsigned wrap-around)
                    uint32_t a = t - b0, b = o - b0;
                    for (;;) {
                        a = cxp_lfind(lp, a);
                        b = cxp_lfind(lp, b);
                        if (a == b) break;
                        const uint32_t win = min(a, b), lose = max(a, b);
                        if (atomicCAS(&lp[lose], lose, win) == lose) break;
                    }
                } else {
                    far = o;
                }
            }
            others[(size_t)t * 3 + e] = far;
        }
    }
    __syncthreads();
    for (uint32_t i = 0; i < CXP_LINK_PER_THREAD; i++) {
        const uint32_t x = i * 256u + threadIdx.x;
        if (b0 + x >= nt) continue;
        const uint32_t r = cxp_lfind(lp, x);
        if (r != x) parent[b0 + x] = (u64)(b0 + r);      // parity 0
    }
}
__global__ void cxp_k_edges_link_cross(uint32_t nt, const uint32_t* others, u64* parent) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
#pragma unroll
    for (int e = 0; e < 3; e++) {
        const uint32_t o = others[(size_t)t * 3 + e];
        if (o != CXP_NONE) cxp_union0(parent, t, o);
    }
}
// ---- the same linking for the MARCH'S OWN meshes, most of it in LDS ------------------------------------------------------
// An edge of the march's mesh lies on at most two triangles (a face of a tetrahedron has two sides), unless weld or clean-up
// merged something into one of its end points (`ever`).  Triangle ids follow the march: 86 % of the edges of the bench mesh join two
// triangles within the same 256 ids, 92 % within 1024 (tools/edge_locality.py).  A workgroup takes CXP_EB consecutive triangles and
// matches their edges in an LDS hash table; visitors of one edge are united in the block's LDS forest.  An edge found twice whose
// end points were never merged into is DONE and never sees the global table.  Only the rest goes through the device-scope table,
// one visit per block (the slot's taker): edges with one triangle in the block (the partner is in another block, or there is none)
// and edges with a flagged end point (their visitors in other blocks do the same, so three or more triangles on one edge still
// meet at the claimant, as in cxp_k_edges_link).  The table's compare-and-swaps (executed at the memory side) and the random
// 16-byte reads of the link step shrink to a sixth.  others[t] (one byte per triangle): bit e set = look edge e up in the second
// kernel, clear = settled here.
#ifndef CXP_EB
#define CXP_EB 512u
#endif
#define CXP_EB_PER (CXP_EB / 256u)      // triangles per thread
#define CXP_EB_SLOTS (4u * CXP_EB)       // > 3 * CXP_EB: an insert always finds a free slot
// (probe_limit / overflow: the global table is first sized for what usually reaches it -- a sixth of the edges -- and a probe
// sequence that long says it was too small for this mesh: the host repeats the stage with the table of the worst case)
__global__ __launch_bounds__(256) void cxp_k_edges_block(const int32_t* tri, uint32_t nt, const uint8_t* ever, u64* tab, u64 mask, u64 mult,
                                                         u64* parent, uint8_t* others, uint32_t probe_limit, uint32_t* overflow) {
    __shared__ u64 lkey[CXP_EB_SLOTS];
    __shared__ uint16_t lfirst[CXP_EB_SLOTS];
    __shared__ uint8_t lpair[CXP_EB_SLOTS];
    __shared__ uint32_t lp[CXP_EB];
    const uint32_t b0 = blockIdx.x * CXP_EB;
    for (uint32_t x = threadIdx.x; x < CXP_EB_SLOTS; x += 256u) { lkey[x] = CXP_EMPTY; lpair[x] = 0; }
    for (uint32_t x = threadIdx.x; x < CXP_EB; x += 256u) lp[x] = x;
    uint32_t lo_[CXP_EB_PER][3], hi_[CXP_EB_PER][3];
    uint32_t flag_[CXP_EB_PER][3];
    uint16_t slot_[CXP_EB_PER][3];
    // the triangles and the flags of their end points: all requested before anything is used
#pragma unroll
    for (uint32_t i = 0; i < CXP_EB_PER; i++) {
        const uint32_t t = min(b0 + i * 256u + threadIdx.x, nt - 1u);
        const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
#pragma unroll
        for (int e = 0; e < 3; e++) {
            lo_[i][e] = min(v[e], v[(e + 1) % 3]);
            hi_[i][e] = max(v[e], v[(e + 1) % 3]);
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < CXP_EB_PER; i++)
#pragma unroll
        for (int e = 0; e < 3; e++) flag_[i][e] = (uint32_t)ever[lo_[i][e]] | (uint32_t)ever[hi_[i][e]];
    __syncthreads();
    // pass 1: every edge visit finds or takes its slot; the taker leaves its triangle there
#pragma unroll
    for (uint32_t i = 0; i < CXP_EB_PER; i++) {
        const uint32_t lt = i * 256u + threadIdx.x;
        if (b0 + lt >= nt) continue;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const u64 key = ((u64)lo_[i][e] << 32) | (u64)hi_[i][e];
            uint32_t slot = (uint32_t)cxp_mix(key) & (CXP_EB_SLOTS - 1u);
            for (uint32_t probes = 0; probes < CXP_EB_SLOTS; probes++) {      // (always ends earlier: the table cannot fill up)
                const u64 cur = atomicCAS((unsigned long long*)&lkey[slot], (unsigned long long)CXP_EMPTY, (unsigned long long)key);
                if (cur == CXP_EMPTY) { lfirst[slot] = (uint16_t)lt; break; }
                if (cur == key) break;
                slot = (slot + 1u) & (CXP_EB_SLOTS - 1u);
            }
            slot_[i][e] = (uint16_t)slot;
        }
    }
    __syncthreads();
    // pass 2: the other visitors of a slot unite with the taker in the block's forest
#pragma unroll
    for (uint32_t i = 0; i < CXP_EB_PER; i++) {
        const uint32_t lt = i * 256u + threadIdx.x;
        if (b0 + lt >= nt) continue;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const uint32_t first = lfirst[slot_[i][e]];
            if (first == lt) continue;
            lpair[slot_[i][e]] = 1;
            uint32_t a = lt, b = first;
            for (;;) {
                a = cxp_lfind(lp, a);
                b = cxp_lfind(lp, b);
                if (a == b) break;
                const uint32_t win = min(a, b), lose = max(a, b);
                if (atomicCAS(&lp[lose], lose, win) == lose) break;
            }
        }
    }
    __syncthreads();
    // pass 3: what the block could not settle claims its place in the global table: an edge with one visitor here, and -- through
    // its taker alone, the block's other visitors are united with it already -- an edge with a flagged end point (its visitors in
    // OTHER blocks do the same and meet at the claimant)
#pragma unroll
    for (uint32_t i = 0; i < CXP_EB_PER; i++) {
        const uint32_t lt = i * 256u + threadIdx.x, t = b0 + lt;
        if (t >= nt) continue;
        uint32_t farbits = 0;      // bit e: edge e of this triangle went to the global table (one BYTE per triangle: the second kernel reads
                                   // nothing else of it -- as three 32-bit words this was 270 MB written and read back at 22.5 M triangles)
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const uint32_t lo = lo_[i][e], hi = hi_[i][e];
            const bool taker = lfirst[slot_[i][e]] == lt;
            if (taker && (flag_[i][e] != 0u || !lpair[slot_[i][e]])) {
                farbits |= 1u << e;
                const u64 key = ((u64)lo << 32) | (u64)hi;
                u64 slot = cxp_edge_slot(lo, hi, mask, mult);
                for (uint32_t probes = 0;; probes++) {
                    if (probes >= probe_limit) { *overflow = 1u; break; }
                    u64 cur = __hip_atomic_load(&tab[2 * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == CXP_EMPTY) cur = atomicCAS(&tab[2 * slot], CXP_EMPTY, key);
                    if (cur == CXP_EMPTY) { __hip_atomic_store(&tab[2 * slot + 1], (u64)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    if (cur == key) break;
                    slot = (slot + 1) & mask;
                }
            }
        }
        others[t] = (uint8_t)farbits;
    }
    // the block's forest into the global parent words (nobody else touches them in this kernel); roots = smallest ids
    for (uint32_t x = threadIdx.x; x < CXP_EB; x += 256u) {
        if (b0 + x >= nt) continue;
        const uint32_t r = cxp_lfind(lp, x);
        if (r != x) parent[b0 + x] = (u64)(b0 + r);      // parity 0
    }
}
// second kernel: the visitors the first one sent to the global table unite with the edge's claimant
__global__ void cxp_k_edges_link_far(const int32_t* tri, uint32_t nt, const u64* tab, u64 mask, u64 mult, const uint8_t* others, u64* parent) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const uint32_t fb = others[t];
    if (!fb) return;
    const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
#pragma unroll
    for (int e = 0; e < 3; e++) {
        if (!((fb >> e) & 1u)) continue;
        const uint32_t p = v[e], q = v[(e + 1) % 3];
        const uint32_t lo = min(p, q), hi = max(p, q);
        const u64 key = ((u64)lo << 32) | (u64)hi;
        u64 slot = cxp_edge_slot(lo, hi, mask, mult);
        while (tab[2 * slot] != key) slot = (slot + 1) & mask;   // every such key was inserted by the first kernel
        const uint32_t o = (uint32_t)tab[2 * slot + 1];
        if (o != t) cxp_union0(parent, t, o);
    }
}
// per component (root triangle): largest x over its vertices.  cls (sharded Level 1 only): per triangle, > 2 = a copy of a
// neighbour slab's triangle, which takes part in the components but not in the choice of the start triangle.
#define CXP_OWN(cls, t) (!(cls) || (cls)[t] <= 2u)
// Triangles are visited from the LAST to the first: the march numbers them by ascending x, so the first waves to run establish
// the maximum and everybody after them sees in a plain read that it has nothing to add (visited in ascending order every wave
// raises the maximum, and same-address atomics serialise).
__global__ void cxp_k_comp_maxx(const int32_t* tri, uint32_t nt, const double* pts, const u64* parent, u64* cmaxx, const uint8_t* cls) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = nt - 1u - idx;
    const bool have = idx < nt && CXP_OWN(cls, t);
    uint32_t root = 0;
    u64 m = 0;
    if (have) {
        root = (uint32_t)parent[t];
#pragma unroll
        for (int s = 0; s < 3; s++) m = max(m, cxp_orderable(pts[(size_t)tri[(size_t)t * 3 + s] * 3]));
    }
    cxp_wave_max64(cmaxx, root, m, have);
}
// The triangles that can hold the start triangle of their component: those with a vertex AT the component's largest x.  A handful
// per component (a face of the volume that the surface runs along: thousands) -- the four kernels below visit this list instead of
// every triangle.  One atomic per wave.
__global__ void cxp_k_comp_list(const int32_t* tri, uint32_t nt, const double* pts, const u64* parent, const u64* cmaxx, const uint8_t* cls,
                                uint32_t* list, uint32_t* nlist) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (t < nt && CXP_OWN(cls, t)) {
        const u64 m = cmaxx[(uint32_t)parent[t]];
#pragma unroll
        for (int s = 0; s < 3; s++) hit = hit || cxp_orderable(pts[(size_t)tri[(size_t)t * 3 + s] * 3]) == m;
    }
    const uint64_t hits = __ballot(hit);
    if (hits == 0ULL) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == (uint32_t)(__ffsll((long long)hits) - 1)) base = atomicAdd(nlist, (uint32_t)__popcll(hits));
    base = (uint32_t)__shfl((int)base, __ffsll((long long)hits) - 1);
    if (hit) list[base + (uint32_t)__popcll(hits & ((1ULL << lane) - 1ULL))] = t;
}
// visit either every triangle (list == nullptr: one per thread) or the triangles of a list (grid-stride)
#define CXP_FOR_TRI(t)                                                                                                   \
    for (uint32_t i_ = blockIdx.x * blockDim.x + threadIdx.x, n_ = list ? *nlist : nt; i_ < n_; i_ += gridDim.x * blockDim.x) { \
        const uint32_t t = list ? list[i_] : i_;
#define CXP_END_FOR }
// among the vertices at that x: the one with the largest index in the reference (surface_geometry.py:79 max((x, index)), its
// numbering being its hash order); here the one with the largest EDGE ID, which does not depend on how the mesh is numbered or
// cut into slabs.  Packed (edge id << 32 | vertex).
__global__ void cxp_k_comp_maxv(const int32_t* tri, uint32_t nt, const double* pts, const u64* parent, const u64* cmaxx, u64* cmaxv,
                                const uint32_t* keys, const uint8_t* cls, const uint32_t* list, const uint32_t* nlist) {
    CXP_FOR_TRI(t)
        if (!CXP_OWN(cls, t)) continue;
        const uint32_t root = (uint32_t)parent[t];
        const u64 m = cmaxx[root];
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const uint32_t v = tri[(size_t)t * 3 + s];
            if (cxp_orderable(pts[(size_t)v * 3]) == m) cxp_max64(&cmaxv[root], ((u64)(keys ? keys[v] : v) << 32) | (u64)v);
        }
    CXP_END_FOR
}
// among that vertex's triangles: the largest |cross(a-b, a-c)[0]| (surface_geometry.py:88-94); the packed
// word keeps the triangle id so that the next kernel can read its sign
__device__ __forceinline__ double cxp_dotx(const int32_t* tri, uint32_t t, const double* pts) {
    const uint32_t a = tri[(size_t)t * 3], b = tri[(size_t)t * 3 + 1], c = tri[(size_t)t * 3 + 2];
    const double aby = pts[(size_t)a * 3 + 1] - pts[(size_t)b * 3 + 1], abz = pts[(size_t)a * 3 + 2] - pts[(size_t)b * 3 + 2];
    const double acy = pts[(size_t)a * 3 + 1] - pts[(size_t)c * 3 + 1], acz = pts[(size_t)a * 3 + 2] - pts[(size_t)c * 3 + 2];
    return aby * acz - abz * acy;
}
__global__ void cxp_k_comp_start(const int32_t* tri, uint32_t nt, const double* pts, const u64* parent, const u64* cmaxv,
                                 u64* cbest, const uint8_t* cls, const uint32_t* list, const uint32_t* nlist) {
    CXP_FOR_TRI(t)
        if (!CXP_OWN(cls, t)) continue;
        const uint32_t root = (uint32_t)parent[t];
        const uint32_t vm = (uint32_t)cmaxv[root];
        if ((uint32_t)tri[(size_t)t * 3] != vm && (uint32_t)tri[(size_t)t * 3 + 1] != vm && (uint32_t)tri[(size_t)t * 3 + 2] != vm) continue;
        cxp_max64(&cbest[root], cxp_orderable(fabs(cxp_dotx(tri, t, pts))));
    CXP_END_FOR
}
__global__ void cxp_k_comp_pick(const int32_t* tri, uint32_t nt, const double* pts, const u64* parent, const u64* cmaxv,
                                const u64* cbest, uint32_t* cstart, const uint8_t* cls, const uint32_t* list, const uint32_t* nlist) {
    CXP_FOR_TRI(t)
        if (!CXP_OWN(cls, t)) continue;
        const uint32_t root = (uint32_t)parent[t];
        const uint32_t vm = (uint32_t)cmaxv[root];
        if ((uint32_t)tri[(size_t)t * 3] != vm && (uint32_t)tri[(size_t)t * 3 + 1] != vm && (uint32_t)tri[(size_t)t * 3 + 2] != vm) continue;
        if (cxp_orderable(fabs(cxp_dotx(tri, t, pts))) == cbest[root]) cxp_max32(&cstart[root], t);
    CXP_END_FOR
}
// the root's flip, so that the start triangle gets dotx > 0 (surface_geometry.py:99-103).  Decided in its own kernel,
// BEFORE any triangle is rewritten: cxp_k_orient reverses triangles in place, and reading the start triangle there
// raced with the thread that reverses it (a component came out partly wound one way and partly the other, rarely)
__global__ void cxp_k_comp_decide(const int32_t* tri, uint32_t nt, const double* pts, const u64* parent, const uint32_t* cstart, u64* cflip,
                                  const uint32_t* list, const uint32_t* nlist) {
    CXP_FOR_TRI(t)
        const u64 w = parent[t];
        const uint32_t root = (uint32_t)w;
        if (cstart[root] != t) continue;
        const uint32_t spar = (uint32_t)(w >> 32) & 1u;
        cflip[root] = (u64)((((cxp_dotx(tri, t, pts) < 0.0) ? 1u : 0u) ^ spar) & 1u);   // in the root's frame
    CXP_END_FOR
}
// final winding: triangle parity relative to the root, and the root's flip
__global__ void cxp_k_orient(int32_t* tri, uint32_t nt, const u64* parent, const u64* cflip, uint32_t* ncomp) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const u64 w = parent[t];
    const uint32_t root = (uint32_t)w, par = (uint32_t)(w >> 32) & 1u;
    if (root == t) atomicAdd(ncomp, 1u);
    if ((par ^ (uint32_t)cflip[root]) & 1u) {
        // reversed(orientation): (a,b,c) -> (c,b,a)
        const int32_t a = tri[(size_t)t * 3];
        tri[(size_t)t * 3] = tri[(size_t)t * 3 + 2];
        tri[(size_t)t * 3 + 2] = a;
    }
}

// ---- sharded Level 1: a slab marched together with two layers of its neighbours' cells ------------------
// Every triangle of the march lies in one cell; the smallest of its three edge ids belongs to an edge that starts in the
// cell's lower x face (a tetrahedron has no three crossing edges inside one face of the cube), so (id >> 3) / plane is the
// cell's layer.  Classes: 0 own, 1 / 2 own and next to the lower / upper neighbour, 3 / 4 first layer of the lower / upper
// neighbour (kept for the components), 255 second layer (only there so that weld, tiny collapse and clean-up of the first
// layer see everything they see in the undivided volume): dropped here.
struct cxp_shard {
    uint32_t plane;            // samples per x plane
    uint32_t own_lo, own_hi;   // own cell layers [own_lo, own_hi) of the local array
    uint32_t layers;           // cell layers of the local array
};
__global__ void cxp_k_shard_classify(const uint32_t* tprio3, uint8_t* alive, uint32_t nt, cxp_shard sh, cx_fdiv dplane, uint8_t* cls,
                                     uint32_t* counts) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    uint8_t c = 255;
    if (alive[t]) {
        const uint32_t layer = cx_div(tprio3[(size_t)t * 3] >> 3, dplane);
        if (layer >= sh.own_lo && layer < sh.own_hi) {
            c = 0;
            if (layer == sh.own_lo && sh.own_lo > 0u) c = 1;
            else if (layer + 1u == sh.own_hi && sh.own_hi < sh.layers) c = 2;
        } else if (layer + 1u == sh.own_lo) c = 3;
        else if (layer == sh.own_hi) c = 4;
        if (c == 255) alive[t] = 0;
        else if (c == 1) atomicAdd(&counts[0], 1u);
        else if (c == 4) atomicAdd(&counts[1], 1u);
    }
    cls[t] = c;
}
__global__ void cxp_k_shard_gather_cls(const uint8_t* cls, const uint32_t* told, uint32_t nt2, uint8_t* cls2) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nt2) cls2[t] = cls[told[t]];
}
// What a rank and its LOWER neighbour must agree on: the neighbour holds copies of this rank's first layer of own triangles
// (its class 4), this rank holds the originals (class 1).  Both write (hash of the three ORIGINAL edge ids in the numbering of
// the whole volume, component label); sorted by the hash the two lists line up entry by entry, and the label pairs say which
// components are one.  (The copies on the other side -- classes 2 / 3 -- would say the same again: a link between two
// slabs' triangles is seen from both.)  Components with such a triangle are marked open.
__device__ __forceinline__ u64 cxp_triple_hash(u64 a, u64 b, u64 c) {
    u64 h = cxp_mix(a + 0x9E3779B97F4A7C15ULL);
    h = cxp_mix(h ^ (b + 0xC2B2AE3D27D4EB4FULL));
    return cxp_mix(h ^ (c + 0x165667B19E3779F9ULL));
}
__global__ void cxp_k_shard_boundary(const uint8_t* cls2, const uint32_t* told, const uint32_t* tprio3, const u64* parent, uint32_t nt2,
                                     u64 key_offset, uint32_t* counters, uint32_t cap1, uint32_t cap4, u64* hash1, uint32_t* label1,
                                     u64* hash4, uint32_t* label4, uint8_t* open) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt2) return;
    const uint32_t c = cls2[t];
    if (c != 1u && c != 4u) return;
    const uint32_t root = (uint32_t)parent[t];
    open[root] = 1;
    const uint32_t o = told[t];
    const u64 h = cxp_triple_hash((u64)tprio3[(size_t)o * 3] + key_offset, (u64)tprio3[(size_t)o * 3 + 1] + key_offset,
                                  (u64)tprio3[(size_t)o * 3 + 2] + key_offset);
    if (c == 1u) {
        const uint32_t at = atomicAdd(&counters[0], 1u);
        if (at < cap1) { hash1[at] = h; label1[at] = root; }
    } else {
        const uint32_t at = atomicAdd(&counters[1], 1u);
        if (at < cap4) { hash4[at] = h; label4[at] = root; }
    }
}
// the start-triangle candidate of every open component, from this slab's own triangles: (label, x of the max-x vertex, its
// edge id, |normal_x| of the start triangle, its sign, whether there is a candidate at all)
struct cxp_cand { double x, nx; uint32_t label, vkey, sign, has; };
__global__ void cxp_k_shard_candidates(const int32_t* tri2, const double* pts2, const uint32_t* keys2, const u64* parent, const uint8_t* open,
                                       uint32_t nt2, const u64* cmaxx, const u64* cmaxv, const uint32_t* cstart, uint32_t* counter,
                                       uint32_t cap, cxp_cand* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt2 || (uint32_t)parent[t] != t || !open[t]) return;
    const uint32_t at = atomicAdd(counter, 1u);
    if (at >= cap) return;
    cxp_cand c;
    c.label = t; c.has = cmaxx[t] != 0 ? 1u : 0u; c.x = 0.0; c.nx = 0.0; c.vkey = 0; c.sign = 0;
    if (c.has) {
        const uint32_t vm = (uint32_t)cmaxv[t];
        const double d = cxp_dotx(tri2, cstart[t], pts2);
        c.x = pts2[(size_t)vm * 3]; c.vkey = keys2[vm]; c.nx = fabs(d); c.sign = d < 0.0 ? 1u : 0u;
    }
    out[at] = c;
}
__global__ void cxp_k_shard_set_flips(const uint32_t* labels, const uint8_t* flips, uint32_t n, uint32_t nt2, u64* cflip) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && labels[i] < nt2) cflip[labels[i]] = (u64)(flips[i] & 1u);
}
__global__ void cxp_k_shard_own_alive(const uint8_t* cls2, uint32_t nt2, uint8_t* alive) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nt2) alive[t] = cls2[t] <= 2u ? 1 : 0;
}


// ---- host orchestration ----------------------------------------------------------------------------------
static inline uint32_t cxp_blocks(size_t n, uint32_t b = 256) { return (uint32_t)((n + b - 1) / b); }
static inline u64 cxp_table_size(size_t n) {
    u64 s = 1024;
    while (s < 2 * (u64)n + 16) s <<= 1;
    return s;
}
// edge tables: n = 3 * triangles is an upper bound that only an open mesh without shared edges reaches; a closed
// mesh has half as many distinct edges.  5/4 of the bound keeps the worst case below a load of 0.8 and the usual
// case near 0.25-0.4 with half the footprint (the table is touched at random: footprint is what costs)
static inline u64 cxp_edge_table_size(size_t n) {
    u64 s = 1024;
    while (s < (5 * (u64)n) / 4 + 16) s <<= 1;
    return s;
}
// (measured and dropped: half that size for the meshes of the march, whose edges are nearly all shared by two triangles, with a
// repeat on overflow -- the fill gets 0.23 ms cheaper, the claim kernel 0.74 ms dearer: rows of 5 slots per vertex instead of 11
// run into each other)

static int cxp_scan(cx_ctx* ctx, cx_post_state* S, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_dev) {
    const uint32_t nb = cxp_blocks(n, CXP_SCAN_BLOCK);
    int rc = cxp_reserve(ctx, S->blocksums, (size_t)(nb + 1) * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t* sums = (uint32_t*)S->blocksums.p;
    hipLaunchKernelGGL(cxp_k_scan_blocks, dim3(nb ? nb : 1), dim3(256), 0, ctx->stream, in, out, sums, n);
    hipLaunchKernelGGL(cxp_k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, sums, nb, total_dev, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(cxp_k_scan_add, dim3(cxp_blocks(n)), dim3(256), 0, ctx->stream, out, sums, n);
    return CX_OK;
}

// exclusive scan for the other translation units (cx_contour2d.hip); sums_tmp holds n/1024 + 2 words
int cx_scan_u32(cx_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums_tmp, uint32_t* total_dev, unsigned long long* total64_dev) {
    const uint32_t nb = cxp_blocks(n, CXP_SCAN_BLOCK);
    hipLaunchKernelGGL(cxp_k_scan_blocks, dim3(nb ? nb : 1), dim3(256), 0, ctx->stream, in, out, sums_tmp, n);
    hipLaunchKernelGGL(cxp_k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, sums_tmp, nb, total_dev, total64_dev);
    if (n) hipLaunchKernelGGL(cxp_k_scan_add, dim3(cxp_blocks(n)), dim3(256), 0, ctx->stream, out, sums_tmp, n);
    return CX_OK;
}

static int cxp_flatten(cx_ctx* ctx, u64* parent, uint32_t n, uint32_t* changed_dev) {
    for (int it = 0; it < 64; it++) {
        CXP_HIP(ctx, hipMemsetAsync(changed_dev, 0, sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(cxp_k_jump, dim3(cxp_blocks(n)), dim3(256), 0, ctx->stream, parent, n, changed_dev);
        uint32_t changed = 0;
        CXP_HIP(ctx, hipMemcpyAsync(&changed, changed_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        CXP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!changed) return CX_OK;
    }
    ctx->err = "union-find did not flatten";
    return CX_ERR_HIP;
}

// Shared tail: clean (optional) + compaction + orientation (optional) on S->pts (nv x 3 doubles),
// S->tri (nt x 3), S->alive.  prio = vertex priorities.  Results in S->pts_out / S->tri_out.
static int cxp_clean_orient(cx_ctx* ctx, cx_post_state* S, uint32_t nv, uint32_t nt, bool do_clean, bool do_orient,
                            uint32_t* tprio3, int64_t* out_counts, bool coherent, uint8_t* involved = nullptr, const cxp_shard* shard = nullptr,
                            uint8_t* ever = nullptr) {   // ever != nullptr: the march's own mesh (cxp_k_edges_block); u8[nv] flags, room for nv more behind them
    int rc;
    double* pts = (double*)S->pts.p;
    uint32_t* prio = (uint32_t*)S->prio.p;
    int32_t* tri = (int32_t*)S->tri.p;
    uint8_t* alive = (uint8_t*)S->alive.p;
    uint32_t* misc = (uint32_t*)S->misc.p;   // [0] changed flag, [1..] counters
    hipStream_t st = ctx->stream;
    S->keys_valid = false;
    S->shard.open = false;
    if (do_clean && nt) {
        u64* parent2 = (u64*)S->parent2.p;
        hipLaunchKernelGGL(cxp_k_iota64, dim3(cxp_blocks(nv)), dim3(256), 0, st, parent2, nv);
        hipLaunchKernelGGL(cxp_k_degenerate, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, pts, parent2, prio);
        if (involved) CXP_HIP(ctx, hipMemsetAsync(involved, 0, nv, st));
        hipLaunchKernelGGL(cxp_k_remap, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, (const uint32_t*)nullptr, parent2, involved, ever);
        // (with the filter only triangles next to a merge enter the table: 5/4 of the bound is plenty and half as much to clear)
        const u64 tsz = involved ? cxp_edge_table_size(nt) : cxp_table_size(nt);
        if ((rc = cxp_reserve(ctx, S->tkeys, tsz * sizeof(u64)))) return rc;
        hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, (u64*)S->tkeys.p, (size_t)tsz, CXP_EMPTY);
        hipLaunchKernelGGL(cxp_k_dedupe_insert, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, tprio3, alive, nt, (u64*)S->tkeys.p, tsz - 1, (const uint8_t*)involved);
        hipLaunchKernelGGL(cxp_k_dedupe_resolve, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, (u64*)S->tkeys.p, tsz - 1, (const uint8_t*)involved);
    }
    // ---- sharded: the second layer of the neighbours' cells has done its work (weld, tiny collapse and clean-up above saw it)
    uint8_t* cls = nullptr;
    uint32_t* told = nullptr;
    if (shard) {
        if ((rc = cxp_reserve(ctx, S->cls, 3 * (size_t)nt + 64))) return rc;
        if ((rc = cxp_reserve(ctx, S->told, ((size_t)nt + 16) * sizeof(uint32_t)))) return rc;
        cls = (uint8_t*)S->cls.p;
        told = (uint32_t*)S->told.p;
        CXP_HIP(ctx, hipMemsetAsync(misc + 6, 0, 2 * sizeof(uint32_t), st));
        if (nt) hipLaunchKernelGGL(cxp_k_shard_classify, dim3(cxp_blocks(nt)), dim3(256), 0, st, tprio3, alive, nt, *shard, cx_fdiv_make(shard->plane), cls, misc + 6);
    }
    // ---- compaction of used vertices and living triangles (byte flags, per-block counts, the scan redone inside the two consumers)
    if ((rc = cxp_reserve(ctx, S->flags, (size_t)(nv + nt + 16) * sizeof(uint32_t)))) return rc;      // (also: the list of possible start triangles below)
    if ((rc = cxp_reserve(ctx, S->scan, (size_t)(nv + 16) * sizeof(uint32_t)))) return rc;
    const uint32_t nbv = cxp_blocks(nv, CXP_SCAN_BLOCK), nbt = cxp_blocks(nt, CXP_SCAN_BLOCK);
    if ((rc = cxp_reserve(ctx, S->blocksums, (size_t)(nbv + nbt + 16) * sizeof(uint32_t)))) return rc;
    uint8_t* used = (uint8_t*)S->flags.p;
    uint32_t* vnew = (uint32_t*)S->scan.p;
    uint32_t* voff = (uint32_t*)S->blocksums.p;
    uint32_t* toff = voff + nbv + 8;
    uint32_t nv2 = 0, nt2 = 0;
    if (nt) {
        CXP_HIP(ctx, hipMemsetAsync(used, 0, (size_t)nv, st));
        hipLaunchKernelGGL(cxp_k_mark_used8, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, used);
        hipLaunchKernelGGL(cxp_k_me_count, dim3(nbv), dim3(256), 0, st, (const uint8_t*)used, nv, voff);
        hipLaunchKernelGGL(cxp_k_me_count, dim3(nbt), dim3(256), 0, st, (const uint8_t*)alive, nt, toff);
        hipLaunchKernelGGL(cxp_k_scan_sums2, dim3(2), dim3(1024), 0, st, voff, nbv, misc + 1, toff, nbt, misc + 2);
        uint32_t h[2];
        CXP_HIP(ctx, hipMemcpyAsync(h, misc + 1, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        CXP_HIP(ctx, hipStreamSynchronize(st));
        nv2 = h[0]; nt2 = h[1];
    }
    if ((rc = cxp_reserve(ctx, S->pts_out, (size_t)(nv2 + 1) * 3 * sizeof(double)))) return rc;
    if ((rc = cxp_reserve(ctx, S->tri_out, (size_t)(nt2 + 1) * 3 * sizeof(int32_t)))) return rc;
    if ((rc = cxp_reserve(ctx, S->keys_out, (size_t)(nv2 + 1) * sizeof(uint32_t)))) return rc;
    double* pts2 = (double*)S->pts_out.p;
    int32_t* tri2 = (int32_t*)S->tri_out.p;
    uint32_t* keys2 = (uint32_t*)S->keys_out.p;
    if (nt2) {
        hipLaunchKernelGGL(cxp_k_compact_pts_fused, dim3(nbv), dim3(256), 0, st, (const double*)pts, (const uint8_t*)used, (const uint32_t*)voff, nv, vnew, pts2,
                           (const uint32_t*)prio, keys2, (const uint8_t*)ever, ever ? ever + nv : nullptr);
        hipLaunchKernelGGL(cxp_k_compact_tri_fused, dim3(nbt), dim3(256), 0, st, (const int32_t*)tri, (const uint8_t*)alive, (const uint32_t*)toff,
                           (const uint32_t*)vnew, nt, tri2, told);
    }
    S->nv_out = nv2; S->nt_out = nt2;
    S->keys_valid = true;
    uint8_t* cls2 = nullptr;
    if (shard) {
        cls2 = cls + nt;
        if (nt2) hipLaunchKernelGGL(cxp_k_shard_gather_cls, dim3(cxp_blocks(nt2)), dim3(256), 0, st, cls, told, nt2, cls2);
    }
    uint32_t ncomp = 0;
    if (do_orient && nt2) {
        // ---- orientation: edge table + parity union-find over triangles
        // (measured and dropped: clearing the table on a second stream while weld / tiny collapse / clean-up run -- no gain, the fill
        // takes from them what it saves)
        const u64 esz_full = cxp_edge_table_size((size_t)nt2 * 3);
        const bool blocks = ever && coherent && !cx_debug_knob("CX_LINK_GLOBAL", 0);
        // the march's own mesh: most edges are settled inside a block of triangles and never reach the table (cxp_k_edges_block): a
        // sixth of them for the bench mesh.  First attempt: one slot per TRIANGLE (a quarter of the worst case: 0.5 instead of 2 GB to
        // clear and to probe); a mesh that needs more says so and the stage is repeated with the full table.
        u64 esz = esz_full;
        if (blocks && !cx_debug_knob("CX_EDGE_TABLE_FULL", 0)) {
            esz = 1024;
            while (esz < (u64)nt2 + 16) esz <<= 1;
            if (cx_debug_knob("CX_EDGE_TABLE_TINY", 0)) esz = 1024;     // tests: force the repeat
            if (esz > esz_full) esz = esz_full;
        }
        if ((rc = cxp_reserve(ctx, S->parent, (size_t)nt2 * sizeof(u64)))) return rc;
        if ((rc = cxp_reserve(ctx, S->comp, (size_t)nt2 * (3 * sizeof(u64) + sizeof(uint32_t))))) return rc;
        u64* parent = (u64*)S->parent.p;
        u64* cmaxx = (u64*)S->comp.p;
        u64* cbest = cmaxx + nt2;
        u64* cmaxv = cbest + nt2;
        uint32_t* cstart = (uint32_t*)(cmaxv + nt2);
        uint32_t* others = (uint32_t*)S->comp.p;   // 12 of the 28 bytes per triangle that the component tables take below
        for (;;) {
            if ((rc = cxp_reserve(ctx, S->tkeys, 2 * esz * sizeof(u64)))) return rc;
            u64* etab = (u64*)S->tkeys.p;
            // (measured and dropped: clearing the table on a second stream while weld / tiny collapse / clean-up run -- no gain, the fill
            // takes from them what it saves)
            hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, etab, (size_t)(2 * esz), CXP_EMPTY);
            hipLaunchKernelGGL(cxp_k_iota64, dim3(cxp_blocks(nt2)), dim3(256), 0, st, parent, nt2);
            const u64 emult = (blocks && esz != esz_full) ? 0 : std::max<u64>(1, esz / std::max<u64>(1, (u64)nv2));
            if (blocks) {
                CXP_HIP(ctx, hipMemsetAsync(misc + 12, 0, sizeof(uint32_t), st));
                hipLaunchKernelGGL(cxp_k_edges_block, dim3((nt2 + CXP_EB - 1u) / CXP_EB), dim3(256), 0, st, tri2, nt2, (const uint8_t*)(ever + nv), etab, esz - 1,
                                   emult, parent, (uint8_t*)others, esz == esz_full ? 0xFFFFFFFFu : 256u, misc + 12);
                if (esz != esz_full) {
                    uint32_t over = 0;
                    CXP_HIP(ctx, hipMemcpyAsync(&over, misc + 12, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                    CXP_HIP(ctx, hipStreamSynchronize(st));
                    if (over) { esz = esz_full; continue; }
                }
                hipLaunchKernelGGL(cxp_k_edges_link_far, dim3(cxp_blocks(nt2)), dim3(256), 0, st, tri2, nt2, etab, esz - 1, emult, (const uint8_t*)others, parent);
            } else {
                hipLaunchKernelGGL(cxp_k_edges_claim, dim3(cxp_blocks(nt2)), dim3(256), 0, st, tri2, nt2, etab, esz - 1, emult);
                if (coherent && !cx_debug_knob("CX_LINK_ONE_STEP", 0)) {
                    hipLaunchKernelGGL(cxp_k_edges_link_local, dim3((nt2 + CXP_LINK_BLOCK - 1u) / CXP_LINK_BLOCK), dim3(256), 0, st, tri2, nt2, etab, esz - 1,
                                       emult, parent, others);
                    hipLaunchKernelGGL(cxp_k_edges_link_cross, dim3(cxp_blocks(nt2)), dim3(256), 0, st, nt2, others, parent);
                } else
                    hipLaunchKernelGGL(cxp_k_edges_link, dim3(cxp_blocks(nt2)), dim3(256), 0, st, tri2, nt2, etab, esz - 1, emult, parent, coherent ? 1 : 0);
            }
            break;
        }
        if ((rc = cxp_flatten(ctx, parent, nt2, misc))) return rc;
        CXP_HIP(ctx, hipMemsetAsync(cmaxx, 0, (size_t)nt2 * (3 * sizeof(u64) + sizeof(uint32_t)), st));
        CXP_HIP(ctx, hipMemsetAsync(misc + 3, 0, sizeof(uint32_t), st));
        const uint8_t* own = cls2;
        hipLaunchKernelGGL(cxp_k_comp_maxx, dim3(cxp_blocks(nt2)), dim3(256), 0, st, tri2, nt2, pts2, parent, cmaxx, own);
        // (the scan / flag arrays of the compaction are free by now: the list of possible start triangles goes there)
        uint32_t* clist = (uint32_t*)S->flags.p;
        const uint32_t* cn = misc + 11;
        const dim3 lgrid(std::min(cxp_blocks(nt2), 1024u));
        CXP_HIP(ctx, hipMemsetAsync(misc + 11, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(cxp_k_comp_list, dim3(cxp_blocks(nt2)), dim3(256), 0, st, tri2, nt2, pts2, parent, cmaxx, own, clist, misc + 11);
        hipLaunchKernelGGL(cxp_k_comp_maxv, lgrid, dim3(256), 0, st, tri2, nt2, pts2, parent, cmaxx, cmaxv, (const uint32_t*)keys2, own, (const uint32_t*)clist, cn);
        hipLaunchKernelGGL(cxp_k_comp_start, lgrid, dim3(256), 0, st, tri2, nt2, pts2, parent, cmaxv, cbest, own, (const uint32_t*)clist, cn);
        hipLaunchKernelGGL(cxp_k_comp_pick, lgrid, dim3(256), 0, st, tri2, nt2, pts2, parent, cmaxv, cbest, cstart, own, (const uint32_t*)clist, cn);
        if (shard) {
            // what the neighbours need: the triangles at the slab boundaries with their labels (they stay on the device), and the
            // start-triangle candidate of every component that reaches a neighbour.  Sizes follow the slab boundary, not the slab.
            uint32_t hb[2] = {0, 0};
            CXP_HIP(ctx, hipMemcpyAsync(hb, misc + 6, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            CXP_HIP(ctx, hipStreamSynchronize(st));
            const size_t n1 = hb[0], n4 = hb[1], nb = n1 + n4;
            // layout of S->bnd: hash1 u64[n1] | hash4 u64[n4] | candidates cxp_cand[nb] | label1 u32[n1] | label4 u32[n4]
            if ((rc = cxp_reserve(ctx, S->bnd, (nb + 2) * (sizeof(u64) + sizeof(cxp_cand) + sizeof(uint32_t)) + 64))) return rc;
            u64* hash1 = (u64*)S->bnd.p;
            u64* hash4 = hash1 + n1;
            cxp_cand* cand = (cxp_cand*)(hash4 + n4);
            uint32_t* label1 = (uint32_t*)(cand + nb);
            uint32_t* label4 = label1 + n1;
            uint8_t* open = cls + 2 * (size_t)nt;
            CXP_HIP(ctx, hipMemsetAsync(open, 0, nt2, st));
            CXP_HIP(ctx, hipMemsetAsync(misc + 8, 0, 3 * sizeof(uint32_t), st));
            if (nb) {
                const u64 key_offset = ((u64)ctx->origin[0] * (u64)shard->plane) << 3;
                hipLaunchKernelGGL(cxp_k_shard_boundary, dim3(cxp_blocks(nt2)), dim3(256), 0, st, cls2, told, tprio3, parent, nt2, key_offset, misc + 8,
                                   (uint32_t)n1, (uint32_t)n4, hash1, label1, hash4, label4, open);
                hipLaunchKernelGGL(cxp_k_shard_candidates, dim3(cxp_blocks(nt2)), dim3(256), 0, st, tri2, pts2, keys2, parent, open, nt2, cmaxx, cmaxv,
                                   cstart, misc + 10, (uint32_t)nb, cand);
            }
            uint32_t h2[3] = {0, 0, 0};
            CXP_HIP(ctx, hipMemcpyAsync(h2, misc + 8, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            CXP_HIP(ctx, hipStreamSynchronize(st));
            if (h2[0] != n1 || h2[1] != n4 || h2[2] > nb) { ctx->err = "sharded Level 1: boundary lists do not add up"; return CX_ERR_HIP; }
            S->shard.n1 = (uint32_t)n1; S->shard.n4 = (uint32_t)n4; S->shard.ncand = h2[2];
        }
        hipLaunchKernelGGL(cxp_k_comp_decide, lgrid, dim3(256), 0, st, tri2, nt2, pts2, parent, cstart, cbest, (const uint32_t*)clist, cn);
        if (!shard) {
            hipLaunchKernelGGL(cxp_k_orient, dim3(cxp_blocks(nt2)), dim3(256), 0, st, tri2, nt2, parent, cbest, misc + 3);
            CXP_HIP(ctx, hipMemcpyAsync(&ncomp, misc + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        }
        CXP_HIP(ctx, hipStreamSynchronize(st));
    } else if (shard) {
        S->shard.n1 = 0; S->shard.n4 = 0; S->shard.ncand = 0;
    }
    if (shard) { S->shard.open = true; S->shard.nv2 = nv2; S->shard.nt2 = nt2; S->shard.nt_in = nt; }
    CXP_HIP(ctx, hipGetLastError());
    if (out_counts) {
        out_counts[0] = nv2; out_counts[1] = nt2; out_counts[4] = ncomp;
    }
    return CX_OK;
}

// sharded Level 1, last step: the flips the ranks agreed on for the components that reach a neighbour replace the local
// decisions, every triangle is wound, and the copies of the neighbours' triangles leave the mesh
static int cxp_shard_finish(cx_ctx* ctx, cx_post_state* S, const uint32_t* labels, const uint8_t* flips, uint32_t n, int64_t* out_counts) {
    int rc;
    hipStream_t st = ctx->stream;
    uint32_t* misc = (uint32_t*)S->misc.p;
    const uint32_t nv2 = S->shard.nv2, nt2 = S->shard.nt2;
    S->shard.open = false;
    uint32_t ncomp = 0, nv3 = 0, nt3 = 0;
    if (nt2) {
        u64* parent = (u64*)S->parent.p;
        u64* cflip = (u64*)S->comp.p + nt2;
        int32_t* tri2 = (int32_t*)S->tri_out.p;
        const uint8_t* cls2 = (const uint8_t*)S->cls.p + S->shard.nt_in;
        if (n) {
            if ((rc = cxp_reserve(ctx, S->keys_tmp, std::max((size_t)n * 5 + 64, (size_t)(nv2 + 1) * sizeof(uint32_t))))) return rc;
            uint32_t* dl = (uint32_t*)S->keys_tmp.p;
            uint8_t* df = (uint8_t*)(dl + n);
            CXP_HIP(ctx, hipMemcpyAsync(dl, labels, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            CXP_HIP(ctx, hipMemcpyAsync(df, flips, (size_t)n, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(cxp_k_shard_set_flips, dim3(cxp_blocks(n)), dim3(256), 0, st, dl, df, n, nt2, cflip);
        }
        CXP_HIP(ctx, hipMemsetAsync(misc + 3, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(cxp_k_orient, dim3(cxp_blocks(nt2)), dim3(256), 0, st, tri2, nt2, parent, cflip, misc + 3);
        // own triangles and the vertices they use (the same ordered compaction as in cxp_clean_orient)
        uint8_t* alive = (uint8_t*)S->alive.p;
        uint8_t* used = (uint8_t*)S->flags.p;
        uint32_t* vnew = (uint32_t*)S->scan.p;
        const uint32_t nbv = cxp_blocks(nv2, CXP_SCAN_BLOCK), nbt = cxp_blocks(nt2, CXP_SCAN_BLOCK);
        if ((rc = cxp_reserve(ctx, S->blocksums, (size_t)(nbv + nbt + 16) * sizeof(uint32_t)))) return rc;
        uint32_t* voff = (uint32_t*)S->blocksums.p;
        uint32_t* toff = voff + nbv + 8;
        hipLaunchKernelGGL(cxp_k_shard_own_alive, dim3(cxp_blocks(nt2)), dim3(256), 0, st, cls2, nt2, alive);
        CXP_HIP(ctx, hipMemsetAsync(used, 0, (size_t)nv2, st));
        hipLaunchKernelGGL(cxp_k_mark_used8, dim3(cxp_blocks(nt2)), dim3(256), 0, st, (const int32_t*)tri2, (const uint8_t*)alive, nt2, used);
        hipLaunchKernelGGL(cxp_k_me_count, dim3(nbv), dim3(256), 0, st, (const uint8_t*)used, nv2, voff);
        hipLaunchKernelGGL(cxp_k_me_count, dim3(nbt), dim3(256), 0, st, (const uint8_t*)alive, nt2, toff);
        hipLaunchKernelGGL(cxp_k_scan_sums2, dim3(2), dim3(1024), 0, st, voff, nbv, misc + 1, toff, nbt, misc + 2);
        uint32_t h[3];
        CXP_HIP(ctx, hipMemcpyAsync(h, misc + 1, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        CXP_HIP(ctx, hipStreamSynchronize(st));
        nv3 = h[0]; nt3 = h[1]; ncomp = h[2];
        // the march's own buffers are free by now: they take the final mesh on its way back into the output buffers
        if ((rc = cxp_reserve(ctx, S->pts, (size_t)(nv3 + 1) * 3 * sizeof(double)))) return rc;
        if ((rc = cxp_reserve(ctx, S->tri, (size_t)(nt3 + 1) * 3 * sizeof(int32_t)))) return rc;
        if ((rc = cxp_reserve(ctx, S->keys_tmp, (size_t)(nv3 + 1) * sizeof(uint32_t)))) return rc;
        if (nt3) {
            hipLaunchKernelGGL(cxp_k_compact_pts_fused, dim3(nbv), dim3(256), 0, st, (const double*)S->pts_out.p, (const uint8_t*)used, (const uint32_t*)voff, nv2, vnew,
                               (double*)S->pts.p, (const uint32_t*)S->keys_out.p, (uint32_t*)S->keys_tmp.p, (const uint8_t*)nullptr, (uint8_t*)nullptr);
            hipLaunchKernelGGL(cxp_k_compact_tri_fused, dim3(nbt), dim3(256), 0, st, (const int32_t*)tri2, (const uint8_t*)alive, (const uint32_t*)toff,
                               (const uint32_t*)vnew, nt2, (int32_t*)S->tri.p, (uint32_t*)nullptr);
        }
        // back into the output buffers (every buffer keeps its size from call to call: nothing is reallocated for the next volume)
        if (nt3) {
            CXP_HIP(ctx, hipMemcpyAsync(S->pts_out.p, S->pts.p, (size_t)nv3 * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
            CXP_HIP(ctx, hipMemcpyAsync(S->tri_out.p, S->tri.p, (size_t)nt3 * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            CXP_HIP(ctx, hipMemcpyAsync(S->keys_out.p, S->keys_tmp.p, (size_t)nv3 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        }
        CXP_HIP(ctx, hipStreamSynchronize(st));
    }
    S->nv_out = nv3; S->nt_out = nt3;
    S->keys_valid = true;
    CXP_HIP(ctx, hipGetLastError());
    if (out_counts) { out_counts[0] = nv3; out_counts[1] = nt3; out_counts[4] = ncomp; }
    return CX_OK;
}

static int cxp_state(cx_ctx* ctx, cx_post_state** out) {
    if (!ctx->post) ctx->post = new (std::nothrow) cx_post_state();
    if (!ctx->post) return CX_ERR_NOMEM;
    *out = ctx->post;
    // every post-pass rewrites the points the morph triangles of an earlier cx_morph_triangles index: those are gone (their segments would
    // point into the new points: cx_morph_eval on them read out of bounds)
    ctx->post->ms_out = 0; ctx->post->mt_out = 0; ctx->post->msorted = false; ctx->post->me_off.clear();
    return cxp_reserve(ctx, ctx->post->misc, 512 * sizeof(uint32_t));      // (words 32..288: bin starts of the start-time sort)
}

// ---- smooth_interpolations(factor) (tetrahedral.py:329-351): every vertex that is part of a triangle moves by
// `factor` towards the mean of the vertices of its triangles (itself included, each neighbour once)
__global__ void cxp_k_unique_edges(const int32_t* tri, const uint8_t* alive, uint32_t nt, u64* ekeys, u64 mask) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
    const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
    for (int e = 0; e < 3; e++) {
        const uint32_t p = v[e], q = v[(e + 1) % 3];
        const u64 key = ((u64)min(p, q) << 32) | (u64)max(p, q);
        u64 slot = cxp_mix(key) & mask;
        for (;;) {
            u64 cur = __hip_atomic_load(&ekeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the second visitor of an edge needs no read-modify-write
            if (cur == CXP_EMPTY) cur = atomicCAS(&ekeys[slot], CXP_EMPTY, key);
            if (cur == CXP_EMPTY || cur == key) break;
            slot = (slot + 1) & mask;
        }
    }
}
__global__ void cxp_k_smooth_accumulate(const u64* ekeys, size_t n, const double* pts, double* sum, uint32_t* cnt) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n || ekeys[i] == CXP_EMPTY) return;
    const uint32_t a = (uint32_t)(ekeys[i] >> 32), b = (uint32_t)ekeys[i];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        atomicAdd(&sum[(size_t)a * 3 + c], pts[(size_t)b * 3 + c]);
        atomicAdd(&sum[(size_t)b * 3 + c], pts[(size_t)a * 3 + c]);
    }
    atomicAdd(&cnt[a], 1u);
    atomicAdd(&cnt[b], 1u);
}
// The two kernels above in one, with the sums put together per workgroup first (default; the two-kernel form stays for comparison
// behind CX_SMOOTH_TWO_STEP): the thread whose compare-and-swap claimed an edge's slot accumulates that edge, right there -- in triangle
// order, where the triangles of a workgroup share their vertices --, into an LDS table keyed by vertex, and the workgroup adds every
// vertex of its table to the global sums ONCE.  (One device-scope add per edge, coordinate and direction -- eight per edge, 270 M on the
// 512^3 bench mesh -- took 12.1 of the 14.4 ms smoothing added to Level 1.)
#define CXP_SM_SLOTS 1024u
__global__ __launch_bounds__(256) void cxp_k_smooth_edges(const int32_t* tri, const uint8_t* alive, uint32_t nt, u64* ekeys, u64 mask, const double* pts,
                                                          double* sum, uint32_t* cnt) {
    __shared__ uint32_t lv[CXP_SM_SLOTS];
    __shared__ uint32_t lc[CXP_SM_SLOTS];
    __shared__ double ls[CXP_SM_SLOTS][3];
    for (uint32_t x = threadIdx.x; x < CXP_SM_SLOTS; x += 256u) { lv[x] = CXP_NONE; lc[x] = 0u; ls[x][0] = 0.0; ls[x][1] = 0.0; ls[x][2] = 0.0; }
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nt && alive[t]) {
        const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
        auto add = [&](uint32_t to, uint32_t from) {      // the point `from` into the sums of vertex `to`
            uint32_t slot = (to * 0x9E3779B1u) >> 22;      // 10 bits
            for (;;) {
                const uint32_t cur = atomicCAS(&lv[slot], CXP_NONE, to);
                if (cur == CXP_NONE || cur == to) break;
                slot = (slot + 1u) & (CXP_SM_SLOTS - 1u);  // (at most 768 vertices per workgroup: a free slot is always found)
            }
            unsafeAtomicAdd(&ls[slot][0], pts[(size_t)from * 3]);
            unsafeAtomicAdd(&ls[slot][1], pts[(size_t)from * 3 + 1]);
            unsafeAtomicAdd(&ls[slot][2], pts[(size_t)from * 3 + 2]);
            atomicAdd(&lc[slot], 1u);
        };
        for (int e = 0; e < 3; e++) {
            const uint32_t p = v[e], q = v[(e + 1) % 3];
            const u64 key = ((u64)min(p, q) << 32) | (u64)max(p, q);
            u64 slot = cxp_mix(key) & mask;
            bool mine = false;
            for (;;) {
                u64 cur = __hip_atomic_load(&ekeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the second visitor of an edge needs no read-modify-write
                if (cur == CXP_EMPTY) { cur = atomicCAS(&ekeys[slot], CXP_EMPTY, key); mine = cur == CXP_EMPTY; }
                if (cur == CXP_EMPTY || cur == key) break;
                slot = (slot + 1) & mask;
            }
            if (mine && p != q) { add(p, q); add(q, p); }
            else if (mine) { add(p, p); add(p, p); }        // (a degenerate edge: what the two-kernel form does with it)
        }
    }
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < CXP_SM_SLOTS; x += 256u) {
        const uint32_t to = lv[x];
        if (to == CXP_NONE) continue;
        atomicAdd(&sum[(size_t)to * 3], ls[x][0]);
        atomicAdd(&sum[(size_t)to * 3 + 1], ls[x][1]);
        atomicAdd(&sum[(size_t)to * 3 + 2], ls[x][2]);
        atomicAdd(&cnt[to], lc[x]);
    }
}
__global__ void cxp_k_smooth_apply(double* pts, const double* sum, const uint32_t* cnt, uint32_t nv, double factor) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv || cnt[v] == 0u) return;
    const double inv = 1.0 / (double)(cnt[v] + 1u);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double p = pts[(size_t)v * 3 + c];
        const double avg = (sum[(size_t)v * 3 + c] + p) * inv;
        pts[(size_t)v * 3 + c] = p - factor * (p - avg);
    }
}

extern "C" int cx_postprocess3d_ex(cx_ctx* ctx, uint32_t flags, double smooth, int64_t* out_counts);

extern "C" int cx_postprocess3d(cx_ctx* ctx, uint32_t flags, int64_t* out_counts) {
    return cx_postprocess3d_ex(ctx, flags, 0.0, out_counts);
}

// buffers of the 3-D post-pass for nv vertices and nt triangles
static int cxp_reserve3d(cx_ctx* ctx, cx_post_state* S, uint32_t nv, uint32_t nt) {
    int rc;
    if ((rc = cxp_reserve(ctx, S->pts, (size_t)(nv + 1) * 3 * sizeof(double)))) return rc;
    if ((rc = cxp_reserve(ctx, S->prio, (size_t)(nv + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = cxp_reserve(ctx, S->rep, (size_t)(nv + 1) * (sizeof(uint32_t) + 1)))) return rc;
    if ((rc = cxp_reserve(ctx, S->tri, (size_t)(nt + 1) * 3 * sizeof(int32_t) * 2))) return rc;   // triangles + priority triples
    if ((rc = cxp_reserve(ctx, S->alive, (size_t)nt + 16))) return rc;
    if ((rc = cxp_reserve(ctx, S->parent, (size_t)(nv + 1) * sizeof(u64)))) return rc;
    if ((rc = cxp_reserve(ctx, S->parent2, (size_t)(nv + 1) * sizeof(u64)))) return rc;
    return CX_OK;
}
// weld -> (smooth) -> tiny collapse -> clean -> orient on S->pts / S->prio / S->tri / S->alive (A6..A10)
// edge_crossings: the vertices are the march's own crossings with their edge ids as priorities (cx_postprocess3d*), not points
// handed over by a caller (cx_postprocess3d_mesh: refined points, slab meshes with global edge ids as ranks)
static int cxp_run3d(cx_ctx* ctx, cx_post_state* S, uint32_t nv, uint32_t nt, const double corner[3], const uint8_t* vkeep, bool do_clean,
                     double smooth, bool coherent, int64_t* counts, bool edge_crossings = false, const cxp_shard* shard = nullptr, bool march_mesh = false) {
    int rc;
    hipStream_t st = ctx->stream;
    double* pts = (double*)S->pts.p;
    uint32_t* prio = (uint32_t*)S->prio.p;
    uint32_t* rep = (uint32_t*)S->rep.p;
    uint8_t* moved = (uint8_t*)(rep + nv + 1);
    int32_t* tri = (int32_t*)S->tri.p;
    uint32_t* tprio3 = (uint32_t*)(tri + (size_t)(nt + 1) * 3);
    uint8_t* alive = (uint8_t*)S->alive.p;
    uint32_t* misc = (uint32_t*)S->misc.p;
    uint8_t* ever = nullptr;      // the march's own crossings: which vertices weld or clean-up merge something into
    // (march_mesh: a mesh the march emitted, handed back by the caller -- slabs assembled on the host, refined points --: an edge lies on
    // at most two triangles until something is merged into one of its ends, which is all the block linking needs; the weld shortcut needs
    // the crossings' own geometry and stays with edge_crossings)
    if (nv && nt && (edge_crossings || march_mesh) && coherent) {
        if ((rc = cxp_reserve(ctx, S->ever, 2 * (size_t)nv + 64))) return rc;
        ever = (uint8_t*)S->ever.p;
        CXP_HIP(ctx, hipMemsetAsync(ever, 0, nv, st));
    }
    if (nv && nt) {
        hipLaunchKernelGGL(cxp_k_tri_prio, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, prio, nt, tprio3);
        // ---- weld (tetrahedral.py:190-215): expander = int(10000 / corner)
        cxp_weld_params W;
        for (int a = 0; a < 3; a++) W.ex[a] = std::trunc((10000 * 1.0) / corner[a]);
        W.thr = 0.0;
        if (edge_crossings && !cx_debug_knob("CX_WELD_ALL", 0)) {
            const double exmin = std::min(W.ex[0], std::min(W.ex[1], W.ex[2]));
            if (exmin >= 16.0) W.thr = 2.0 / exmin + 1e-9;
        }
        // (W.thr > 0: four of five crossings are alone in their bucket and skip the table -- 5/4 of the bound instead of twice it)
        const u64 wsz = W.thr > 0.0 ? cxp_edge_table_size(nv) : cxp_table_size(nv);
        if ((rc = cxp_reserve(ctx, S->tkeys, std::max(wsz, cxp_table_size(nt)) * sizeof(u64)))) return rc;
        if ((rc = cxp_reserve(ctx, S->tvals, wsz * sizeof(u64)))) return rc;
        hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, (u64*)S->tkeys.p, (size_t)wsz, CXP_EMPTY);
        hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, (u64*)S->tvals.p, (size_t)wsz, (u64)0);
        hipLaunchKernelGGL(cxp_k_weld_insert, dim3(cxp_blocks(nv)), dim3(256), 0, st, pts, prio, nv, W, (u64*)S->tkeys.p, (u64*)S->tvals.p, wsz - 1, vkeep);
        hipLaunchKernelGGL(cxp_k_weld_lookup, dim3(cxp_blocks(nv)), dim3(256), 0, st, pts, nv, W, (u64*)S->tkeys.p, (u64*)S->tvals.p, wsz - 1, rep, vkeep, prio);
        // meshes of the march hold no triangle twice: only triangles with a vertex something was welded into can have a twin
        // (`moved` is free until the tiny collapse: it carries the flags)
        uint8_t* involved = (coherent && !cx_debug_knob("CX_DEDUPE_ALL", 0)) ? moved : nullptr;
        if (involved) CXP_HIP(ctx, hipMemsetAsync(involved, 0, nv, st));
        hipLaunchKernelGGL(cxp_k_remap, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, rep, (const u64*)nullptr, involved, ever);
        const u64 tsz = involved ? cxp_edge_table_size(nt) : cxp_table_size(nt);
        hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, (u64*)S->tkeys.p, (size_t)tsz, CXP_EMPTY);
        hipLaunchKernelGGL(cxp_k_dedupe_insert, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, tprio3, alive, nt, (u64*)S->tkeys.p, tsz - 1, (const uint8_t*)involved);
        hipLaunchKernelGGL(cxp_k_dedupe_resolve, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, (u64*)S->tkeys.p, tsz - 1, (const uint8_t*)involved);
        CXP_HIP(ctx, hipMemsetAsync(misc + 4, 0, 2 * sizeof(uint32_t), st));
        hipLaunchKernelGGL(cxp_k_count_alive, dim3(std::min(cxp_blocks(nt), 1024u)), dim3(256), 0, st, alive, nt, misc + 4);
        if (smooth > 0.0) {
            // ---- smooth_interpolations (tetrahedral.py:547-550), between the weld and the tiny collapse
            const u64 esz = cxp_table_size((size_t)nt * 3);
            if ((rc = cxp_reserve(ctx, S->tkeys, esz * sizeof(u64)))) return rc;
            if ((rc = cxp_reserve(ctx, S->pts_out, (size_t)(nv + 1) * 3 * sizeof(double)))) return rc;
            if ((rc = cxp_reserve(ctx, S->flags, (size_t)(nv + nt + 16) * sizeof(uint32_t)))) return rc;
            double* sum = (double*)S->pts_out.p;
            uint32_t* cnt = (uint32_t*)S->flags.p;
            CXP_HIP(ctx, hipMemsetAsync(sum, 0, (size_t)nv * 3 * sizeof(double), st));
            CXP_HIP(ctx, hipMemsetAsync(cnt, 0, (size_t)nv * sizeof(uint32_t), st));
            hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, (u64*)S->tkeys.p, (size_t)esz, CXP_EMPTY);
            if (cx_debug_knob("CX_SMOOTH_TWO_STEP", 0)) {
                hipLaunchKernelGGL(cxp_k_unique_edges, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, (u64*)S->tkeys.p, esz - 1);
                hipLaunchKernelGGL(cxp_k_smooth_accumulate, dim3(cxp_blocks(esz)), dim3(256), 0, st, (const u64*)S->tkeys.p, (size_t)esz, pts, sum, cnt);
            } else {
                hipLaunchKernelGGL(cxp_k_smooth_edges, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, (u64*)S->tkeys.p, esz - 1, (const double*)pts, sum, cnt);
            }
            hipLaunchKernelGGL(cxp_k_smooth_apply, dim3(cxp_blocks(nv)), dim3(256), 0, st, pts, sum, cnt, nv, smooth);
        }
        // ---- tiny collapse (tetrahedral.py:353-375), epsilon = 1e-4, scaled by 1/corner
        u64* parent = (u64*)S->parent.p;
        hipLaunchKernelGGL(cxp_k_iota64, dim3(cxp_blocks(nv)), dim3(256), 0, st, parent, nv);
        CXP_HIP(ctx, hipMemsetAsync(moved, 0, nv, st));
        hipLaunchKernelGGL(cxp_k_tiny, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, alive, nt, pts, 1.0 / corner[0], 1.0 / corner[1],
                           1.0 / corner[2], 1e-4, parent, prio, moved);
        // roots keep their own coordinates, so moving members in place is race free
        hipLaunchKernelGGL(cxp_k_move, dim3(cxp_blocks(nv)), dim3(256), 0, st, pts, pts, parent, moved, nv);
        hipLaunchKernelGGL(cxp_k_count_alive, dim3(std::min(cxp_blocks(nt), 1024u)), dim3(256), 0, st, alive, nt, misc + 5);
        uint32_t h[2];
        CXP_HIP(ctx, hipMemcpyAsync(h, misc + 4, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        CXP_HIP(ctx, hipStreamSynchronize(st));
        counts[2] = h[0]; counts[3] = h[1];
    }
    // (`moved` is free again after cxp_k_move)
    return cxp_clean_orient(ctx, S, nv, nt, do_clean, true, tprio3, counts, coherent,
                            (coherent && nv && nt && !cx_debug_knob("CX_DEDUPE_ALL", 0)) ? moved : nullptr, shard, ever);
}

extern "C" int cx_postprocess3d_ex(cx_ctx* ctx, uint32_t flags, double smooth, int64_t* out_counts) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->extracted) { ctx->err = "cx_postprocess3d: no valid extraction"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S;
    int rc = cxp_state(ctx, &S);
    if (rc) return rc;
    const uint32_t nv = (uint32_t)ctx->counts.n_vertices, nt = (uint32_t)ctx->counts.n_triangles;
    hipStream_t st = ctx->stream;
    int64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((rc = cxp_reserve3d(ctx, S, nv, nt))) return rc;
    const cx_params& P = ctx->last;
    const uint8_t* vkeep = nullptr;
    // an array with a rim of samples around the reference's grid starts at a NEGATIVE lattice point (cx_set_origin):
    // the post-pass then works in the reference's own lattice coordinates, so that the weld buckets truncate as the
    // reference's do (towards zero) and the Level-1 points come back in its coordinates.  Slabs (origin >= 0) stay local.
    cxp_origin3 org{{0.0, 0.0, 0.0}};
    if (ctx->origin[0] < 0 || ctx->origin[1] < 0 || ctx->origin[2] < 0)
        for (int a = 0; a < 3; a++) org.o[a] = (double)ctx->origin[a];
    if (nv && nt) {
        hipLaunchKernelGGL(cxp_k_vertices_f64, dim3(cxp_blocks(nv)), dim3(256), 0, st, P.grid, ctx->grid64_valid ? ctx->grid64 : nullptr, P.n1, P.n2,
                           P.div_plane, P.div_row, P.value, ctx->verts, nv, (double*)S->pts.p, (uint32_t*)S->prio.p, org);
        CXP_HIP(ctx, hipMemcpyAsync(S->tri.p, ctx->tris, (size_t)nt * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        if (ctx->keep_valid) {   // cx_select_seeded3d: only the triangles (and vertices) of the selected components exist
            CXP_HIP(ctx, hipMemcpyAsync(S->alive.p, ctx->tri_keep, nt, hipMemcpyDeviceToDevice, st));
            vkeep = ctx->tri_keep + nt;
        } else {
            CXP_HIP(ctx, hipMemsetAsync(S->alive.p, 1, nt, st));
        }
    }
    // the reference's `corner` (voxels per axis); a sample array with a margin around the reference's grid
    // carries the reference's own corner in ctx->corner_ref (cx_set_reference_corner)
    double corner[3] = {(double)(P.n0 - 1), (double)(P.n1 - 1), (double)(P.n2 - 1)};
    for (int a = 0; a < 3; a++)
        if (ctx->corner_ref[a] > 0) corner[a] = (double)ctx->corner_ref[a];
    if ((rc = cxp_run3d(ctx, S, nv, nt, corner, vkeep, !(flags & 1u), smooth, true, counts, true))) return rc;   // the march winds every triangle low -> high
    ctx->post_valid = true;
    if (out_counts) memcpy(out_counts, counts, sizeof(counts));
    return CX_OK;
}

// ---- sharded Level 1 (the reference is one process: tetrahedral.py:190-215, 353-375, surface_geometry.py:14-140) ----
// The context holds the extraction of a slab that was marched TOGETHER with two layers of cells of each neighbour slab
// (cx_set_origin = where the local array starts in the whole volume, cx_set_reference_corner = the whole volume's corner).
// Weld buckets are narrower than a cell and never straddle an integer plane (tetrahedral.py:192-196), tiny and degenerate
// triangles chain around one lattice point: with two layers, everything this step decides about the slab's own cells and
// about the FIRST layer of its neighbours is what the undivided volume decides.  Components and the max-x rule are global:
// begin() labels the components of own + first-layer triangles and leaves, for the host to exchange, the first-layer and
// own-boundary triangles with their labels and one start-triangle candidate per component that reaches them (sizes follow the
// slab boundary); finish() takes the agreed flips for those components.  own_lo / own_hi: own cell layers of the local array.
extern "C" int cx_postprocess3d_shard_begin(cx_ctx* ctx, uint32_t flags, int64_t own_lo, int64_t own_hi, int64_t* out_counts,
                                            int64_t* n_own_lower, int64_t* n_upper_copies, int64_t* n_candidates) {
    if (!ctx || !n_own_lower || !n_upper_copies || !n_candidates) return CX_ERR_INVALID;
    if (!ctx->extracted) { ctx->err = "cx_postprocess3d_shard_begin: no valid extraction"; return CX_ERR_STATE; }
    const cx_params& P = ctx->last;
    if (own_lo < 0 || own_hi <= own_lo || own_hi > (int64_t)P.n0 - 1) { ctx->err = "cx_postprocess3d_shard_begin: own cell layers outside the local array"; return CX_ERR_INVALID; }
    if (ctx->keep_valid) { ctx->err = "cx_postprocess3d_shard_begin: not after a seeded selection"; return CX_ERR_STATE; }
    if (ctx->origin[0] < 0) { ctx->err = "cx_postprocess3d_shard_begin: the local array starts before the volume (cx_set_origin)"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S;
    int rc = cxp_state(ctx, &S);
    if (rc) return rc;
    const uint32_t nv = (uint32_t)ctx->counts.n_vertices, nt = (uint32_t)ctx->counts.n_triangles;
    hipStream_t st = ctx->stream;
    int64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((rc = cxp_reserve3d(ctx, S, nv, nt))) return rc;
    // coordinates of the WHOLE volume, added to the lattice points before the interpolation: bit for bit what the undivided
    // volume computes, so that the weld buckets truncate identically
    const cxp_origin3 org{{(double)ctx->origin[0], (double)ctx->origin[1], (double)ctx->origin[2]}};
    if (nv && nt) {
        hipLaunchKernelGGL(cxp_k_vertices_f64, dim3(cxp_blocks(nv)), dim3(256), 0, st, P.grid, ctx->grid64_valid ? ctx->grid64 : nullptr, P.n1, P.n2,
                           P.div_plane, P.div_row, P.value, ctx->verts, nv, (double*)S->pts.p, (uint32_t*)S->prio.p, org);
        CXP_HIP(ctx, hipMemcpyAsync(S->tri.p, ctx->tris, (size_t)nt * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        CXP_HIP(ctx, hipMemsetAsync(S->alive.p, 1, nt, st));
    }
    double corner[3] = {(double)(P.n0 - 1), (double)(P.n1 - 1), (double)(P.n2 - 1)};
    for (int a = 0; a < 3; a++)
        if (ctx->corner_ref[a] > 0) corner[a] = (double)ctx->corner_ref[a];
    const cxp_shard sh{P.n1 * P.n2, (uint32_t)own_lo, (uint32_t)own_hi, P.n0 - 1};
    ctx->post_valid = false;
    if ((rc = cxp_run3d(ctx, S, nv, nt, corner, nullptr, !(flags & 1u), 0.0, true, counts, true, &sh))) return rc;
    if (out_counts) memcpy(out_counts, counts, sizeof(counts));
    *n_own_lower = S->shard.n1;
    *n_upper_copies = S->shard.n4;
    *n_candidates = S->shard.ncand;
    return CX_OK;
}

// One of the two boundary lists of cx_postprocess3d_shard_begin: which = 1, this slab's own triangles next to its LOWER neighbour
// (n_own_lower entries); which = 4, its copies of the UPPER neighbour's first layer (n_upper_copies).  Per triangle a 64-bit hash
// of its three original edge ids in the numbering of the whole volume, and its component label here.  Rank r's list 4 and rank
// r+1's list 1 hold the same triangles: sorted by hash they pair up the labels.  hash / label: DEVICE OR HOST memory (the lists
// are meant to stay on the device: a torch tensor sorts and sends them).
extern "C" int cx_postprocess3d_shard_boundary(cx_ctx* ctx, int which, uint64_t* hash, uint32_t* label) {
    if (!ctx || (which != 1 && which != 4)) return CX_ERR_INVALID;
    if (!ctx->post || !ctx->post->shard.open) { ctx->err = "cx_postprocess3d_shard_boundary: call cx_postprocess3d_shard_begin first"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S = ctx->post;
    const size_t n1 = S->shard.n1, n4 = S->shard.n4, nb = n1 + n4;
    const size_t n = which == 1 ? n1 : n4;
    if (!n) return CX_OK;
    if (!hash || !label) return CX_ERR_INVALID;
    const u64* hash1 = (const u64*)S->bnd.p;
    const cxp_cand* cand = (const cxp_cand*)(hash1 + nb);
    const uint32_t* label1 = (const uint32_t*)(cand + nb);
    CXP_HIP(ctx, hipMemcpyAsync(hash, which == 1 ? hash1 : hash1 + n1, n * sizeof(u64), hipMemcpyDefault, ctx->stream));
    CXP_HIP(ctx, hipMemcpyAsync(label, which == 1 ? label1 : label1 + n1, n * sizeof(uint32_t), hipMemcpyDefault, ctx->stream));
    CXP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CX_OK;
}

// the start-triangle candidates of cx_postprocess3d_shard_begin (host arrays of n_candidates entries): per component that reaches
// a neighbour its label and, among this slab's OWN triangles (surface_geometry.py:79-103): x of the max-x vertex, that vertex's
// edge id (local: add (origin_x * n1 * n2) << 3 for the whole volume's), |normal_x| of the start triangle, 1 if its normal_x is
// negative; has = 0 if the component has no own triangle here.
extern "C" int cx_postprocess3d_shard_candidates(cx_ctx* ctx, uint32_t* cand_label, double* cand_x, uint32_t* cand_vertex_key, double* cand_nx,
                                                 uint8_t* cand_negative, uint8_t* cand_has) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->post || !ctx->post->shard.open) { ctx->err = "cx_postprocess3d_shard_candidates: call cx_postprocess3d_shard_begin first"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S = ctx->post;
    const size_t nb = (size_t)S->shard.n1 + S->shard.n4, nc = S->shard.ncand;
    if (!nc) return CX_OK;
    if (!cand_label || !cand_x || !cand_vertex_key || !cand_nx || !cand_negative || !cand_has) return CX_ERR_INVALID;
    const cxp_cand* cand = (const cxp_cand*)((const u64*)S->bnd.p + nb);
    std::vector<cxp_cand> h(nc);
    CXP_HIP(ctx, hipMemcpyAsync(h.data(), cand, nc * sizeof(cxp_cand), hipMemcpyDeviceToHost, ctx->stream));
    CXP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < nc; i++) {
        cand_label[i] = h[i].label; cand_x[i] = h[i].x; cand_vertex_key[i] = h[i].vkey; cand_nx[i] = h[i].nx;
        cand_negative[i] = (uint8_t)h[i].sign; cand_has[i] = (uint8_t)h[i].has;
    }
    return CX_OK;
}

// flips[i] (0 / 1) for component labels[i] of cx_postprocess3d_shard_candidates: the decision of the rank that holds the component's
// start triangle.  Components that reach no neighbour keep the local decision.  Afterwards cx_level1_download /
// cx_level1_download_keys / cx_level1_write hand out this slab's OWN triangles and the vertices they use, in the coordinates
// of the whole volume.  out_counts as cx_postprocess3d ([0] vertices, [1] triangles, [4] components seen locally).
extern "C" int cx_postprocess3d_shard_finish(cx_ctx* ctx, const uint32_t* labels, const uint8_t* flips, int64_t n, int64_t* out_counts) {
    if (!ctx || n < 0 || (n && (!labels || !flips))) return CX_ERR_INVALID;
    if (!ctx->post || !ctx->post->shard.open) { ctx->err = "cx_postprocess3d_shard_finish: call cx_postprocess3d_shard_begin first"; return CX_ERR_STATE; }
    if (n > (int64_t)ctx->post->shard.nt2) { ctx->err = "cx_postprocess3d_shard_finish: more labels than triangles"; return CX_ERR_INVALID; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    int64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc = cxp_shard_finish(ctx, ctx->post, labels, flips, (uint32_t)n, counts);
    if (rc) return rc;
    ctx->post_valid = true;
    if (out_counts) memcpy(out_counts, counts, sizeof(counts));
    return CX_OK;
}

// edge id ((linear index of the lower lattice point << 3) | direction, local to the array that was marched) of every vertex of
// cx_level1_download, in its order: the identity of a Level-1 vertex across slabs and runs (the representative that survived
// weld, tiny collapse and clean-up)
extern "C" int cx_level1_download_keys(cx_ctx* ctx, uint32_t* keys) {
    if (!ctx || !keys) return CX_ERR_INVALID;
    if (!ctx->post || !ctx->post_valid || !ctx->post->keys_valid) { ctx->err = "cx_level1_download_keys: run cx_postprocess3d first"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S = ctx->post;
    if (!S->nv_out) return CX_OK;
    return cx_copy_to_host1(ctx, keys, S->keys_out.p, (size_t)S->nv_out * sizeof(uint32_t));
}

// float64 coordinates of the Level-0 vertices exactly as the reference interpolates them (tetrahedral.py:471-487) in the
// grid coordinates of the whole volume (the origin of cx_set_origin is added to the lattice points BEFORE the
// interpolation, so a slab yields bit for bit what the whole volume would), in the order of cx_level0_download -- what cx_postprocess3d works on, for callers that assemble the meshes of several slabs
extern "C" int cx_level0_points_f64(cx_ctx* ctx, double* points_xyz) {
    if (!ctx || !points_xyz) return CX_ERR_INVALID;
    if (!ctx->extracted) { ctx->err = "cx_level0_points_f64: no valid extraction"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S;
    int rc = cxp_state(ctx, &S);
    if (rc) return rc;
    const uint32_t nv = (uint32_t)ctx->counts.n_vertices;
    if (!nv) return CX_OK;
    if ((rc = cxp_reserve(ctx, S->pts, (size_t)(nv + 1) * 3 * sizeof(double)))) return rc;
    if ((rc = cxp_reserve(ctx, S->prio, (size_t)(nv + 1) * sizeof(uint32_t)))) return rc;
    const cx_params& P = ctx->last;
    const cxp_origin3 org{{(double)ctx->origin[0], (double)ctx->origin[1], (double)ctx->origin[2]}};
    hipLaunchKernelGGL(cxp_k_vertices_f64, dim3(cxp_blocks(nv)), dim3(256), 0, ctx->stream, P.grid, ctx->grid64_valid ? ctx->grid64 : nullptr, P.n1, P.n2,
                       P.div_plane, P.div_row, P.value, ctx->verts, nv, (double*)S->pts.p, (uint32_t*)S->prio.p, org);
    if ((rc = cx_copy_to_host1(ctx, points_xyz, S->pts.p, (size_t)nv * 3 * sizeof(double)))) return rc;
    ctx->post_valid = false;   // the post-pass buffers no longer hold a Level-1 mesh
    return CX_OK;
}

// Level 1 of a Level-0 mesh handed over by the caller, e.g. assembled from the slabs of several GPUs: vertices in
// ASCENDING EDGE-ID ORDER (their index is their priority) with the coordinates of cx_level0_points_f64 in the grid
// coordinates of the whole volume, triangles as indices wound as the march wound them.
extern "C" int cx_postprocess3d_mesh(cx_ctx* ctx, const double* points_xyz, int64_t nv64, const int32_t* tris, int64_t nt64, const int64_t* corner3,
                                     uint32_t flags, double smooth, int64_t* out_counts) {
    if (!ctx || !corner3 || nv64 < 0 || nt64 < 0 || (nv64 && !points_xyz) || (nt64 && !tris)) return CX_ERR_INVALID;
    if (nv64 >= 0x7FFFFFFFLL || nt64 >= 0x7FFFFFFFLL || corner3[0] < 1 || corner3[1] < 1 || corner3[2] < 1) {
        ctx->err = "cx_postprocess3d_mesh: corner >= 1 per axis and fewer than 2^31 vertices / triangles";
        return CX_ERR_INVALID;
    }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S;
    int rc = cxp_state(ctx, &S);
    if (rc) return rc;
    const uint32_t nv = (uint32_t)nv64, nt = (uint32_t)nt64;
    {   // a bad index would be a fault on the device (smallest and largest index: a loop without an exit, which the compiler vectorises;
        // with the comparison and its return inside, 67 M iterations took their 10 ms one by one)
        int32_t lo = 0, hi = -1;
        for (int64_t n = 0; n < nt64 * 3; n++) { lo = tris[n] < lo ? tris[n] : lo; hi = tris[n] > hi ? tris[n] : hi; }
        if (lo < 0 || (int64_t)hi >= nv64) { ctx->err = "cx_postprocess3d_mesh: triangle index out of range"; return CX_ERR_INVALID; }
    }
    hipStream_t st = ctx->stream;
    int64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((rc = cxp_reserve3d(ctx, S, nv, nt))) return rc;
    if (nv && nt) {
        CXP_HIP(ctx, hipMemcpyAsync(S->pts.p, points_xyz, (size_t)nv * 3 * sizeof(double), hipMemcpyHostToDevice, st));
        CXP_HIP(ctx, hipMemcpyAsync(S->tri.p, tris, (size_t)nt * 3 * sizeof(int32_t), hipMemcpyHostToDevice, st));
        CXP_HIP(ctx, hipMemsetAsync(S->alive.p, 1, nt, st));
        hipLaunchKernelGGL(cxp_k_iota_prio, dim3(cxp_blocks(nv)), dim3(256), 0, st, (uint32_t*)S->prio.p, nv);
    }
    const double corner[3] = {(double)corner3[0], (double)corner3[1], (double)corner3[2]};
    if ((rc = cxp_run3d(ctx, S, nv, nt, corner, nullptr, !(flags & 1u), smooth, !(flags & 4u), counts, false, nullptr, (flags & 8u) != 0u))) return rc;
    ctx->post_valid = true;
    if (out_counts) memcpy(out_counts, counts, sizeof(counts));
    return CX_OK;
}

extern "C" int cx_level1_download(cx_ctx* ctx, double* points_xyz, int32_t* tris) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->post || !ctx->post_valid) { ctx->err = "cx_level1_download: run cx_postprocess3d first"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S = ctx->post;
    void* d[2] = {(points_xyz && S->nv_out) ? (void*)points_xyz : nullptr, (tris && S->nt_out) ? (void*)tris : nullptr};
    const void* sp[2] = {S->pts_out.p, S->tri_out.p};
    const size_t nb[2] = {(size_t)S->nv_out * 3 * sizeof(double), (size_t)S->nt_out * 3 * sizeof(int32_t)};
    return cx_copy_to_host(ctx, 2, d, sp, nb);   // pinned, double-buffered, several host threads (cx_xfer.hip)
}

// device pointers of the Level-1 mesh: for consumers that live on the GPU (a torch tensor, a renderer's vertex buffer) the 541 MB
// download of a 512^3 mesh -- 60 % of the API path's time -- never has to happen.  Everything enqueued on the context's stream
// is complete when this returns.
extern "C" int cx_level1_device_ptrs(cx_ctx* ctx, void** points_xyz, void** tris, int64_t* n_vertices, int64_t* n_triangles) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->post || !ctx->post_valid) { ctx->err = "cx_level1_device_ptrs: run cx_postprocess3d first"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    CXP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cx_post_state* S = ctx->post;
    if (points_xyz) *points_xyz = S->nv_out ? S->pts_out.p : nullptr;
    if (tris) *tris = S->nt_out ? S->tri_out.p : nullptr;
    if (n_vertices) *n_vertices = (int64_t)S->nv_out;
    if (n_triangles) *n_triangles = (int64_t)S->nt_out;
    return CX_OK;
}

// ---- binary mesh files straight from the Level-1 device buffers (what every caller of the reference does next,
// html_demo.py:118-161, without the detour through Python arrays).  The file's records are laid out ON THE DEVICE, a chunk at a
// time (world coordinates = grid * delta + mins, rounded as numpy rounds them: no fused multiply-add), and streamed through two
// pinned staging buffers: the copy of chunk i+1 runs while chunk i is written to the file.
__global__ void cxw_k_points(const double* __restrict__ pts, uint32_t first, uint32_t n, double m0, double m1, double m2, double d0, double d1,
                             double d2, int as_f32, void* out) {
#pragma clang fp contract(off)   // grid * delta + mins with two roundings, as numpy computes it (a fused multiply-add differs in the last bit)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = pts + (size_t)(first + i) * 3;
    const double x = p[0] * d0 + m0, y = p[1] * d1 + m1, z = p[2] * d2 + m2;
    if (as_f32) {
        float* o = reinterpret_cast<float*>(out) + (size_t)i * 3;
        o[0] = (float)x; o[1] = (float)y; o[2] = (float)z;
    } else {
        double* o = reinterpret_cast<double*>(out) + (size_t)i * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}
// PLY faces: uchar 3 + three little-endian int32 = 13 bytes per triangle
__global__ void cxw_k_faces13(const int32_t* __restrict__ tri, uint32_t first, uint32_t n, uint8_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t* o = out + (size_t)i * 13;
    o[0] = 3;
    for (int k = 0; k < 3; k++) {
        const uint32_t v = (uint32_t)tri[(size_t)(first + i) * 3 + k];
        o[1 + 4 * k] = (uint8_t)v; o[2 + 4 * k] = (uint8_t)(v >> 8); o[3 + 4 * k] = (uint8_t)(v >> 16); o[4 + 4 * k] = (uint8_t)(v >> 24);
    }
}
#include <cstdio>
extern "C" int cx_level1_write(cx_ctx* ctx, int format, const char* path, const double* mins_delta, double* out_info) {
    if (!ctx || !path || (format != CX_FILE_PLY && format != CX_FILE_GLTF_BIN)) return CX_ERR_INVALID;
    if (!ctx->post || !ctx->post_valid) { ctx->err = "cx_level1_write: run cx_postprocess3d first"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S = ctx->post;
    hipStream_t st = ctx->stream;
    const uint32_t nv = (uint32_t)S->nv_out, nt = (uint32_t)S->nt_out;
    double m[3] = {0, 0, 0}, d[3] = {1, 1, 1};
    if (mins_delta) for (int a = 0; a < 3; a++) { m[a] = mins_delta[a]; d[a] = mins_delta[3 + a]; }
    const uint32_t CHUNK = 1u << 20;                       // elements per chunk
    const size_t stage_bytes = (size_t)CHUNK * 24u;         // the widest record: 3 doubles
    uint8_t* dstage = nullptr;
    uint8_t* hstage[2] = {nullptr, nullptr};
    FILE* f = fopen(path, "wb");
    if (!f) { ctx->err = std::string("cx_level1_write: cannot open ") + path; return CX_ERR_INVALID; }
    int rc = CX_OK;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    size_t written = 0;
    do {
        hipError_t e;
#define CXW_TRY(call) if ((e = (call)) != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e); rc = (e == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP; break; }
        CXW_TRY(hipMalloc(&dstage, 2 * stage_bytes));
        CXW_TRY(hipHostMalloc(&hstage[0], stage_bytes));
        CXW_TRY(hipHostMalloc(&hstage[1], stage_bytes));
        if (format == CX_FILE_PLY) {
            char header[512];
            const int hl = snprintf(header, sizeof(header),
                                    "ply\nformat binary_little_endian 1.0\ncomment contourist_amd isosurface\nelement vertex %u\n"
                                    "property double x\nproperty double y\nproperty double z\n"
                                    "element face %u\nproperty list uchar int vertex_indices\nend_header\n", nv, nt);
            if (fwrite(header, 1, (size_t)hl, f) != (size_t)hl) { rc = CX_ERR_INVALID; ctx->err = "cx_level1_write: write failed"; break; }
            written += (size_t)hl;
        }
        // section 0: points, section 1: faces / indices.  Chunk c of a section is prepared on the device into half c & 1 of
        // dstage and copied to hstage[c & 1]; the previous chunk is written to the file meanwhile.
        for (int section = 0; section < 2 && rc == CX_OK; section++) {
            const uint32_t count = section == 0 ? nv : nt;
            const size_t rec = section == 0 ? (format == CX_FILE_PLY ? 24u : 12u) : (format == CX_FILE_PLY ? 13u : 12u);
            size_t pending_bytes = 0;
            int pending = -1;
            for (uint32_t first = 0, c = 0; rc == CX_OK; first += CHUNK, c++) {
                const bool more = first < count;
                const uint32_t n = more ? std::min(CHUNK, count - first) : 0u;
                const int half = (int)(c & 1u);
                if (more) {
                    uint8_t* dst = dstage + (size_t)half * stage_bytes;
                    if (section == 0)
                        hipLaunchKernelGGL(cxw_k_points, dim3(cxp_blocks(n)), dim3(256), 0, st, (const double*)S->pts_out.p, first, n, m[0], m[1], m[2],
                                           d[0], d[1], d[2], format == CX_FILE_PLY ? 0 : 1, (void*)dst);
                    else if (format == CX_FILE_PLY)
                        hipLaunchKernelGGL(cxw_k_faces13, dim3(cxp_blocks(n)), dim3(256), 0, st, (const int32_t*)S->tri_out.p, first, n, dst);
                    const void* src = (section == 1 && format != CX_FILE_PLY) ? (const void*)((const int32_t*)S->tri_out.p + (size_t)first * 3) : (const void*)dst;
                    e = hipMemcpyAsync(hstage[half], src, (size_t)n * rec, hipMemcpyDeviceToHost, st);
                    if (e != hipSuccess) { ctx->err = std::string("hipMemcpyAsync: ") + hipGetErrorString(e); rc = CX_ERR_HIP; break; }
                }
                if (pending >= 0) {   // the previous chunk is complete (synchronised below, last iteration); the one just enqueued copies meanwhile
                    if (fwrite(hstage[pending], 1, pending_bytes, f) != pending_bytes) { rc = CX_ERR_INVALID; ctx->err = "cx_level1_write: write failed"; break; }
                    written += pending_bytes;
                }
                if (!more) break;
                e = hipStreamSynchronize(st);
                if (e != hipSuccess) { ctx->err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); rc = CX_ERR_HIP; break; }
                if (section == 0) {   // bounds of the positions (glTF needs them in its JSON; reported for both formats)
                    if (format != CX_FILE_PLY) {
                        const float* q = reinterpret_cast<const float*>(hstage[half]);
                        for (uint32_t i = 0; i < n; i++)
                            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)q[3 * i + a]); hi[a] = std::max(hi[a], (double)q[3 * i + a]); }
                    } else {
                        const double* q = reinterpret_cast<const double*>(hstage[half]);
                        for (uint32_t i = 0; i < n; i++)
                            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], q[3 * i + a]); hi[a] = std::max(hi[a], q[3 * i + a]); }
                    }
                }
                pending = half;
                pending_bytes = (size_t)n * rec;
            }
        }
#undef CXW_TRY
    } while (0);
    // a short write may only surface when the buffered bytes are flushed (disk full): check both, and leave no partial file
    if (fflush(f) != 0 && rc == CX_OK) { rc = CX_ERR_INVALID; ctx->err = "cx_level1_write: write failed (flush)"; }
    if (fclose(f) != 0 && rc == CX_OK) { rc = CX_ERR_INVALID; ctx->err = "cx_level1_write: write failed (close)"; }
    if (dstage) (void)hipFree(dstage);
    for (int k = 0; k < 2; k++)
        if (hstage[k]) (void)hipHostFree(hstage[k]);
    if (rc) { (void)remove(path); return rc; }
    if (out_info) {
        out_info[0] = (double)nv; out_info[1] = (double)nt; out_info[2] = (double)written;
        for (int a = 0; a < 3; a++) { out_info[3 + a] = nv ? lo[a] : 0.0; out_info[6 + a] = nv ? hi[a] : 0.0; }
    }
    return CX_OK;
}

// SurfaceGeometry(vertices, triangles) on a caller's mesh.  mode: 0 = orient only, 1 = clean + orient, 2 = clean only.
extern "C" int cx_surface_geometry(cx_ctx* ctx, double* points_xyz, int64_t* nv_io, int32_t* tris, int64_t* nt_io, int mode) {
    if (!ctx || !points_xyz || !tris || !nv_io || !nt_io || mode < 0 || mode > 2) return CX_ERR_INVALID;
    if (*nv_io < 0 || *nt_io < 0 || *nv_io > 0x7FFFFFF0LL || *nt_io > 0x7FFFFFF0LL) return CX_ERR_INVALID;
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S;
    int rc = cxp_state(ctx, &S);
    if (rc) return rc;
    const uint32_t nv = (uint32_t)*nv_io, nt = (uint32_t)*nt_io;
    for (int64_t n = 0; n < (int64_t)nt * 3; n++)
        if (tris[n] < 0 || tris[n] >= (int64_t)nv) { ctx->err = "cx_surface_geometry: triangle index out of range"; return CX_ERR_INVALID; }
    hipStream_t st = ctx->stream;
    if ((rc = cxp_reserve(ctx, S->pts, (size_t)(nv + 1) * 3 * sizeof(double)))) return rc;
    if ((rc = cxp_reserve(ctx, S->prio, (size_t)(nv + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = cxp_reserve(ctx, S->tri, (size_t)(nt + 1) * 3 * sizeof(int32_t) * 2))) return rc;
    if ((rc = cxp_reserve(ctx, S->alive, (size_t)nt + 16))) return rc;
    if ((rc = cxp_reserve(ctx, S->parent2, (size_t)(nv + 1) * sizeof(u64)))) return rc;
    int32_t* tri = (int32_t*)S->tri.p;
    uint32_t* tprio3 = (uint32_t*)(tri + (size_t)(nt + 1) * 3);
    if (nv) CXP_HIP(ctx, hipMemcpyAsync(S->pts.p, points_xyz, (size_t)nv * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    if (nt) CXP_HIP(ctx, hipMemcpyAsync(tri, tris, (size_t)nt * 3 * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (nv) hipLaunchKernelGGL(cxp_k_iota_prio, dim3(cxp_blocks(nv)), dim3(256), 0, st, (uint32_t*)S->prio.p, nv);
    if (nt) {
        CXP_HIP(ctx, hipMemsetAsync(S->alive.p, 1, nt, st));
        hipLaunchKernelGGL(cxp_k_tri_prio, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, (const uint32_t*)S->prio.p, nt, tprio3);
        // a caller's triangle may repeat a vertex: such rows are not triangles (surface_geometry.py:63)
        hipLaunchKernelGGL(cxp_k_remap, dim3(cxp_blocks(nt)), dim3(256), 0, st, tri, (uint8_t*)S->alive.p, nt, (const uint32_t*)S->prio.p,
                           (const u64*)nullptr);
    }
    int64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((rc = cxp_clean_orient(ctx, S, nv, nt, mode != 0, mode != 2, tprio3, counts, false))) return rc;   // caller's windings: arbitrary
    if (S->nv_out) CXP_HIP(ctx, hipMemcpyAsync(points_xyz, S->pts_out.p, (size_t)S->nv_out * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (S->nt_out) CXP_HIP(ctx, hipMemcpyAsync(tris, S->tri_out.p, (size_t)S->nt_out * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    CXP_HIP(ctx, hipStreamSynchronize(st));
    *nv_io = S->nv_out; *nt_io = S->nt_out;
    ctx->post_valid = false;
    return CX_OK;
}


// =====================================================================================================
// 4-D: the post-steps of GridContour4D.find_tetrahedra
//   bin_times(nbins=100)            pentatopes.py:162-169
//   drop_instant_tetrahedra(1e-7)   pentatopes.py:171-189
//   remove_tiny_simplices(1e-3)     tetrahedral.py:353-375 (called at pentatopes.py:125)
// =====================================================================================================
// float64 4-D vertex coordinates exactly as the reference interpolates them, with t snapped to its bin
__global__ void cxp_k_vertices4_f64(const float* __restrict__ A, uint32_t n1, uint32_t n2, uint32_t n3, cx_fdiv d3, cx_fdiv d2, cx_fdiv d1,
                                    double value, const uint32_t* __restrict__ vkeys, uint32_t nv, double min_interval, double* pts,
                                    uint32_t* prio, int o0, int o1, int o2, int o3) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const uint32_t key = vkeys[v];
    const uint32_t lin = key >> 4, d = key & 15u;
    uint32_t q[4];
    q[0] = cx_div(lin, d3);
    uint32_t r = lin - q[0] * (n1 * n2 * n3);
    q[1] = cx_div(r, d2);
    r -= q[1] * (n2 * n3);
    q[2] = cx_div(r, d1);
    q[3] = r - q[2] * n3;
    const uint32_t lin2 = lin + ((d & 8u) ? n1 * n2 * n3 : 0u) + ((d & 4u) ? n2 * n3 : 0u) + ((d & 2u) ? n3 : 0u) + (d & 1u);
    const double f0 = (double)A[lin], f1 = (double)A[lin2];
    const bool owner_low = !(f0 > f1);
    const double flow = owner_low ? f0 : f1, fhigh = owner_low ? f1 : f0;
    double ratio = 0.5;
    const double den = 1.0 * (fhigh - flow);
    if (!(fabs(den) <= 1e-8)) ratio = (value - flow) / den;
    const uint32_t db[4] = {(d >> 3) & 1u, (d >> 2) & 1u, (d >> 1) & 1u, d & 1u};
    const int org[4] = {o0, o1, o2, o3};   // (negative) origin of the array in the reference's lattice: the points are interpolated there
    double x[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const double qa = (double)((int)q[a] + org[a]);
        const double low = owner_low ? qa : qa + (double)db[a];
        const double high = owner_low ? qa + (double)db[a] : qa;
        x[a] = low + ratio * (high - low);
    }
    // bin_times: bin = int(t / min_interval); t = bin * min_interval
    x[3] = (double)(long long)(x[3] / min_interval) * min_interval;
#pragma unroll
    for (int a = 0; a < 4; a++) pts[(size_t)v * 4 + a] = x[a];
    prio[v] = key;
}

// the same for points handed over by the caller (linear_interpolate=False: refined on the host with the caller's function,
// tetrahedral.py:488-505): only bin_times and the priority word
__global__ void cxp_k_vertices4_given(const uint32_t* __restrict__ vkeys, uint32_t nv, double min_interval, double* pts, uint32_t* prio) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const double t = pts[(size_t)v * 4 + 3];
    pts[(size_t)v * 4 + 3] = (double)(long long)(t / min_interval) * min_interval;
    prio[v] = vkeys[v];
}

__global__ void cxp_k_and_mask(uint8_t* alive, const uint8_t* keep, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && !keep[t]) alive[t] = 0;
}
__global__ void cxp_k_drop_instant(const int32_t* tets, uint8_t* alive, uint32_t nt, const double* pts, double eps) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    double lo = pts[(size_t)tets[(size_t)t * 4] * 4 + 3], hi = lo;
#pragma unroll
    for (int s = 1; s < 4; s++) {
        const double x = pts[(size_t)tets[(size_t)t * 4 + s] * 4 + 3];
        lo = fmin(lo, x); hi = fmax(hi, x);
    }
    alive[t] = ((hi - lo) < eps) ? 0 : 1;
}

__global__ void cxp_k_tiny4(const int32_t* tets, uint8_t* alive, uint32_t nt, const double* pts, double ic0, double ic1, double ic2,
                            double ic3, double eps, u64* parent, const uint32_t* prio, uint8_t* moved) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
    const double ic[4] = {ic0, ic1, ic2, ic3};
    uint32_t v[4];
#pragma unroll
    for (int s = 0; s < 4; s++) v[s] = (uint32_t)tets[(size_t)t * 4 + s];
    double worst = 0.0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        double lo = pts[(size_t)v[0] * 4 + a], hi = lo;
#pragma unroll
        for (int s = 1; s < 4; s++) {
            const double x = pts[(size_t)v[s] * 4 + a];
            lo = fmin(lo, x); hi = fmax(hi, x);
        }
        worst = fmax(worst, (hi - lo) * ic[a]);
    }
    if (worst < eps) {
        alive[t] = 0;
#pragma unroll
        for (int s = 1; s < 4; s++) cxp_union(parent, prio, v[0], v[s], 0);
#pragma unroll
        for (int s = 0; s < 4; s++) moved[v[s]] = 1;
    }
}
__global__ void cxp_k_move4(double* pts, const u64* parent, const uint8_t* moved, uint32_t nv) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv || !moved[v]) return;
    uint32_t par;
    const uint32_t r = cxp_find(parent, v, par);
    if (r == v) return;
#pragma unroll
    for (int a = 0; a < 4; a++) pts[(size_t)v * 4 + a] = pts[(size_t)r * 4 + a];
}
__global__ void cxp_k_compact_tets(const int32_t* tets, const uint8_t* alive, const uint32_t* tnew, uint32_t nt, int32_t* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !alive[t]) return;
#pragma unroll
    for (int s = 0; s < 4; s++) out[(size_t)tnew[t] * 4 + s] = tets[(size_t)t * 4 + s];
}

extern "C" int cx_postprocess4d_points(cx_ctx* ctx, int32_t nbins, const double* points_xyzt, int64_t* out_counts);
extern "C" int cx_postprocess4d(cx_ctx* ctx, int32_t nbins, int64_t* out_counts) {
    return cx_postprocess4d_points(ctx, nbins, nullptr, out_counts);
}
extern "C" int cx_postprocess4d_points(cx_ctx* ctx, int32_t nbins, const double* points_xyzt, int64_t* out_counts) {
    if (!ctx || nbins <= 0) return CX_ERR_INVALID;
    cx_state4* G = ctx->s4;
    if (!G || !G->extracted) { ctx->err = "cx_postprocess4d: no valid 4-D extraction"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S;
    int rc = cxp_state(ctx, &S);
    if (rc) return rc;
    const uint32_t nv = (uint32_t)G->counts.n_vertices, nt = (uint32_t)G->counts.n_triangles;
    hipStream_t st = ctx->stream;
    if ((rc = cxp_reserve(ctx, S->pts, (size_t)(nv + 1) * 4 * sizeof(double)))) return rc;
    if ((rc = cxp_reserve(ctx, S->prio, (size_t)(nv + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = cxp_reserve(ctx, S->rep, (size_t)nv + 16))) return rc;
    if ((rc = cxp_reserve(ctx, S->alive, (size_t)nt + 16))) return rc;
    if ((rc = cxp_reserve(ctx, S->parent, (size_t)(nv + 1) * sizeof(u64)))) return rc;
    if ((rc = cxp_reserve(ctx, S->flags, (size_t)(nt + 16) * sizeof(uint32_t)))) return rc;
    if ((rc = cxp_reserve(ctx, S->scan, (size_t)(nt + 16) * sizeof(uint32_t)))) return rc;
    double* pts = (double*)S->pts.p;
    uint32_t* prio = (uint32_t*)S->prio.p;
    uint8_t* moved = (uint8_t*)S->rep.p;
    uint8_t* alive = (uint8_t*)S->alive.p;
    u64* parent = (u64*)S->parent.p;
    uint32_t* misc = (uint32_t*)S->misc.p;
    int64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t nt2 = 0;
    if (nv && nt) {
        const uint32_t n1 = (uint32_t)G->n[1], n2 = (uint32_t)G->n[2], n3 = (uint32_t)G->n[3];
        // an array with a rim of samples around the reference's grid (negative origin, cx_set_origin4d): the reference's own
        // corner and lattice (scales of the tiny collapse, bin width, coordinates)
        int org[4];
        double corner[4];
        for (int a = 0; a < 4; a++) {
            org[a] = (G->origin[a] < 0) ? (int)G->origin[a] : 0;
            corner[a] = (double)(G->n[a] - 1 + 2 * org[a]);
        }
        const double min_interval = corner[3] * (1.0 / (double)nbins);
        if (points_xyzt) {   // the caller's points (in the reference's lattice), one per Level-0 vertex
            CXP_HIP(ctx, hipMemcpyAsync(pts, points_xyzt, (size_t)nv * 4 * sizeof(double), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(cxp_k_vertices4_given, dim3(cxp_blocks(nv)), dim3(256), 0, st, G->vkeys, nv, min_interval, pts, prio);
        } else
        hipLaunchKernelGGL(cxp_k_vertices4_f64, dim3(cxp_blocks(nv)), dim3(256), 0, st, G->grid, n1, n2, n3, cx_fdiv_make(n1 * n2 * n3),
                           cx_fdiv_make(n2 * n3), cx_fdiv_make(n3), G->value, G->vkeys, nv, min_interval, pts, prio, org[0], org[1], org[2], org[3]);
        hipLaunchKernelGGL(cxp_k_drop_instant, dim3(cxp_blocks(nt)), dim3(256), 0, st, G->tets, alive, nt, pts, 1e-7);
        // cx_select_seeded4d: only the tetrahedra of the selected components exist
        if (G->keep_valid) hipLaunchKernelGGL(cxp_k_and_mask, dim3(cxp_blocks(nt)), dim3(256), 0, st, alive, (const uint8_t*)G->tet_keep, nt);
        CXP_HIP(ctx, hipMemsetAsync(misc + 4, 0, 2 * sizeof(uint32_t), st));
        hipLaunchKernelGGL(cxp_k_count_alive, dim3(std::min(cxp_blocks(nt), 1024u)), dim3(256), 0, st, alive, nt, misc + 4);
        hipLaunchKernelGGL(cxp_k_iota64, dim3(cxp_blocks(nv)), dim3(256), 0, st, parent, nv);
        CXP_HIP(ctx, hipMemsetAsync(moved, 0, nv, st));
        hipLaunchKernelGGL(cxp_k_tiny4, dim3(cxp_blocks(nt)), dim3(256), 0, st, G->tets, alive, nt, pts, 1.0 / corner[0], 1.0 / corner[1],
                           1.0 / corner[2], 1.0 / corner[3], 1e-3, parent, prio, moved);
        hipLaunchKernelGGL(cxp_k_move4, dim3(cxp_blocks(nv)), dim3(256), 0, st, pts, parent, moved, nv);
        hipLaunchKernelGGL(cxp_k_count_alive, dim3(std::min(cxp_blocks(nt), 1024u)), dim3(256), 0, st, alive, nt, misc + 5);
        uint32_t* tflag = (uint32_t*)S->flags.p;
        uint32_t* tnew = (uint32_t*)S->scan.p;
        hipLaunchKernelGGL(cxp_k_alive_u32, dim3(cxp_blocks(nt)), dim3(256), 0, st, alive, nt, tflag);
        if ((rc = cxp_scan(ctx, S, tflag, tnew, nt, misc + 2))) return rc;
        uint32_t h[3];
        CXP_HIP(ctx, hipMemcpyAsync(h, misc + 4, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        CXP_HIP(ctx, hipMemcpyAsync(h + 2, misc + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        CXP_HIP(ctx, hipStreamSynchronize(st));
        counts[2] = h[0]; counts[3] = h[1];
        nt2 = h[2];
        if ((rc = cxp_reserve(ctx, S->tri_out, (size_t)(nt2 + 1) * 4 * sizeof(int32_t)))) return rc;
        hipLaunchKernelGGL(cxp_k_compact_tets, dim3(cxp_blocks(nt)), dim3(256), 0, st, G->tets, alive, tnew, nt, (int32_t*)S->tri_out.p);
        CXP_HIP(ctx, hipGetLastError());
    }
    S->nv_out = nv; S->nt_out = nt2;
    counts[0] = nv; counts[1] = nt2;
    G->post_valid = true;
    if (out_counts) memcpy(out_counts, counts, sizeof(counts));
    return CX_OK;
}

extern "C" int cx_level1_4d_download(cx_ctx* ctx, double* points_xyzt, int32_t* tets) {
    if (!ctx) return CX_ERR_INVALID;
    cx_state4* G = ctx->s4;
    if (!G || !G->post_valid || !ctx->post) { ctx->err = "cx_level1_4d_download: run cx_postprocess4d first"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S = ctx->post;
    void* d[2] = {(points_xyzt && S->nv_out) ? (void*)points_xyzt : nullptr, (tets && S->nt_out) ? (void*)tets : nullptr};
    const void* sp[2] = {S->pts.p, S->tri_out.p};
    const size_t nb[2] = {(size_t)S->nv_out * 4 * sizeof(double), (size_t)S->nt_out * 4 * sizeof(int32_t)};
    return cx_copy_to_host(ctx, 2, d, sp, nb);
}


// =====================================================================================================
// 4-D rows B4 / B5: morph triangles
//   GridContour4D.collect_morph_triangles                     pentatopes.py:314-368
//   MorphGeometry.triangulate_tetrahedron_at_midpoints / add_tetrahedron / interpolate_pair_3d
//                                                             morph_geometry.py:145-237
//   MorphTriangles.__init__ / orient_triangles / compute_triangle_stats / time_compatible_triangles
//                                                             morph_geometry.py:7-22, 49-89
// Canonical numbering: the 4 vertices of a tetrahedron are ordered by edge id (the reference orders them by
// its dict numbering, which only decides how a 4-segment slice is split).
// =====================================================================================================
// (grid-stride, reduced per wave before the two read-before-atomic updates: one thread per point with its own pair of updates took
// 0.28 ms for 4.7 M points -- the first waves all raise the maximum)
__global__ __launch_bounds__(256) void cxp_k_minmax_t(const double* pts, uint32_t nv, u64* mm) {
    u64 lo = ~0ULL, hi = 0ULL;
    for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < nv; v += gridDim.x * 256u) {
        const u64 o = cxp_orderable(pts[(size_t)v * 4 + 3]);
        lo = o < lo ? o : lo;
        hi = o > hi ? o : hi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 l2 = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(lo >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)lo, o);
        const u64 h2 = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(hi >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63u) == 0u) {
        if (__hip_atomic_load(&mm[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > lo) atomicMin(&mm[0], lo);
        cxp_max64(&mm[1], hi);
    }
}
__device__ __forceinline__ double cxp_from_orderable(u64 o) {
    const u64 b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFULL) : ~o;
    return __longlong_as_double((long long)b);
}

// slices of one tetrahedron: calls emit(p0, p1, p2) with vertex-index pairs packed (i << 32 | j), i before j in
// canonical order; returns the number of triangles
template <typename Emit>
__device__ __forceinline__ uint32_t cxp_morph_slices(const int32_t* tets, uint32_t t, const double* pts, const uint32_t* prio, double t_eps,
                                                     Emit emit) {
    uint32_t v[4];
    double tv[4];
#pragma unroll
    for (int s = 0; s < 4; s++) v[s] = (uint32_t)tets[(size_t)t * 4 + s];
    // (a,b,c,d) = vertices in ascending priority (sorting network); the stored order is positively oriented
    // (cxp_k_tets_orient), tsign = orientation of (a,b,c,d)
    double tsign = 1.0;
#define CXP_CSWAP(x, y) if (prio[v[x]] > prio[v[y]]) { const uint32_t tmp = v[x]; v[x] = v[y]; v[y] = tmp; tsign = -tsign; }
    CXP_CSWAP(0, 1) CXP_CSWAP(2, 3) CXP_CSWAP(0, 2) CXP_CSWAP(1, 3) CXP_CSWAP(1, 2)
#undef CXP_CSWAP
#pragma unroll
    for (int s = 0; s < 4; s++) tv[s] = pts[(size_t)v[s] * 4 + 3];
    double ts[4] = {tv[0], tv[1], tv[2], tv[3]};
#define CXP_DSWAP(x, y) if (ts[x] > ts[y]) { const double tmp = ts[x]; ts[x] = ts[y]; ts[y] = tmp; }
    CXP_DSWAP(0, 1) CXP_DSWAP(2, 3) CXP_DSWAP(0, 2) CXP_DSWAP(1, 3) CXP_DSWAP(1, 2)
#undef CXP_DSWAP
    // Everything below lives in named scalars, one set per edge, and the crossed edges are picked by select chains from a 6-bit set: with
    // small arrays indexed by a running count `m` (or by select chains over array elements, which the compiler folds back into an indexed
    // load) the arrays went to scratch memory -- 160 bytes per lane, 108 registers, 4 waves per SIMD, cxp_k_morph_emit 1.65 ms on config 4.
    // Same arithmetic, same order of the edges: scan order (a,b)(a,c)(a,d)(b,c)(b,d)(c,d) = edges 0..5.
    uint32_t n = 0;
    const double wx = tv[1] - tv[0], wy = tv[2] - tv[0], wz = tv[3] - tv[0];
#define CXP_EDGE(e, I, J)                                                                                                   \
    const bool cross##e = !(mid + 1e-5 < fmin(tv[I], tv[J]) || mid - 1e-5 > fmax(tv[I], tv[J])); /* morph_geometry.py:218 */ \
    const u64 sp##e = ((u64)v[I] << 32) | (u64)v[J];                                                                         \
    const bool zero##e = fabs(tv[I] - tv[J]) <= t_eps;                                           /* pentatopes.py:341-345 */  \
    const double den##e = tv[J] - tv[I];                                                                                     \
    const double lam##e = (den##e != 0.0) ? fmin(1.0, fmax(0.0, (mid - tv[I]) / den##e)) : 0.5;                              \
    const double px##e = ((I) == 1 ? 1.0 : 0.0) + lam##e * (((J) == 1 ? 1.0 : 0.0) - ((I) == 1 ? 1.0 : 0.0));                \
    const double py##e = ((I) == 2 ? 1.0 : 0.0) + lam##e * (((J) == 2 ? 1.0 : 0.0) - ((I) == 2 ? 1.0 : 0.0));                \
    const double pz##e = ((I) == 3 ? 1.0 : 0.0) + lam##e * (((J) == 3 ? 1.0 : 0.0) - ((I) == 3 ? 1.0 : 0.0));
#define CXP_PICK(name, k) ((k) == 5 ? name##5 : (k) == 4 ? name##4 : (k) == 3 ? name##3 : (k) == 2 ? name##2 : (k) == 1 ? name##1 : name##0)
#pragma unroll
    for (int g = 0; g < 3; g++) {
        if (!((ts[g + 1] - ts[g]) > 1e-4)) continue;               // morph_geometry.py:150
        const double mid = 0.5 * (ts[g + 1] + ts[g]);
        // the slice points in the tetrahedron's own affine frame: a = 0, b = e1, c = e2, d = e3
        CXP_EDGE(0, 0, 1) CXP_EDGE(1, 0, 2) CXP_EDGE(2, 0, 3) CXP_EDGE(3, 1, 2) CXP_EDGE(4, 1, 3) CXP_EDGE(5, 2, 3)
        const uint32_t cm = (cross0 ? 1u : 0u) | (cross1 ? 2u : 0u) | (cross2 ? 4u : 0u) | (cross3 ? 8u : 0u) | (cross4 ? 16u : 0u) | (cross5 ? 32u : 0u);
        // winding: all tetrahedra are oriented alike with respect to the field gradient, so the slice triangle whose
        // normal (inside the tetrahedron) points towards later times is wound alike everywhere -- a rule on the
        // order of the vertex times only, which bin_times and the tiny collapse do not disturb
        auto wound = [&](int i0, int i1, int i2) {
            const double x0 = CXP_PICK(px, i0), y0 = CXP_PICK(py, i0), z0 = CXP_PICK(pz, i0);
            const double ax = CXP_PICK(px, i1) - x0, ay = CXP_PICK(py, i1) - y0, az = CXP_PICK(pz, i1) - z0;
            const double bx = CXP_PICK(px, i2) - x0, by = CXP_PICK(py, i2) - y0, bz = CXP_PICK(pz, i2) - z0;
            const double det = (ay * bz - az * by) * wx + (az * bx - ax * bz) * wy + (ax * by - ay * bx) * wz;
            const u64 s0 = CXP_PICK(sp, i0), s1 = CXP_PICK(sp, i1), s2 = CXP_PICK(sp, i2);
            if (det * tsign >= 0.0) emit(s0, s1, s2); else emit(s0, s2, s1);
        };
        const int m = __popc(cm);
        const int e0 = (int)__ffs(cm) - 1;                          // the first crossed edge in scan order
        if (m == 3) {
            const uint32_t c1 = cm & (cm - 1u), c2 = c1 & (c1 - 1u);
            const int e1 = (int)__ffs(c1) - 1, e2 = (int)__ffs(c2) - 1;
            if (!(CXP_PICK(zero, e0) || CXP_PICK(zero, e1) || CXP_PICK(zero, e2))) { wound(e0, e1, e2); n++; }
        } else if (m == 4) {                                       // morph_geometry.py:176-186
            // p2 = the crossed edge without a vertex in common with the first one (the last such one in scan order)
            const u64 s0 = CXP_PICK(sp, e0);
            const uint32_t a0 = (uint32_t)(s0 >> 32), a1 = (uint32_t)s0;
            int p2 = -1;
#define CXP_DISJ(e)                                                                                          \
            {                                                                                                \
                const uint32_t b0 = (uint32_t)(sp##e >> 32), b1 = (uint32_t)sp##e;                            \
                if (cross##e && e != e0 && a0 != b0 && a0 != b1 && a1 != b0 && a1 != b1) p2 = e;             \
            }
            CXP_DISJ(0) CXP_DISJ(1) CXP_DISJ(2) CXP_DISJ(3) CXP_DISJ(4) CXP_DISJ(5)
#undef CXP_DISJ
            if (p2 >= 0) {
                const bool z02 = CXP_PICK(zero, e0) || CXP_PICK(zero, p2);
#define CXP_THIRD(e) if (cross##e && e != e0 && e != p2 && !(z02 || zero##e)) { wound(e0, p2, e); n++; }
                CXP_THIRD(0) CXP_THIRD(1) CXP_THIRD(2) CXP_THIRD(3) CXP_THIRD(4) CXP_THIRD(5)
#undef CXP_THIRD
            }
        }
    }
#undef CXP_EDGE
#undef CXP_PICK
    return n;
}

// Every tetrahedron of the 4-D march lies in the level set of the linear interpolant of ONE pentatope (Kuhn simplex
// of its hypercube, pentatopes.py:15-26).  Its four indices are put in the order that makes
// det[p1-p0, p2-p0, p3-p0, gradient] positive, with the points as the march interpolated them (before bin_times) and
// the gradient of that interpolant: along the axis added at step i of the pentatope's lattice path it is the
// difference of the two consecutive corner samples; the pentatope is the one whose path adds the axes in the order of
// decreasing fractional coordinate of the tetrahedron's centroid.
struct cxp_grid4 {
    const float* A;
    int n[4];
    double value;
};
__device__ __forceinline__ void cxp_crossing4(const cxp_grid4& G, uint32_t key, double x[4]) {
    const uint32_t lin = key >> 4, d = key & 15u;
    const uint32_t s3 = (uint32_t)G.n[3], s2 = s3 * (uint32_t)G.n[2], s1 = s2 * (uint32_t)G.n[1];
    uint32_t q[4];
    q[0] = lin / s1;
    uint32_t r = lin - q[0] * s1;
    q[1] = r / s2;
    r -= q[1] * s2;
    q[2] = r / s3;
    q[3] = r - q[2] * s3;
    const uint32_t lin2 = lin + ((d & 8u) ? s1 : 0u) + ((d & 4u) ? s2 : 0u) + ((d & 2u) ? s3 : 0u) + (d & 1u);
    const double f0 = (double)G.A[lin], f1 = (double)G.A[lin2];
    const bool owner_low = !(f0 > f1);
    const double flow = owner_low ? f0 : f1, fhigh = owner_low ? f1 : f0;
    double ratio = 0.5;
    const double den = 1.0 * (fhigh - flow);
    if (!(fabs(den) <= 1e-8)) ratio = (G.value - flow) / den;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const double db = (double)((d >> (3 - a)) & 1u);
        const double low = owner_low ? (double)q[a] : (double)q[a] + db, high = owner_low ? (double)q[a] + db : (double)q[a];
        x[a] = low + ratio * (high - low);
    }
}
__global__ void cxp_k_tets_orient(int32_t* tets, uint32_t nt, const uint32_t* keys, cxp_grid4 G) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    double p[4][4], c[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        cxp_crossing4(G, keys[(uint32_t)tets[(size_t)t * 4 + k]], p[k]);
#pragma unroll
        for (int a = 0; a < 4; a++) c[a] += 0.25 * p[k][a];
    }
    int base[4], ord[4] = {0, 1, 2, 3};
    double frac[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        int i0 = (int)floor(c[a]);
        i0 = max(0, min(i0, G.n[a] - 2));
        base[a] = i0;
        frac[a] = c[a] - (double)i0;
    }
#pragma unroll
    for (int i = 1; i < 4; i++)
#pragma unroll
        for (int j = i; j > 0; j--)
            if (frac[ord[j]] > frac[ord[j - 1]]) { const int x = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = x; }
    int q[4] = {base[0], base[1], base[2], base[3]};
    const size_t s3 = (size_t)G.n[3], s2 = s3 * (size_t)G.n[2], s1 = s2 * (size_t)G.n[1];
    double prev = (double)G.A[q[0] * s1 + q[1] * s2 + q[2] * s3 + q[3]], g[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        q[ord[i]] += 1;
        const double cur = (double)G.A[q[0] * s1 + q[1] * s2 + q[2] * s3 + q[3]];
        g[ord[i]] = cur - prev;
        prev = cur;
    }
    // det of the rows (p1-p0, p2-p0, p3-p0, g), expanded along the last row
    double m[3][4];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int a = 0; a < 4; a++) m[k][a] = p[k + 1][a] - p[0][a];
    auto det3 = [&](int a, int b, int cc) {
        return m[0][a] * (m[1][b] * m[2][cc] - m[1][cc] * m[2][b]) - m[0][b] * (m[1][a] * m[2][cc] - m[1][cc] * m[2][a]) +
               m[0][cc] * (m[1][a] * m[2][b] - m[1][b] * m[2][a]);
    };
    const double det = -g[0] * det3(1, 2, 3) + g[1] * det3(0, 2, 3) - g[2] * det3(0, 1, 3) + g[3] * det3(0, 1, 2);
    if (det < 0.0) {
        const int32_t x = tets[(size_t)t * 4 + 2];
        tets[(size_t)t * 4 + 2] = tets[(size_t)t * 4 + 3];
        tets[(size_t)t * 4 + 3] = x;
    }
}
__global__ void cxp_k_morph_count(const int32_t* tets, uint32_t nt, const double* pts, const uint32_t* prio, const u64* mm, uint32_t* counts) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const double t_eps = 1e-7 * (cxp_from_orderable(mm[1]) - cxp_from_orderable(mm[0]));
    counts[t] = cxp_morph_slices(tets, t, pts, prio, t_eps, [](u64, u64, u64) {});
}
// (Measured and dropped: the workgroup's triangles put together in LDS and written out as consecutive words -- 0.81 -> 0.83 ms: the kernel is
// bound by its arithmetic and gathers at 128 registers, not by its stores.)
__global__ void cxp_k_morph_emit(const int32_t* tets, uint32_t nt, const double* pts, const uint32_t* prio, const u64* mm,
                                 const uint32_t* offsets, u64* pairs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const double t_eps = 1e-7 * (cxp_from_orderable(mm[1]) - cxp_from_orderable(mm[0]));
    u64* out = pairs + (size_t)offsets[t] * 3;
    cxp_morph_slices(tets, t, pts, prio, t_eps, [&](u64 p0, u64 p1, u64 p2) { out[0] = p0; out[1] = p1; out[2] = p2; out += 3; });
}
// segments = distinct vertex pairs
// (slots by the first vertex of the pair, as in the 3-D edge table: vertex ids follow the march, so the triangles of a wave
// probe neighbouring lines; segment ids = slot order then follow the vertices too, which carries on into the edge table)
// A workgroup takes 1024 consecutive keys (the three pairs of ~340 neighbouring triangles: every segment shows up several times
// among them) and puts them through a set in LDS first; only the first of each key in the block goes on to the device-scope table,
// whose reads and compare-and-swaps are executed at the memory side of the fabric and bound this kernel.
#define CXP_SEG_KEYS 1024u
#define CXP_SEG_SLOTS 2048u
__global__ __launch_bounds__(256) void cxp_k_seg_insert(const u64* pairs, size_t n, u64* tkeys, u64 mask, u64 mult) {
    __shared__ u64 lset[CXP_SEG_SLOTS];
    for (uint32_t x = threadIdx.x; x < CXP_SEG_SLOTS; x += 256u) lset[x] = CXP_EMPTY;
    __syncthreads();
    const size_t b0 = (size_t)blockIdx.x * CXP_SEG_KEYS;
    u64 key[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const size_t i = b0 + k * 256u + threadIdx.x;
        key[k] = (i < n) ? pairs[i] : CXP_EMPTY;
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (key[k] == CXP_EMPTY) continue;
        uint32_t ls = (uint32_t)cxp_mix(key[k]) & (CXP_SEG_SLOTS - 1u);
        bool first = false;
        for (uint32_t probes = 0; probes < CXP_SEG_SLOTS; probes++) {      // (ends earlier: 1024 keys, 2048 slots)
            const u64 cur = atomicCAS((unsigned long long*)&lset[ls], (unsigned long long)CXP_EMPTY, (unsigned long long)key[k]);
            if (cur == CXP_EMPTY) { first = true; break; }
            if (cur == key[k]) break;
            ls = (ls + 1u) & (CXP_SEG_SLOTS - 1u);
        }
        if (!first) continue;
        u64 slot = cxp_edge_slot((uint32_t)(key[k] >> 32), (uint32_t)key[k], mask, mult);
        for (;;) {
            u64 cur = __hip_atomic_load(&tkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (see cxp_k_edge_lists)
            if (cur == CXP_EMPTY) cur = atomicCAS(&tkeys[slot], CXP_EMPTY, key[k]);
            if (cur == CXP_EMPTY || cur == key[k]) break;
            slot = (slot + 1) & mask;
        }
    }
}
// Segment ids = rank of the occupied slot, by ordered compaction without a flag or id word per SLOT (the table has 134 M slots on config 4
// for 12 M segments: flags + a full scan + ids for every slot were 1.5 GB written and 2 GB read, ~1 ms): a workgroup counts the occupied
// slots of its 4 096 (cxp_k_occ_count), one workgroup scans the block counts (cxp_k_scan_sums), and cxp_k_seg_write4 redoes the scan inside
// its block, lists the occupied slots in LDS and writes one record per list entry -- ids[slot] only for occupied slots (what
// cxp_k_tri_segments looks up), segments / midpoints / time ranges in id order, next to each other.
#define CXP_OCC_BLOCK 4096u
__device__ __forceinline__ uint32_t cxp_occ16(const u64* tkeys, size_t base, size_t n) {
    uint32_t m = 0;
    if (base + 16u <= n) {
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(tkeys + base + 2u * k);
            m |= (w.x != CXP_EMPTY ? 1u : 0u) << (2u * k);
            m |= (w.y != CXP_EMPTY ? 2u : 0u) << (2u * k);
        }
    } else {
        for (uint32_t k = 0; k < 16u && base + k < n; k++) m |= (tkeys[base + k] != CXP_EMPTY ? 1u : 0u) << k;
    }
    return m;
}
__global__ __launch_bounds__(256) void cxp_k_occ_count(const u64* tkeys, size_t n, uint32_t* count) {
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t c = __popc(cxp_occ16(tkeys, (size_t)blockIdx.x * CXP_OCC_BLOCK + threadIdx.x * 16u, n));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0) count[blockIdx.x] = s_n;
}
__global__ __launch_bounds__(256) void cxp_k_seg_write4(const u64* tkeys, size_t n, const uint32_t* boff, const uint32_t* total, uint32_t nblocks,
                                                        const double* pts, uint32_t* ids, int32_t* segs, double* mid, double* stime) {
    __shared__ uint32_t s[256];
    __shared__ uint16_t list[CXP_OCC_BLOCK];
    const uint32_t first = boff[blockIdx.x];
    const uint32_t next = (blockIdx.x + 1u < nblocks) ? boff[blockIdx.x + 1u] : *total;
    if (next == first) return;
    const size_t base = (size_t)blockIdx.x * CXP_OCC_BLOCK;
    uint32_t m = cxp_occ16(tkeys, base + threadIdx.x * 16u, n);
    // exclusive prefix of the threads' counts (Hillis-Steele over 256), then the positions of the set bits
    const uint32_t cnt = __popc(m);
    s[threadIdx.x] = cnt;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint32_t x = (threadIdx.x >= o) ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += x;
        __syncthreads();
    }
    uint32_t pos = s[threadIdx.x] - cnt;
    const uint32_t nl = s[255];
    while (m) {
        const uint32_t k = __ffs(m) - 1u;
        m &= m - 1u;
        list[pos++] = (uint16_t)(threadIdx.x * 16u + k);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nl; j += 256u) {
        const size_t slot = base + list[j];
        const u64 key = tkeys[slot];
        const uint32_t a = (uint32_t)(key >> 32), b = (uint32_t)key;
        const double4 pa = *reinterpret_cast<const double4*>(pts + (size_t)a * 4);
        const double4 pb = *reinterpret_cast<const double4*>(pts + (size_t)b * 4);
        const uint32_t sg = first + j;
        ids[slot] = sg;
        const bool swap = pa.w > pb.w;
        segs[(size_t)sg * 2] = (int32_t)(swap ? b : a); segs[(size_t)sg * 2 + 1] = (int32_t)(swap ? a : b);
        mid[(size_t)sg * 3] = swap ? 0.5 * (pb.x + pa.x) : 0.5 * (pa.x + pb.x);
        mid[(size_t)sg * 3 + 1] = swap ? 0.5 * (pb.y + pa.y) : 0.5 * (pa.y + pb.y);
        mid[(size_t)sg * 3 + 2] = swap ? 0.5 * (pb.z + pa.z) : 0.5 * (pa.z + pb.z);
        stime[(size_t)sg * 2] = swap ? pb.w : pa.w; stime[(size_t)sg * 2 + 1] = swap ? pa.w : pb.w;
    }
}
// triangles as segment-id triples + their time range (morph_geometry.py:69-89)
__global__ void cxp_k_tri_segments(const u64* pairs, uint32_t nt, const u64* tkeys, const uint32_t* ids, u64 mask, u64 mult, const double* stime,
                                   const u64* mm, int32_t* tris, double* ttime) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    double lo = cxp_from_orderable(mm[0]), hi = cxp_from_orderable(mm[1]);
    // the three keys, their three probes side by side, the three ids, the six times -- and only then the stores (one edge after the
    // other with its store at the end of each round, every load of the next round waited for that store: they may alias for all the
    // compiler knows)
    const u64 key[3] = {pairs[(size_t)t * 3], pairs[(size_t)t * 3 + 1], pairs[(size_t)t * 3 + 2]};
    u64 slot[3];
#pragma unroll
    for (int e = 0; e < 3; e++) slot[e] = cxp_edge_slot((uint32_t)(key[e] >> 32), (uint32_t)key[e], mask, mult);
    for (;;) {
        const u64 k0 = tkeys[slot[0]], k1 = tkeys[slot[1]], k2 = tkeys[slot[2]];
        const bool m0 = k0 == key[0], m1 = k1 == key[1], m2 = k2 == key[2];
        if (m0 && m1 && m2) break;
        if (!m0) slot[0] = (slot[0] + 1) & mask;
        if (!m1) slot[1] = (slot[1] + 1) & mask;
        if (!m2) slot[2] = (slot[2] + 1) & mask;
    }
    const uint32_t s0 = ids[slot[0]], s1 = ids[slot[1]], s2 = ids[slot[2]];
    const double a0 = stime[(size_t)s0 * 2], b0 = stime[(size_t)s0 * 2 + 1], a1 = stime[(size_t)s1 * 2], b1 = stime[(size_t)s1 * 2 + 1];
    const double a2 = stime[(size_t)s2 * 2], b2 = stime[(size_t)s2 * 2 + 1];
    lo = fmax(fmax(lo, a0), fmax(a1, a2));
    hi = fmin(fmin(hi, b0), fmin(b1, b2));
    tris[(size_t)t * 3] = (int32_t)s0; tris[(size_t)t * 3 + 1] = (int32_t)s1; tris[(size_t)t * 3 + 2] = (int32_t)s2;
    ttime[(size_t)t * 2] = lo; ttime[(size_t)t * 2 + 1] = hi;
}
// edge table with a linked list of the triangles on each edge (an "edge" = pair of segment ids).  A workgroup takes CXP_EL consecutive
// triangles: their visits of one edge are chained in LDS first (a table of keys with a head per slot, one LDS exchange per visit), and
// only the block's FIRST visitor of an edge -- the tail of that chain -- goes to the device-scope table: it finds or claims the key,
// swaps the block's local head in as the edge's new head and hangs the old head behind itself.  One device-scope exchange per (edge,
// block) instead of one per visit (triangle ids follow the march: most of an edge's triangles sit in one block); the reads and
// read-modify-writes at the memory side of the fabric are what this stage is bound by.  The order inside a list is of no consequence
// (cxp_k_edge_union_compat looks at every pair).  Measured and dropped (third session of round 4): key and head of a slot next to each other
// in ONE table (one line per probe, as the 3-D edge table has it) -- 2.99 -> 3.22 ms on config 4.
#define CXP_EL 512u
#define CXP_EL_PER (CXP_EL / 256u)
#define CXP_EL_SLOTS (4u * CXP_EL)
__global__ __launch_bounds__(256) void cxp_k_edge_lists(const int32_t* tri, uint32_t nt, u64* ekeys, u64* eheads, u64 mask, u64 mult, uint32_t* next) {
    __shared__ u64 lkey[CXP_EL_SLOTS];
    __shared__ uint32_t lhead[CXP_EL_SLOTS];
    for (uint32_t x = threadIdx.x; x < CXP_EL_SLOTS; x += 256u) { lkey[x] = CXP_EMPTY; lhead[x] = CXP_NONE; }
    __syncthreads();
    const uint32_t b0 = blockIdx.x * CXP_EL;
    u64 key_[CXP_EL_PER][3];
    uint16_t slot_[CXP_EL_PER][3];
    uint32_t tails = 0;       // bit 3 i + e: this visit is the block's first of its edge
#pragma unroll
    for (uint32_t i = 0; i < CXP_EL_PER; i++) {
        const uint32_t t = b0 + i * 256u + threadIdx.x;
        if (t >= nt) continue;
        const uint32_t v[3] = {(uint32_t)tri[(size_t)t * 3], (uint32_t)tri[(size_t)t * 3 + 1], (uint32_t)tri[(size_t)t * 3 + 2]};
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const uint32_t p = v[e], q = v[(e + 1) % 3];
            const u64 key = ((u64)min(p, q) << 32) | (u64)max(p, q);
            key_[i][e] = key;
            uint32_t ls = (uint32_t)cxp_mix(key) & (CXP_EL_SLOTS - 1u);
            for (uint32_t probes = 0; probes < CXP_EL_SLOTS; probes++) {      // (ends earlier: 3 CXP_EL visits, 4 CXP_EL slots)
                const u64 cur = atomicCAS((unsigned long long*)&lkey[ls], (unsigned long long)CXP_EMPTY, (unsigned long long)key);
                if (cur == CXP_EMPTY || cur == key) break;
                ls = (ls + 1u) & (CXP_EL_SLOTS - 1u);
            }
            slot_[i][e] = (uint16_t)ls;
            const uint32_t me = t * 3u + (uint32_t)e;
            const uint32_t prev = atomicExch(&lhead[ls], me);
            if (prev == CXP_NONE) tails |= 1u << (3u * i + (uint32_t)e);
            else next[me] = prev;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < CXP_EL_PER; i++) {
        const uint32_t t = b0 + i * 256u + threadIdx.x;
        if (t >= nt) continue;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            if (!((tails >> (3u * i + (uint32_t)e)) & 1u)) continue;
            const u64 key = key_[i][e];
            u64 slot = cxp_edge_slot((uint32_t)(key >> 32), (uint32_t)key, mask, mult);
            for (;;) {
                // device-scope read first: every visitor of an edge but the first finds the key there and needs no read-modify-write
                u64 cur = __hip_atomic_load(&ekeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == CXP_EMPTY) cur = atomicCAS(&ekeys[slot], CXP_EMPTY, key);
                if (cur == CXP_EMPTY || cur == key) break;
                slot = (slot + 1) & mask;
            }
            const u64 old = atomicExch(&eheads[slot], (u64)lhead[slot_[i][e]]);      // the block's chain in front of what was there
            next[t * 3u + (uint32_t)e] = (old == CXP_EMPTY) ? CXP_NONE : (uint32_t)old;
        }
    }
}
// link every pair of time-compatible triangles on a common edge (morph_geometry.py:61-67, surface_geometry.py:117-128).
// Every visit walks the REST of its edge's list -- from the node behind its own to the end -- so each pair of nodes is looked at
// once, by the one nearer the head; no visit has to find the edge's slot in the table again.  (Until round 3 every visit probed the
// table for its edge and walked the whole list from the head, keeping the triangles with a smaller id: a random table read per
// visit and twice the hops.)
__global__ void cxp_k_edge_union_compat(uint32_t nt, const uint32_t* next, const double* ttime, u64* parent) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const double lo = ttime[(size_t)t * 2], hi = ttime[(size_t)t * 2 + 1];
    uint32_t it[3] = {next[t * 3u], next[t * 3u + 1u], next[t * 3u + 2u]};      // the three walks side by side
    while (it[0] != CXP_NONE || it[1] != CXP_NONE || it[2] != CXP_NONE) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
            if (it[e] == CXP_NONE) continue;
            const uint32_t o = it[e] / 3u;
            const double l2 = fmax(lo, ttime[(size_t)o * 2]), h2 = fmin(hi, ttime[(size_t)o * 2 + 1]);
            it[e] = next[it[e]];
            if (l2 < h2 && o != t) cxp_union0(parent, t, o);   // connectivity only: the slices are wound alike (cxp_morph_slices); path halving (cxp_find0)
        }
    }
}


static int cxp_morph_sort_by_start(cx_ctx* ctx, cx_post_state* S, uint32_t nseg, uint32_t ntri, const u64* mm_host);
extern "C" int cx_morph_triangles(cx_ctx* ctx, int64_t* out_counts) {
    if (!ctx) return CX_ERR_INVALID;
    cx_state4* G = ctx->s4;
    if (!G || !G->post_valid || !ctx->post) { ctx->err = "cx_morph_triangles: run cx_postprocess4d first"; return CX_ERR_STATE; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S = ctx->post;
    hipStream_t st = ctx->stream;
    const uint32_t nv = (uint32_t)S->nv_out, nt = (uint32_t)S->nt_out;   // vertices / surviving tetrahedra
    const double* pts = (const double*)S->pts.p;
    const uint32_t* prio = (const uint32_t*)S->prio.p;
    const int32_t* tets = (const int32_t*)S->tri_out.p;
    uint32_t* misc = (uint32_t*)S->misc.p;
    u64* mm = (u64*)(misc + 16);
    int rc;
    int64_t counts[8] = {nv, 0, 0, 0, 0, 0, 0, 0};
    u64 h_mm[2] = {0, 0};      // min / max t of all points (MorphTriangles.min_value / max_value), orderable encodings
    S->ms_out = 0; S->mt_out = 0; S->msorted = false;
    if (nv && nt) {
        const u64 init[2] = {~0ULL, 0ULL};
        CXP_HIP(ctx, hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(cxp_k_minmax_t, dim3(std::min(cxp_blocks(nv), 2048u)), dim3(256), 0, st, pts, nv, mm);
        if ((rc = cxp_reserve(ctx, S->flags, (size_t)(nt + 16) * sizeof(uint32_t)))) return rc;
        if ((rc = cxp_reserve(ctx, S->scan, (size_t)(nt + 16) * sizeof(uint32_t)))) return rc;
        uint32_t* cnt = (uint32_t*)S->flags.p;
        uint32_t* off = (uint32_t*)S->scan.p;
        cxp_grid4 G4;
        G4.A = G->grid;
        for (int k = 0; k < 4; k++) G4.n[k] = (int)G->n[k];
        G4.value = G->value;
        // the march's table already emits every tetrahedron in that order (tools/gen_tables.py orient_tet: exact, and valid
        // where samples EQUAL the isovalue and the determinant below vanishes); CX_TETS_ORIENT=1 (debug) recomputes it from the data
        if (cx_debug_knob("CX_TETS_ORIENT", 0))
            hipLaunchKernelGGL(cxp_k_tets_orient, dim3(cxp_blocks(nt)), dim3(256), 0, st, (int32_t*)S->tri_out.p, nt, prio, G4);
        hipLaunchKernelGGL(cxp_k_morph_count, dim3(cxp_blocks(nt)), dim3(256), 0, st, tets, nt, pts, prio, mm, cnt);
        if ((rc = cxp_scan(ctx, S, cnt, off, nt, misc + 1))) return rc;
        uint32_t ntri = 0;
        u64* h = h_mm;
        CXP_HIP(ctx, hipMemcpyAsync(&ntri, misc + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        CXP_HIP(ctx, hipMemcpyAsync(h, mm, sizeof(h_mm), hipMemcpyDeviceToHost, st));
        CXP_HIP(ctx, hipStreamSynchronize(st));
        S->me_off.clear();
        if (ntri) {
            if ((rc = cxp_reserve(ctx, S->mpairs, (size_t)ntri * 3 * sizeof(u64)))) return rc;
            u64* pairs = (u64*)S->mpairs.p;
            hipLaunchKernelGGL(cxp_k_morph_emit, dim3(cxp_blocks(nt)), dim3(256), 0, st, tets, nt, pts, prio, mm, off, pairs);
            // ---- segments
            const size_t np = (size_t)ntri * 3;
            // (n keys of which a third or less are distinct -- every segment shows up in several triangles: the edge-table rule, 5/4 of the
            // bound, keeps the load below 0.8 in the worst case and near 0.25 here with half the slots to clear, flag and scan)
            const u64 ssz = cxp_edge_table_size(np);
            if ((rc = cxp_reserve(ctx, S->tkeys, ssz * sizeof(u64)))) return rc;
            const uint32_t nob = cxp_blocks(ssz, CXP_OCC_BLOCK);
            if ((rc = cxp_reserve(ctx, S->flags, (size_t)(nob + 16) * sizeof(uint32_t)))) return rc;
            if ((rc = cxp_reserve(ctx, S->scan, (size_t)(ssz + 16) * sizeof(uint32_t)))) return rc;
            u64* skeys = (u64*)S->tkeys.p;
            uint32_t* boff = (uint32_t*)S->flags.p;        // occupied slots per block of 4 096, then their exclusive scan
            uint32_t* sid = (uint32_t*)S->scan.p;          // segment id per slot, written for occupied slots only
            hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, skeys, (size_t)ssz, CXP_EMPTY);
            const u64 smult = std::max<u64>(1, ssz / std::max<u64>(1, (u64)nv));
            hipLaunchKernelGGL(cxp_k_seg_insert, dim3(cxp_blocks(np, CXP_SEG_KEYS)), dim3(256), 0, st, pairs, np, skeys, ssz - 1, smult);
            hipLaunchKernelGGL(cxp_k_occ_count, dim3(nob), dim3(256), 0, st, (const u64*)skeys, (size_t)ssz, boff);
            hipLaunchKernelGGL(cxp_k_scan_sums, dim3(1), dim3(1024), 0, st, boff, nob, misc + 2, (unsigned long long*)nullptr);
            uint32_t nseg = 0;
            CXP_HIP(ctx, hipMemcpyAsync(&nseg, misc + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            CXP_HIP(ctx, hipStreamSynchronize(st));
            if ((rc = cxp_reserve(ctx, S->msegs, (size_t)(nseg + 1) * 2 * sizeof(int32_t)))) return rc;
            if ((rc = cxp_reserve(ctx, S->mmid, (size_t)(nseg + 1) * 3 * sizeof(double)))) return rc;
            if ((rc = cxp_reserve(ctx, S->mtime, ((size_t)(nseg + 1) * 2 + (size_t)(ntri + 1) * 2) * sizeof(double)))) return rc;
            if ((rc = cxp_reserve(ctx, S->mtris, (size_t)(ntri + 1) * 3 * sizeof(int32_t)))) return rc;
            int32_t* segs = (int32_t*)S->msegs.p;
            double* mid = (double*)S->mmid.p;
            double* stime = (double*)S->mtime.p;
            double* ttime = stime + (size_t)(nseg + 1) * 2;
            int32_t* tris = (int32_t*)S->mtris.p;
            hipLaunchKernelGGL(cxp_k_seg_write4, dim3(nob), dim3(256), 0, st, (const u64*)skeys, (size_t)ssz, (const uint32_t*)boff, (const uint32_t*)(misc + 2), nob,
                               pts, sid, segs, mid, stime);
            hipLaunchKernelGGL(cxp_k_tri_segments, dim3(cxp_blocks(ntri)), dim3(256), 0, st, pairs, ntri, skeys, sid, ssz - 1, smult, stime, mm, tris, ttime);
            CXP_HIP(ctx, hipStreamSynchronize(st));   // the segment table is reused below
            // ---- orientation on the segment midpoints, time-compatible neighbours only
            const u64 esz = cxp_edge_table_size((size_t)ntri * 3);
            const u64 emult4 = std::max<u64>(1, esz / std::max<u64>(1, (u64)nseg));
            if ((rc = cxp_reserve(ctx, S->tkeys, esz * sizeof(u64)))) return rc;
            if ((rc = cxp_reserve(ctx, S->tvals, esz * sizeof(u64)))) return rc;
            if ((rc = cxp_reserve(ctx, S->parent, (size_t)ntri * sizeof(u64)))) return rc;
            if ((rc = cxp_reserve(ctx, S->mnext, (size_t)ntri * 3 * sizeof(uint32_t)))) return rc;
            if ((rc = cxp_reserve(ctx, S->comp, (size_t)ntri * (3 * sizeof(u64) + sizeof(uint32_t))))) return rc;
            u64* ekeys = (u64*)S->tkeys.p;
            u64* eheads = (u64*)S->tvals.p;
            u64* parent = (u64*)S->parent.p;
            uint32_t* next = (uint32_t*)S->mnext.p;
            u64* cmaxx = (u64*)S->comp.p;
            u64* cbest = cmaxx + ntri;
            u64* cmaxv = cbest + ntri;
            uint32_t* cstart = (uint32_t*)(cmaxv + ntri);
            hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, ekeys, (size_t)esz, CXP_EMPTY);
            hipLaunchKernelGGL(cxp_k_fill64, dim3(2048), dim3(256), 0, st, eheads, (size_t)esz, CXP_EMPTY);
            hipLaunchKernelGGL(cxp_k_iota64, dim3(cxp_blocks(ntri)), dim3(256), 0, st, parent, ntri);
            hipLaunchKernelGGL(cxp_k_edge_lists, dim3(cxp_blocks(ntri, CXP_EL)), dim3(256), 0, st, tris, ntri, ekeys, eheads, esz - 1, emult4, next);
            hipLaunchKernelGGL(cxp_k_edge_union_compat, dim3(cxp_blocks(ntri)), dim3(256), 0, st, ntri, (const uint32_t*)next, (const double*)ttime, parent);
            if ((rc = cxp_flatten(ctx, parent, ntri, misc))) return rc;
            CXP_HIP(ctx, hipMemsetAsync(cmaxx, 0, (size_t)ntri * (3 * sizeof(u64) + sizeof(uint32_t)), st));
            CXP_HIP(ctx, hipMemsetAsync(misc + 3, 0, sizeof(uint32_t), st));
            const uint32_t* nokeys = nullptr;   // segment midpoints: the largest index breaks the tie
            const uint8_t* nocls = nullptr;
            hipLaunchKernelGGL(cxp_k_comp_maxx, dim3(cxp_blocks(ntri)), dim3(256), 0, st, tris, ntri, mid, parent, cmaxx, nocls);
            // the four kernels that pick a component's start triangle visit the list of triangles AT the component's largest x
            // (cxp_k_comp_list; the list sits where the edge lists' `next` words were: free by now)
            uint32_t* clist = (uint32_t*)S->mnext.p;
            const uint32_t* cn = misc + 11;
            const dim3 lgrid(std::min(cxp_blocks(ntri), 1024u));
            CXP_HIP(ctx, hipMemsetAsync(misc + 11, 0, sizeof(uint32_t), st));
            hipLaunchKernelGGL(cxp_k_comp_list, dim3(cxp_blocks(ntri)), dim3(256), 0, st, tris, ntri, mid, parent, cmaxx, nocls, clist, misc + 11);
            hipLaunchKernelGGL(cxp_k_comp_maxv, lgrid, dim3(256), 0, st, tris, ntri, mid, parent, cmaxx, cmaxv, nokeys, nocls, (const uint32_t*)clist, cn);
            hipLaunchKernelGGL(cxp_k_comp_start, lgrid, dim3(256), 0, st, tris, ntri, mid, parent, cmaxv, cbest, nocls, (const uint32_t*)clist, cn);
            hipLaunchKernelGGL(cxp_k_comp_pick, lgrid, dim3(256), 0, st, tris, ntri, mid, parent, cmaxv, cbest, cstart, nocls, (const uint32_t*)clist, cn);
            hipLaunchKernelGGL(cxp_k_comp_decide, lgrid, dim3(256), 0, st, tris, ntri, mid, parent, cstart, cbest, (const uint32_t*)clist, cn);
            hipLaunchKernelGGL(cxp_k_orient, dim3(cxp_blocks(ntri)), dim3(256), 0, st, tris, ntri, parent, cbest, misc + 3);
            uint32_t ncomp = 0;
            CXP_HIP(ctx, hipMemcpyAsync(&ncomp, misc + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            CXP_HIP(ctx, hipStreamSynchronize(st));
            CXP_HIP(ctx, hipGetLastError());
            S->ms_out = nseg; S->mt_out = ntri;
            counts[1] = nseg; counts[2] = ntri; counts[4] = ncomp;
        }
        memcpy(&counts[5], &h[0], 8); memcpy(&counts[6], &h[1], 8);   // orderable encodings, decoded by the host side
    }
    // Segments and triangles sorted by the bin of their start time: the surface at a time t is then a window of ids (cx_morph_eval_many).
    // AFTER the orientation: on sorted triangles the edge lists get a little shorter work (2.9 -> 2.7 ms) but the time-compatible unions
    // take 9.4 ms instead of 2.6 (measured on config 4: in march order the triangles of one edge follow each other within a few ids and a
    // list's nodes share cache lines; sorted by time they lie a layer's worth of ids apart).
    if (S->ms_out && S->mt_out && (rc = cxp_morph_sort_by_start(ctx, S, (uint32_t)S->ms_out, (uint32_t)S->mt_out, h_mm))) {
        S->ms_out = 0; S->mt_out = 0;
        return rc;
    }
    if (out_counts) memcpy(out_counts, counts, sizeof(counts));
    return CX_OK;
}

extern "C" int cx_morph_download(cx_ctx* ctx, double* points_xyzt, int32_t* segments, int32_t* triangles) {
    if (!ctx || !ctx->post) return CX_ERR_INVALID;
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    cx_post_state* S = ctx->post;
    void* d[3] = {(points_xyzt && S->nv_out) ? (void*)points_xyzt : nullptr, (segments && S->ms_out) ? (void*)segments : nullptr,
                  (triangles && S->mt_out) ? (void*)triangles : nullptr};
    const void* sp[3] = {S->pts.p, S->msegs.p, S->mtris.p};
    const size_t nb[3] = {(size_t)S->nv_out * 4 * sizeof(double), (size_t)S->ms_out * 2 * sizeof(int32_t), (size_t)S->mt_out * 3 * sizeof(int32_t)};
    return cx_copy_to_host(ctx, 3, d, sp, nb);
}

// ---- stable sort of the morph triangles and their segments by the bin of their start time ----------------------------------------
// (the last step of cx_morph_triangles).  A morph triangle lives for a layer or two of the time axis, and ids follow the march, whose
// fastest axis is time: in id order the triangles that exist at one t are spread over ALL ids (nearly every block of 1 024 consecutive
// triangles exists at every t), and rounds 1-4 walked flag arrays as long as all triangles and all segments for every surface.  Sorted
// by start-time bin -- CXP_SB_BINS bins over the time range of all points, ids keeping their order inside a bin (stable: the
// neighbourhood the march gives the ids survives inside a time layer) -- the triangles that exist at t start inside the window of
// bins [bin(t - longest life) - 1, bin(t) + 1], a contiguous range of ids, and so do the segments they use.  The arrays the caller
// downloads (cx_morph_download) are the sorted ones: the order of morph triangles carries no meaning in the reference (a Python list
// filled from a dict, pentatopes.py:314-368), and the host-side MorphTriangles.triangles_at of the mirrored class sees the same order.
//
// Counting sort with no atomics: a wave takes CXP_SB_UNIT consecutive elements, counts them per bin in LDS (lanes of one bin add up
// among themselves: neighbouring triangles start at the same few times) and leaves the non-zero counts in a [bin][unit] matrix; an
// exclusive scan of that matrix in memory order is the position of every (bin, unit)'s first element; a second walk hands out ranks.
#define CXP_SB_BINS 256u
#define CXP_SB_UNIT 1024u
__device__ __forceinline__ uint32_t cxp_sb_bin(double x, double lo, double inv_width) {
    const double b = (x - lo) * inv_width;
    return b <= 0.0 ? 0u : (b >= (double)(CXP_SB_BINS - 1u) ? CXP_SB_BINS - 1u : (uint32_t)b);
}
// range = {start, end} per element (16 bytes, in order)
__global__ __launch_bounds__(256) void cxp_k_sb_hist(const double* range, uint32_t n, double lo, double inv_width, uint8_t* bins, uint32_t* counts,
                                                     uint32_t nunits, u64* maxdur) {
    __shared__ uint32_t h[4][CXP_SB_BINS];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (uint32_t x = lane; x < CXP_SB_BINS; x += 64u) h[w][x] = 0;
    __syncthreads();
    const uint32_t unit = blockIdx.x * 4u + w;
    double dur = 0.0;
    for (uint32_t r = 0; r < CXP_SB_UNIT / 64u; r++) {
        const size_t q = (size_t)unit * CXP_SB_UNIT + r * 64u + lane;
        const bool in = q < n;
        uint32_t bin = 0;
        if (in) {
            const double2 ab = *reinterpret_cast<const double2*>(range + q * 2);
            bin = cxp_sb_bin(ab.x, lo, inv_width);
            dur = fmax(dur, ab.y - ab.x);
            bins[q] = (uint8_t)bin;
        }
        uint64_t todo = __ballot(in);
        while (todo) {      // wave-uniform: one round per bin present among the 64
            const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t bb = (uint32_t)__shfl((int)bin, (int)leader);
            const uint64_t same = __ballot(in && bin == bb) & todo;
            if (lane == leader) h[w][bb] += (uint32_t)__popcll(same);
            todo &= ~same;
        }
    }
    __syncthreads();
    if (unit < nunits)
        for (uint32_t x = lane; x < CXP_SB_BINS; x += 64u)
            if (h[w][x]) counts[(size_t)x * nunits + unit] = h[w][x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dur = fmax(dur, __shfl_xor(dur, o));
    if (lane == 0 && dur > 0.0) cxp_max64(maxdur, cxp_orderable(dur));
}
__global__ __launch_bounds__(256) void cxp_k_sb_rank(const uint8_t* bins, uint32_t n, const uint32_t* offs, uint32_t nunits, uint32_t* rank) {
    __shared__ uint32_t h[4][CXP_SB_BINS];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (uint32_t x = lane; x < CXP_SB_BINS; x += 64u) h[w][x] = 0;
    __syncthreads();
    const uint32_t unit = blockIdx.x * 4u + w;
    uint32_t mine[CXP_SB_UNIT / 64u];
#pragma unroll
    for (uint32_t r = 0; r < CXP_SB_UNIT / 64u; r++) {
        const size_t q = (size_t)unit * CXP_SB_UNIT + r * 64u + lane;
        mine[r] = q < n ? (uint32_t)bins[q] : 0xFFFFFFFFu;
    }
    // which bins the unit holds (their cursors are the only ones loaded: a unit of neighbouring ids holds a few)
#pragma unroll
    for (uint32_t r = 0; r < CXP_SB_UNIT / 64u; r++) {
        const bool in = mine[r] != 0xFFFFFFFFu;
        uint64_t todo = __ballot(in);
        while (todo) {
            const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t bb = (uint32_t)__shfl((int)mine[r], (int)leader);
            const uint64_t same = __ballot(in && mine[r] == bb) & todo;
            if (lane == leader) h[w][bb] = 1u;
            todo &= ~same;
        }
    }
    __syncthreads();
    if (unit < nunits)
        for (uint32_t x = lane; x < CXP_SB_BINS; x += 64u)
            if (h[w][x]) h[w][x] = offs[(size_t)x * nunits + unit];
    __syncthreads();
    volatile uint32_t* cur = h[w];
#pragma unroll
    for (uint32_t r = 0; r < CXP_SB_UNIT / 64u; r++) {
        const bool in = mine[r] != 0xFFFFFFFFu;
        uint64_t todo = __ballot(in);
        uint32_t my_rank = 0;
        while (todo) {
            const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t bb = (uint32_t)__shfl((int)mine[r], (int)leader);
            const uint64_t same = __ballot(in && mine[r] == bb) & todo;
            const uint32_t base = cur[bb];
            if (in && mine[r] == bb) my_rank = base + (uint32_t)__popcll(same & ((1ULL << lane) - 1ULL));
            __builtin_amdgcn_wave_barrier();
            if (lane == leader) cur[bb] = base + (uint32_t)__popcll(same);
            __builtin_amdgcn_wave_barrier();
            todo &= ~same;
        }
        if (in) rank[(size_t)unit * CXP_SB_UNIT + r * 64u + lane] = my_rank;
    }
}
// first element of every bin (the scanned matrix at unit 0) + the total
__global__ void cxp_k_sb_starts(const uint32_t* offs, uint32_t nunits, uint32_t n, uint32_t* out) {
    const uint32_t b = threadIdx.x;
    if (b < CXP_SB_BINS) out[b] = offs[(size_t)b * nunits];
    if (b == CXP_SB_BINS) out[b] = n;
}
__global__ void cxp_k_sb_move_segs(const int32_t* segs, const double* stime, const double* mid, uint32_t ns, const uint32_t* rank, int32_t* segs2,
                                   double* stime2, double* mid2) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    const int2 ab = *reinterpret_cast<const int2*>(segs + (size_t)s * 2);
    const double2 tt = *reinterpret_cast<const double2*>(stime + (size_t)s * 2);
    double m0 = 0.0, m1 = 0.0, m2 = 0.0;
    if (mid2) { m0 = mid[(size_t)s * 3]; m1 = mid[(size_t)s * 3 + 1]; m2 = mid[(size_t)s * 3 + 2]; }    // (the midpoints only while somebody still reads them)
    const uint32_t r = rank[s];
    *reinterpret_cast<int2*>(segs2 + (size_t)r * 2) = ab;
    *reinterpret_cast<double2*>(stime2 + (size_t)r * 2) = tt;
    if (mid2) { mid2[(size_t)r * 3] = m0; mid2[(size_t)r * 3 + 1] = m1; mid2[(size_t)r * 3 + 2] = m2; }
}
__global__ void cxp_k_sb_move_tris(const int32_t* tris, const double* ttime, uint32_t nt, const uint32_t* rank_t, const uint32_t* rank_s, int32_t* tris2,
                                   double* ttime2) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nt) return;
    const uint32_t a = (uint32_t)tris[(size_t)q * 3], b = (uint32_t)tris[(size_t)q * 3 + 1], c = (uint32_t)tris[(size_t)q * 3 + 2];
    const double2 tt = *reinterpret_cast<const double2*>(ttime + (size_t)q * 2);
    const uint32_t r = rank_t[q];
    const uint32_t na = rank_s[a], nb = rank_s[b], nc = rank_s[c];
    tris2[(size_t)r * 3] = (int32_t)na; tris2[(size_t)r * 3 + 1] = (int32_t)nb; tris2[(size_t)r * 3 + 2] = (int32_t)nc;
    *reinterpret_cast<double2*>(ttime2 + (size_t)r * 2) = tt;
}
static inline double cxp_host_from_orderable(u64 o) {
    const u64 b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFULL) : ~o;
    double d;
    memcpy(&d, &b, sizeof(d));
    return d;
}
// ranks of n elements by the bin of range[2 q]; bin starts (CXP_SB_BINS + 1) and the longest range[2 q + 1] - range[2 q] to the host.
// Scratch: S->mnext (bins), S->flags (count matrix), S->scan (its scan), S->misc + 32.. (starts), S->misc + 20 (longest life).
static int cxp_sb_ranks(cx_ctx* ctx, cx_post_state* S, const double* range, uint32_t n, double lo, double inv_width, uint32_t* rank,
                        uint32_t* starts_host, double* maxdur_host) {
    int rc;
    hipStream_t st = ctx->stream;
    const uint32_t nunits = cxp_blocks(n, CXP_SB_UNIT);
    const size_t cells = (size_t)CXP_SB_BINS * nunits;
    if (cells >= 0xFFFFFFFFull) { ctx->err = "cx_morph_triangles: too many morph triangles for the start-time sort"; return CX_ERR_INVALID; }
    if ((rc = cxp_reserve(ctx, S->mnext, (size_t)n + 64))) return rc;
    if ((rc = cxp_reserve(ctx, S->flags, (cells + 16) * sizeof(uint32_t)))) return rc;
    if ((rc = cxp_reserve(ctx, S->scan, (cells + 16) * sizeof(uint32_t)))) return rc;
    uint8_t* bins = (uint8_t*)S->mnext.p;
    uint32_t* counts = (uint32_t*)S->flags.p;
    uint32_t* offs = (uint32_t*)S->scan.p;
    uint32_t* misc = (uint32_t*)S->misc.p;
    u64* maxdur = (u64*)(misc + 20);
    CXP_HIP(ctx, hipMemsetAsync(counts, 0, cells * sizeof(uint32_t), st));
    CXP_HIP(ctx, hipMemsetAsync(maxdur, 0, sizeof(u64), st));
    const uint32_t nblocks = cxp_blocks(nunits, 4);
    hipLaunchKernelGGL(cxp_k_sb_hist, dim3(nblocks), dim3(256), 0, st, range, n, lo, inv_width, bins, counts, nunits, maxdur);
    if ((rc = cxp_scan(ctx, S, counts, offs, (uint32_t)cells, misc + 22))) return rc;
    hipLaunchKernelGGL(cxp_k_sb_rank, dim3(nblocks), dim3(256), 0, st, (const uint8_t*)bins, n, (const uint32_t*)offs, nunits, rank);
    hipLaunchKernelGGL(cxp_k_sb_starts, dim3(1), dim3(512), 0, st, (const uint32_t*)offs, nunits, n, misc + 32);
    u64 md = 0;
    CXP_HIP(ctx, hipMemcpyAsync(starts_host, misc + 32, (CXP_SB_BINS + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    CXP_HIP(ctx, hipMemcpyAsync(&md, maxdur, sizeof(md), hipMemcpyDeviceToHost, st));
    CXP_HIP(ctx, hipStreamSynchronize(st));
    CXP_HIP(ctx, hipGetLastError());
    *maxdur_host = md ? cxp_host_from_orderable(md) : 0.0;
    return CX_OK;
}
// mm_host = orderable min / max of the time of all points (S->misc + 16, read by the caller)
static int cxp_morph_sort_by_start(cx_ctx* ctx, cx_post_state* S, uint32_t nseg, uint32_t ntri, const u64* mm_host) {
    int rc;
    hipStream_t st = ctx->stream;
    S->msorted = false;
    if (!nseg || !ntri) return CX_OK;
    const double lo = cxp_host_from_orderable(mm_host[0]), hi = cxp_host_from_orderable(mm_host[1]);
    const double inv_width = (hi > lo) ? (double)CXP_SB_BINS / (hi - lo) : 0.0;
    if ((rc = cxp_reserve(ctx, S->msegs2, (size_t)(nseg + 1) * 2 * sizeof(int32_t)))) return rc;
    if ((rc = cxp_reserve(ctx, S->mtris2, (size_t)(ntri + 1) * 3 * sizeof(int32_t)))) return rc;
    if ((rc = cxp_reserve(ctx, S->mtime2, ((size_t)(nseg + 1) * 2 + (size_t)(ntri + 1) * 2) * sizeof(double)))) return rc;
    if ((rc = cxp_reserve(ctx, S->parent, ((size_t)nseg + ntri + 64) * sizeof(uint32_t)))) return rc;
    uint32_t* rank_s = (uint32_t*)S->parent.p;
    uint32_t* rank_t = rank_s + nseg + 16;
    const int32_t* segs = (const int32_t*)S->msegs.p;
    const int32_t* tris = (const int32_t*)S->mtris.p;
    const double* stime = (const double*)S->mtime.p;
    const double* ttime = stime + (size_t)(nseg + 1) * 2;
    double* stime2 = (double*)S->mtime2.p;
    double* ttime2 = stime2 + (size_t)(nseg + 1) * 2;
    if ((rc = cxp_sb_ranks(ctx, S, stime, nseg, lo, inv_width, rank_s, S->mbin_s, &S->ms_maxdur))) return rc;
    if ((rc = cxp_sb_ranks(ctx, S, ttime, ntri, lo, inv_width, rank_t, S->mbin_t, &S->mt_maxdur))) return rc;
    // (the segment midpoints are scratch of the orientation, which is over: they stay behind)
    hipLaunchKernelGGL(cxp_k_sb_move_segs, dim3(cxp_blocks(nseg)), dim3(256), 0, st, segs, stime, (const double*)nullptr, nseg, (const uint32_t*)rank_s,
                       (int32_t*)S->msegs2.p, stime2, (double*)nullptr);
    hipLaunchKernelGGL(cxp_k_sb_move_tris, dim3(cxp_blocks(ntri)), dim3(256), 0, st, tris, ttime, ntri, (const uint32_t*)rank_t, (const uint32_t*)rank_s,
                       (int32_t*)S->mtris2.p, ttime2);
    CXP_HIP(ctx, hipStreamSynchronize(st));
    CXP_HIP(ctx, hipGetLastError());
    std::swap(S->msegs, S->msegs2);
    std::swap(S->mtris, S->mtris2);
    std::swap(S->mtime, S->mtime2);
    S->mt_lo = lo;
    S->mt_inv_width = inv_width;
    S->msorted = true;
    return CX_OK;
}

// ---- B6: the surfaces at times t[0..n) from the morph triangles (misc/morph_triangles.js:26-140; MorphTriangles.triangles_at):
// a triangle is visible while t lies inside the intervals of all three of its segments (lo <= t <= hi with the intersection of their
// ranges, which cxp_k_tri_segments left per triangle); its corners are the points of its segments at t (linear interpolation, low t ->
// high t).  Points and triangles of a surface are compacted in index order (the order numpy.unique / boolean indexing give on the host).
//
// ALL the times of a call go through ONE set of launches (config 4's per-t isosurface stream: 64 surfaces).  Time i tests the window of
// triangle ids [tf, tf + tn) that can exist at t_i and flags inside the window of segment ids [sf, sf + sn) (see the sort above); the
// windows of all times lie one behind the other in the flag arrays, each padded to whole blocks of CXP_ME_BLOCK flags, and a block finds
// its time by bisection over the descriptors.  Ordered compaction without materialised scans: flags are bytes, a workgroup counts the
// flags of its block (first pass), one workgroup per (time, kind) turns the block counts into offsets, and the consumer kernels redo the
// scan INSIDE their block while they write.  The surfaces lie one behind the other in the two output arrays, tightly.
#define CXP_ME_BLOCK 4096u
struct cxp_me_desc {
    double t;
    uint32_t tf, tn, sf, sn;   // windows of triangle / segment ids
    uint32_t tb0, sb0;         // first block of the time's triangle flags / segment flags (blocks of CXP_ME_BLOCK, all times one behind the other)
};
__device__ __forceinline__ uint32_t cxp_me_time_of_tblock(const cxp_me_desc* D, uint32_t nd, uint32_t blk) {
    uint32_t lo = 0, hi = nd;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (D[mid].tb0 <= blk) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t cxp_me_time_of_sblock(const cxp_me_desc* D, uint32_t nd, uint32_t blk) {
    uint32_t lo = 0, hi = nd;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (D[mid].sb0 <= blk) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t cxp_block_excl_t(uint32_t t, uint32_t* s, uint32_t& block_total) {   // exclusive prefix of one number per thread, 256 threads
    s[threadIdx.x] = t;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint32_t x = (threadIdx.x >= o) ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += x;
        __syncthreads();
    }
    block_total = s[255];
    const uint32_t r = s[threadIdx.x] - t;
    __syncthreads();
    return r;
}
// 16 flags (bytes, 0 / 1) as a bit mask; p is 16-byte aligned and the 16 bytes exist (the windows are padded to whole blocks)
__device__ __forceinline__ uint32_t cxp_flags16(const uint8_t* p) {
    const uint4 w = *reinterpret_cast<const uint4*>(p);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    uint32_t m = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
        m |= (((ww[k] & 1u) | ((ww[k] >> 7) & 2u) | ((ww[k] >> 14) & 4u) | ((ww[k] >> 21) & 8u))) << (4u * k);
    return m;
}
// the triangles of every window that exist at the window's time: a flag byte per window position (written for ALL positions of the
// padded window: nothing to clear) and the flags of their segments (set only; cleared by the kernel that consumes them).
// 16 workgroups of 256 per block of CXP_ME_BLOCK window positions.  err: a segment outside its window (cannot happen while the sort and
// the windows agree; checked because the write would land in another time's flags)
__global__ __launch_bounds__(256) void cxp_k_me_visible(const cxp_me_desc* D, uint32_t nd, const double* ttime, const int32_t* tris, uint8_t* tflag,
                                                        uint8_t* sused, uint32_t* err) {
    const uint32_t blk = blockIdx.x >> 4;
    const uint32_t i = cxp_me_time_of_tblock(D, nd, blk);
    const cxp_me_desc d = D[i];
    const uint32_t in_blk = (blockIdx.x & 15u) * 256u + threadIdx.x;
    const uint32_t x = (blk - d.tb0) * CXP_ME_BLOCK + in_blk;
    bool vis = false;
    if (x < d.tn) {
        const size_t q = (size_t)d.tf + x;
        const double2 r = *reinterpret_cast<const double2*>(ttime + q * 2);
        vis = r.x <= d.t && d.t <= r.y;
        if (vis) {
            const uint32_t a = (uint32_t)tris[q * 3] - d.sf, b = (uint32_t)tris[q * 3 + 1] - d.sf, c = (uint32_t)tris[q * 3 + 2] - d.sf;
            if (a < d.sn && b < d.sn && c < d.sn) {
                uint8_t* su = sused + (size_t)d.sb0 * CXP_ME_BLOCK;
                su[a] = 1; su[b] = 1; su[c] = 1;
            } else {
                vis = false;
                *err = 1u;
            }
        }
    }
    tflag[(size_t)blk * CXP_ME_BLOCK + in_blk] = vis ? 1 : 0;
}
// block counts: workgroups [0, nsb) the segment flags, the rest the triangle flags
__global__ __launch_bounds__(256) void cxp_k_me_count16(const uint8_t* sused, uint32_t nsb, const uint8_t* tflag, uint32_t* scnt, uint32_t* tcnt) {
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const bool second = blockIdx.x >= nsb;
    const uint32_t blk = second ? blockIdx.x - nsb : blockIdx.x;
    const uint8_t* flags = second ? tflag : sused;
    uint32_t c = __popc(cxp_flags16(flags + (size_t)blk * CXP_ME_BLOCK + threadIdx.x * 16u));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0) (second ? tcnt : scnt)[blk] = s_n;
}
// one workgroup per (time, kind): its block counts -> exclusive offsets inside the time's surface (in place), the total -> totals[2 i + kind]
__global__ __launch_bounds__(256) void cxp_k_me_offsets(const cxp_me_desc* D, uint32_t nd, uint32_t* scnt, uint32_t* tcnt, uint32_t* totals, u64* bases) {
    __shared__ uint32_t s[256];
    const uint32_t i = blockIdx.x >> 1, kind = blockIdx.x & 1u;
    const cxp_me_desc d = D[i];
    const uint32_t b0 = kind ? d.tb0 : d.sb0;
    const uint32_t nb = (kind ? d.tn : d.sn) ? ((kind ? d.tn : d.sn) + CXP_ME_BLOCK - 1u) / CXP_ME_BLOCK : 0u;
    uint32_t* cnt = (kind ? tcnt : scnt) + b0;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += 256u) {     // block-uniform
        const uint32_t k = base + threadIdx.x;
        const uint32_t v = k < nb ? cnt[k] : 0u;
        uint32_t tot;
        const uint32_t e = cxp_block_excl_t(v, s, tot);
        if (k < nb) cnt[k] = carry + e;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        totals[2u * i + kind] = carry;
        if (nd == 1u) bases[kind] = 0ull;      // (one surface: nothing before it, cxp_k_me_bases is not launched)
    }
}
// exclusive prefix of the totals over the times, both kinds (one workgroup; 64-bit: the sums of many surfaces may pass 2^32)
__global__ __launch_bounds__(256) void cxp_k_me_bases(const uint32_t* totals, uint32_t nd, u64* bases) {
    __shared__ u64 sp[256], st[256];
    u64 cp = 0, ct = 0;
    for (uint32_t base = 0; base < nd; base += 256u) {
        const uint32_t k = base + threadIdx.x;
        const u64 vp = k < nd ? totals[2u * k] : 0ull, vt = k < nd ? totals[2u * k + 1u] : 0ull;
        sp[threadIdx.x] = vp; st[threadIdx.x] = vt;
        __syncthreads();
        for (uint32_t o = 1; o < 256; o <<= 1) {
            const u64 xp = (threadIdx.x >= o) ? sp[threadIdx.x - o] : 0ull, xt = (threadIdx.x >= o) ? st[threadIdx.x - o] : 0ull;
            __syncthreads();
            sp[threadIdx.x] += xp; st[threadIdx.x] += xt;
            __syncthreads();
        }
        if (k < nd) { bases[2u * k] = cp + sp[threadIdx.x] - vp; bases[2u * k + 1u] = ct + st[threadIdx.x] - vt; }
        cp += sp[255]; ct += st[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) { bases[2u * nd] = cp; bases[2u * nd + 1u] = ct; }
}
// (A block whose count is zero leaves before it reads a flag.)  The consumers work in two steps: the positions of the block's set flags go
// into a list in LDS (the scan inside the block), then ONE THREAD PER LIST ENTRY does the work -- every lane busy, the gathers of a wave
// independent of each other, the stores of neighbouring lanes next to each other.  (First version: every thread walked the set bits of its
// own 16 flags, three of them on average, each round a chain of dependent gathers and a 24-byte store of its own: 0.67 + 0.71 ms for the 64
// surfaces of config 4.)
__device__ __forceinline__ uint32_t cxp_me_list(uint32_t m, uint16_t* list, uint32_t* s) {
    uint32_t tot;
    uint32_t pos = cxp_block_excl_t(__popc(m), s, tot);
    while (m) {
        const uint32_t k = __ffs(m) - 1u;
        m &= m - 1u;
        list[pos++] = (uint16_t)(threadIdx.x * 16u + k);
    }
    __syncthreads();
    return tot;
}
__global__ __launch_bounds__(256) void cxp_k_me_points(const cxp_me_desc* D, uint32_t nd, const double* P4, const int32_t* segs, uint8_t* sused,
                                                       const uint32_t* soff, const uint32_t* totals, const u64* bases, uint32_t* snew, double* out) {
    __shared__ uint32_t s[256];
    __shared__ uint16_t list[CXP_ME_BLOCK];
    const uint32_t blk = blockIdx.x;
    const uint32_t i = cxp_me_time_of_sblock(D, nd, blk);
    const cxp_me_desc d = D[i];
    const uint32_t nb = (d.sn + CXP_ME_BLOCK - 1u) / CXP_ME_BLOCK;
    const uint32_t rel = blk - d.sb0;
    if (rel >= nb) return;
    const uint32_t first = soff[blk];                                    // inside the surface
    const uint32_t next = (rel + 1u < nb) ? soff[blk + 1u] : totals[2u * i];
    if (next == first) return;
    const size_t fbase = (size_t)blk * CXP_ME_BLOCK;
    const uint32_t m = cxp_flags16(sused + fbase + threadIdx.x * 16u);
    if (m) *reinterpret_cast<uint4*>(sused + fbase + threadIdx.x * 16u) = make_uint4(0u, 0u, 0u, 0u);      // (the flags are all zero again when the call ends)
    const uint32_t n = cxp_me_list(m, list, s);
    const u64 gbase = bases[2u * i] + first;
    const double t = d.t;
    const size_t sg0 = (size_t)d.sf + (size_t)rel * CXP_ME_BLOCK;
    for (uint32_t j = threadIdx.x; j < n; j += 256u) {
        const uint32_t x = list[j];
        snew[fbase + x] = first + j;
        const int2 ab = *reinterpret_cast<const int2*>(segs + (sg0 + x) * 2);
        const double4 a = *reinterpret_cast<const double4*>(P4 + (size_t)ab.x * 4);
        const double4 b = *reinterpret_cast<const double4*>(P4 + (size_t)ab.y * 4);
        const double lo = a.w, hi = b.w;
        const double lam = (hi > lo) ? (t - lo) / (hi - lo) : 0.0;
        double* o = out + (size_t)(gbase + j) * 3;
        o[0] = a.x + lam * (b.x - a.x); o[1] = a.y + lam * (b.y - a.y); o[2] = a.z + lam * (b.z - a.z);
    }
}
__global__ __launch_bounds__(256) void cxp_k_me_tris(const cxp_me_desc* D, uint32_t nd, const int32_t* tris, const uint8_t* tflag, const uint32_t* toff,
                                                     const uint32_t* totals, const u64* bases, const uint32_t* snew, int32_t* out) {
    __shared__ uint32_t s[256];
    __shared__ uint16_t list[CXP_ME_BLOCK];
    const uint32_t blk = blockIdx.x;
    const uint32_t i = cxp_me_time_of_tblock(D, nd, blk);
    const cxp_me_desc d = D[i];
    const uint32_t nb = (d.tn + CXP_ME_BLOCK - 1u) / CXP_ME_BLOCK;
    const uint32_t rel = blk - d.tb0;
    if (rel >= nb) return;
    const uint32_t first = toff[blk];
    const uint32_t next = (rel + 1u < nb) ? toff[blk + 1u] : totals[2u * i + 1u];
    if (next == first) return;
    const uint32_t m = cxp_flags16(tflag + (size_t)blk * CXP_ME_BLOCK + threadIdx.x * 16u);
    const uint32_t n = cxp_me_list(m, list, s);
    const u64 gbase = bases[2u * i + 1u] + first;
    const uint32_t* sn_ = snew + (size_t)d.sb0 * CXP_ME_BLOCK;
    const size_t q0 = (size_t)d.tf + (size_t)rel * CXP_ME_BLOCK;
    for (uint32_t j = threadIdx.x; j < n; j += 256u) {
        const size_t q = q0 + list[j];
        const uint32_t sa = (uint32_t)tris[q * 3] - d.sf, sb = (uint32_t)tris[q * 3 + 1] - d.sf, sc = (uint32_t)tris[q * 3 + 2] - d.sf;
        const uint32_t a = sn_[sa], b = sn_[sb], c = sn_[sc];
        int32_t* o = out + (size_t)(gbase + j) * 3;
        o[0] = (int32_t)a; o[1] = (int32_t)b; o[2] = (int32_t)c;
    }
}
#define CXP_ME_MAX_TIMES 65536
#define CXP_ME_MAX_BYTES (96ull << 30)
extern "C" int cx_morph_eval_many(cx_ctx* ctx, const double* times, int32_t n_times, int64_t* out_counts) {
    if (!ctx || !ctx->post || n_times < 0 || (n_times && !times)) return CX_ERR_INVALID;
    if (n_times > CXP_ME_MAX_TIMES) { ctx->err = "cx_morph_eval_many: more than 65536 times in one call"; return CX_ERR_INVALID; }
    cx_post_state* S = ctx->post;
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t ns = (uint32_t)S->ms_out, nt = (uint32_t)S->mt_out, nd = (uint32_t)n_times;
    S->me_off.assign((size_t)nd * 4, 0);
    if (out_counts) memset(out_counts, 0, (size_t)nd * 2 * sizeof(int64_t));
    if (!nd || !ns || !nt) return CX_OK;
    if (!S->msorted) { ctx->err = "cx_morph_eval: no sorted morph triangles (cx_morph_triangles first)"; return CX_ERR_STATE; }
    for (uint32_t i = 0; i < nd; i++)
        if (!(times[i] == times[i])) { ctx->err = "cx_morph_eval: a time is not a number"; return CX_ERR_INVALID; }
    int rc;
    // the windows: a triangle (segment) that exists at t starts in [t - longest life, t].  The bin of a start time is computed here exactly
    // as on the device (one subtraction, one product with the same factor: nothing to contract, the same double everywhere), and it is
    // monotone in its argument, so the bins of the two ends bound the window; the lower end gives way by 1e-9 of the time range for the
    // rounding of `end - start` and `t - longest life`.  (First version: a bin of slack on either side -- 8.4 bins tested per time on config 4
    // where 6.4 hold everything.)
    auto bin_of = [&](double x) {
        const double b = (x - S->mt_lo) * S->mt_inv_width;
        return b <= 0.0 ? 0u : (b >= (double)(CXP_SB_BINS - 1u) ? CXP_SB_BINS - 1u : (uint32_t)b);
    };
    const double give = S->mt_inv_width > 0.0 ? 1e-9 * ((double)CXP_SB_BINS / S->mt_inv_width) : 0.0;
    // (descriptors and totals of up to 1 151 times travel through pinned memory, so that neither copy is staged by the runtime; measured
    // on config 4: 7.8 ms for 64 single calls either way -- a call for ONE time is five dependent launches of 2-23 us each, 0.12 ms)
    if (!S->me_pinned && hipHostMalloc(&S->me_pinned, 49152) != hipSuccess) { S->me_pinned = nullptr; (void)hipGetLastError(); }
    const bool pin = S->me_pinned && (size_t)(nd + 1) * sizeof(cxp_me_desc) <= 36864 && ((size_t)2 * nd + 1) * sizeof(uint32_t) <= 12288;
    std::vector<cxp_me_desc> Dv(pin ? 0 : nd + 1);
    cxp_me_desc* D = pin ? (cxp_me_desc*)S->me_pinned : Dv.data();
    uint64_t tblocks = 0, sblocks = 0, pts_bound = 0, tri_bound = 0;
    for (uint32_t i = 0; i < nd; i++) {
        const double t = times[i];
        const uint32_t b_hi = bin_of(t);
        const uint32_t bt = bin_of(t - S->mt_maxdur - give), bs = bin_of(t - S->ms_maxdur - give);
        cxp_me_desc& d = D[i];
        d.t = t;
        d.tf = S->mbin_t[bt]; d.tn = S->mbin_t[b_hi + 1u] - d.tf;
        d.sf = S->mbin_s[bs]; d.sn = S->mbin_s[b_hi + 1u] - d.sf;
        d.tb0 = (uint32_t)tblocks; d.sb0 = (uint32_t)sblocks;
        tblocks += (d.tn + CXP_ME_BLOCK - 1u) / CXP_ME_BLOCK;
        sblocks += (d.sn + CXP_ME_BLOCK - 1u) / CXP_ME_BLOCK;
        pts_bound += std::min<uint64_t>(d.sn, 3ull * d.tn);
        tri_bound += d.tn;
    }
    D[nd].t = 0.0; D[nd].tf = D[nd].tn = D[nd].sf = D[nd].sn = 0; D[nd].tb0 = (uint32_t)tblocks; D[nd].sb0 = (uint32_t)sblocks;
    // the surfaces cannot hold more triangles than their windows, nor more points than the windows' segments (or three per triangle):
    // everything is reserved before anything is enqueued, so that the call runs to its end without the host in the middle
    const uint64_t bytes = (tblocks + sblocks) * (uint64_t)CXP_ME_BLOCK * 5ull + pts_bound * 24ull + tri_bound * 12ull;
    if (tblocks + sblocks >= (1ull << 20) * 16ull || bytes > CXP_ME_MAX_BYTES) {
        ctx->err = "cx_morph_eval_many: the windows of these times need more than 96 GB of work space: fewer times per call";
        return CX_ERR_NOMEM;
    }
    if (!tblocks || !sblocks) return CX_OK;       // every window is empty: every surface is
    const uint32_t ntb = (uint32_t)tblocks, nsb = (uint32_t)sblocks;
    {
        const size_t before = S->meflags.bytes;     // (not the address: a freed block often comes back at the same one, larger)
        if ((rc = cxp_reserve(ctx, S->meflags, (size_t)sblocks * CXP_ME_BLOCK + 256))) return rc;
        if (S->meflags.bytes != before) S->meflags_clean = false;
    }
    if ((rc = cxp_reserve(ctx, S->metflag, (size_t)tblocks * CXP_ME_BLOCK + 256))) return rc;
    if ((rc = cxp_reserve(ctx, S->menew, (size_t)sblocks * CXP_ME_BLOCK * sizeof(uint32_t) + 256))) return rc;
    if ((rc = cxp_reserve(ctx, S->mecnt, ((size_t)tblocks + sblocks + 4ull * nd + 64) * sizeof(uint32_t) + (2ull * nd + 8) * sizeof(u64)))) return rc;
    if ((rc = cxp_reserve(ctx, S->medesc, (size_t)(nd + 1) * sizeof(cxp_me_desc)))) return rc;
    if ((rc = cxp_reserve(ctx, S->me_pts, (size_t)(pts_bound + 1) * 3 * sizeof(double)))) return rc;
    if ((rc = cxp_reserve(ctx, S->me_tri, (size_t)(tri_bound + 1) * 3 * sizeof(int32_t)))) return rc;
    // (the segment flags are a buffer of their own, ALL of it zero between two calls: the next call's windows may be laid out differently)
    uint8_t* sused = (uint8_t*)S->meflags.p;
    uint8_t* tflag = (uint8_t*)S->metflag.p;
    uint32_t* snew = (uint32_t*)S->menew.p;
    u64* bases = (u64*)S->mecnt.p;                                   // 2 nd + 2 words of 64 bits first (alignment)
    uint32_t* scnt = (uint32_t*)(bases + 2ull * nd + 8);
    uint32_t* tcnt = scnt + nsb + 8;
    uint32_t* totals = tcnt + ntb + 8;
    uint32_t* err = totals + 2ull * nd;                              // (right behind the totals: one copy back for both)
    cxp_me_desc* Dd = (cxp_me_desc*)S->medesc.p;
    const double* P4 = (const double*)S->pts.p;
    const int32_t* segs = (const int32_t*)S->msegs.p;
    const int32_t* tris = (const int32_t*)S->mtris.p;
    const double* ttime = (const double*)S->mtime.p + (size_t)(ns + 1) * 2;   // behind the segments' ranges (cx_morph_triangles)
    // (the segment flags count as zeroed only while the last call ran to its end; they may have moved since: a larger call reserves anew)
    const bool clean = S->meflags_clean;
    S->meflags_clean = false;
    if (!clean) CXP_HIP(ctx, hipMemsetAsync(sused, 0, S->meflags.bytes, st));
    CXP_HIP(ctx, hipMemcpyAsync(Dd, D, (size_t)(nd + 1) * sizeof(cxp_me_desc), hipMemcpyHostToDevice, st));
    CXP_HIP(ctx, hipMemsetAsync(err, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(cxp_k_me_visible, dim3(ntb * 16u), dim3(256), 0, st, (const cxp_me_desc*)Dd, nd, ttime, tris, tflag, sused, err);
    hipLaunchKernelGGL(cxp_k_me_count16, dim3(nsb + ntb), dim3(256), 0, st, (const uint8_t*)sused, nsb, (const uint8_t*)tflag, scnt, tcnt);
    hipLaunchKernelGGL(cxp_k_me_offsets, dim3(2u * nd), dim3(256), 0, st, (const cxp_me_desc*)Dd, nd, scnt, tcnt, totals, bases);
    if (nd > 1u) hipLaunchKernelGGL(cxp_k_me_bases, dim3(1), dim3(256), 0, st, (const uint32_t*)totals, nd, bases);
    hipLaunchKernelGGL(cxp_k_me_points, dim3(nsb), dim3(256), 0, st, (const cxp_me_desc*)Dd, nd, P4, segs, sused, (const uint32_t*)scnt, (const uint32_t*)totals,
                       (const u64*)bases, snew, (double*)S->me_pts.p);
    hipLaunchKernelGGL(cxp_k_me_tris, dim3(ntb), dim3(256), 0, st, (const cxp_me_desc*)Dd, nd, tris, (const uint8_t*)tflag, (const uint32_t*)tcnt,
                       (const uint32_t*)totals, (const u64*)bases, (const uint32_t*)snew, (int32_t*)S->me_tri.p);
    std::vector<uint32_t> totv(pin ? 0 : (size_t)2 * nd + 1);
    uint32_t* tot = pin ? (uint32_t*)((char*)S->me_pinned + 36864) : totv.data();
    CXP_HIP(ctx, hipMemcpyAsync(tot, totals, ((size_t)2 * nd + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    CXP_HIP(ctx, hipStreamSynchronize(st));
    CXP_HIP(ctx, hipGetLastError());
    if (tot[2 * nd]) { ctx->err = "cx_morph_eval: a visible triangle uses a segment outside its time's window (internal error)"; return CX_ERR_HIP; }
    S->meflags_clean = true;
    int64_t p0 = 0, t0 = 0;
    for (uint32_t i = 0; i < nd; i++) {
        S->me_off[4 * i] = p0; S->me_off[4 * i + 1] = tot[2 * i]; S->me_off[4 * i + 2] = t0; S->me_off[4 * i + 3] = tot[2 * i + 1];
        p0 += tot[2 * i]; t0 += tot[2 * i + 1];
        if (out_counts) { out_counts[2 * i] = tot[2 * i]; out_counts[2 * i + 1] = tot[2 * i + 1]; }
    }
    return CX_OK;
}
extern "C" int cx_morph_eval_many_download(cx_ctx* ctx, int32_t i, double* points_xyz, int32_t* triangles) {
    if (!ctx || !ctx->post) return CX_ERR_INVALID;
    cx_post_state* S = ctx->post;
    if (i < 0 || (size_t)i * 4 >= S->me_off.size()) { ctx->err = "cx_morph_eval_many_download: no such surface in the last call"; return CX_ERR_INVALID; }
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t* o = &S->me_off[(size_t)i * 4];
    void* d[2] = {(points_xyz && o[1]) ? (void*)points_xyz : nullptr, (triangles && o[3]) ? (void*)triangles : nullptr};
    const void* sp[2] = {o[1] ? (const void*)((const double*)S->me_pts.p + (size_t)o[0] * 3) : nullptr,
                         o[3] ? (const void*)((const int32_t*)S->me_tri.p + (size_t)o[2] * 3) : nullptr};
    const size_t nb[2] = {(size_t)o[1] * 3 * sizeof(double), (size_t)o[3] * 3 * sizeof(int32_t)};
    return cx_copy_to_host(ctx, 2, d, sp, nb);
}
// all surfaces of the last call in one transfer: points of surface 0, 1, ... one behind the other (sum of the point counts x 3 doubles),
// triangles likewise (indices local to their surface) -- one pipelined copy instead of two small ones per surface
extern "C" int cx_morph_eval_many_download_all(cx_ctx* ctx, double* points_xyz, int32_t* triangles) {
    if (!ctx || !ctx->post) return CX_ERR_INVALID;
    cx_post_state* S = ctx->post;
    CXP_HIP(ctx, hipSetDevice(ctx->device));
    int64_t np_ = 0, nt_ = 0;
    for (size_t i = 0; i + 3 < S->me_off.size(); i += 4) { np_ += S->me_off[i + 1]; nt_ += S->me_off[i + 3]; }
    void* d[2] = {(points_xyz && np_) ? (void*)points_xyz : nullptr, (triangles && nt_) ? (void*)triangles : nullptr};
    const void* sp[2] = {S->me_pts.p, S->me_tri.p};
    const size_t nb[2] = {(size_t)np_ * 3 * sizeof(double), (size_t)nt_ * 3 * sizeof(int32_t)};
    return cx_copy_to_host(ctx, 2, d, sp, nb);
}
extern "C" int cx_morph_eval_many_device_ptrs(cx_ctx* ctx, int32_t i, void** points_xyz, void** triangles) {
    if (!ctx || !ctx->post) return CX_ERR_INVALID;
    cx_post_state* S = ctx->post;
    if (i < 0 || (size_t)i * 4 >= S->me_off.size()) { ctx->err = "cx_morph_eval_many_device_ptrs: no such surface in the last call"; return CX_ERR_INVALID; }
    const int64_t* o = &S->me_off[(size_t)i * 4];
    if (points_xyz) *points_xyz = o[1] ? (void*)((double*)S->me_pts.p + (size_t)o[0] * 3) : nullptr;
    if (triangles) *triangles = o[3] ? (void*)((int32_t*)S->me_tri.p + (size_t)o[2] * 3) : nullptr;
    return CX_OK;
}
extern "C" int cx_morph_eval(cx_ctx* ctx, double t, int64_t* out_counts) {
    return cx_morph_eval_many(ctx, &t, 1, out_counts);
}
extern "C" int cx_morph_eval_download(cx_ctx* ctx, double* points_xyz, int32_t* triangles) {
    if (!ctx || !ctx->post) return CX_ERR_INVALID;
    if (ctx->post->me_off.empty()) return CX_OK;      // nothing evaluated: nothing to copy
    return cx_morph_eval_many_download(ctx, 0, points_xyz, triangles);
}
