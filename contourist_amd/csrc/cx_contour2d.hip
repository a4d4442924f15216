// cx_contour2d.hip -- 2-D contour lines at several isovalues in one pass (C ABI: cx_contour2d_extract / _download).
//
// Reference semantics restated (contourist/triangulated.py, contourist/multiple_2d_contour.py):
//   adjacent_offsets            triangulated.py:10-12    the lattice is triangulated with the (1,1) diagonal
//   contour_pair_interpolation  :339-353   pair (low, high): f(low) <= z <= f(high); ratio 0.5 if |fhigh-flow| <= 1e-8
//   adjacent_pairs              :63-75     two pairs are adjacent when they share a triangle (and so an end point)
//   expand_contour_pairs        :322-331   growth from the seeds over pairs that share an end point in the same role
//   search_grid                 :198-212   without end points: every crossing lattice edge (i,j)-(i+1,j), (i,j)-(i,j+1)
//                                          with i < n-1 and j < m-1 is a seed
//   get_contour_sequences       :226-297   walk over adjacencies -> polylines; a point np.allclose to the previous
//                                          one is dropped; closed when it has no end or first and last are np.allclose
//   classify_endpoint_values    multiple_2d_contour.py:48-59   the levels a segment crosses, by bisection in the sorted values
// Here: a sample equal to the isovalue counts as high only (f < z is low, as in the 3-D march), so every crossed
// triangle holds exactly one segment and every pair has at most two neighbours.  Segments are directed with the low
// side on the left, which gives every crossing one successor and one predecessor; polylines are then ranked with
// pointer jumping.  No atomics on the data path except the two lock-free union-finds (growth groups, chains).
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "cx_ctx.h"

#pragma clang fp contract(off)   // low + ratio*(high-low) and grid*delta + mins round like the reference's float64

#define C2_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return (e__ == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;                      \
        }                                                                                        \
    } while (0)

#define C2_NIL 0xFFFFFFFFu
typedef unsigned long long c2_u64;

struct c2_buf {
    void* p = nullptr;
    size_t cap = 0;
};
struct cx_state2 {
    c2_buf grid, values, cnt, base, sums, pts, keys, succ, pred, parent, cparent, mark, head, cyc, ptr[2], dist[2], len, hflag, cidx,
        chead, clen, coff, opts, okeys, ochain, keep, fidx, fpts, fkeys, chains, scal, seeds;
    cx_counts2d counts = {0, 0, 0, 0};
    bool valid = false;
};

static int c2_reserve(cx_ctx* ctx, c2_buf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return CX_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc(2-D contour buffers): ") + hipGetErrorString(e);
        b.p = nullptr;
        return CX_ERR_NOMEM;
    }
    b.cap = want;
    return CX_OK;
}
void cx_state2_free(cx_ctx* ctx) {
    if (!ctx->s2) return;
    cx_state2* S = ctx->s2;
    c2_buf* all[] = {&S->grid, &S->values, &S->cnt, &S->base, &S->sums, &S->pts, &S->keys, &S->succ, &S->pred, &S->parent, &S->cparent,
                     &S->mark, &S->head, &S->cyc, &S->ptr[0], &S->ptr[1], &S->dist[0], &S->dist[1], &S->len, &S->hflag, &S->cidx, &S->chead,
                     &S->clen, &S->coff, &S->opts, &S->okeys, &S->ochain, &S->keep, &S->fidx, &S->fpts, &S->fkeys, &S->chains, &S->scal,
                     &S->seeds};
    for (c2_buf* b : all)
        if (b->p) (void)hipFree(b->p);
    delete S;
    ctx->s2 = nullptr;
}

struct c2_grid {
    const float* A;
    uint32_t n, m;          // samples per axis; A[i*m + j]
    const double* values;   // sorted ascending, distinct
    uint32_t nvalues;
    const uint32_t* base;   // exclusive scan of the crossings per lattice edge, edge = 3*(i*m+j) + d
};
// the three forward lattice edges of a point: d 0: (1,0), 1: (0,1), 2: (1,1)
__device__ __forceinline__ bool c2_edge_valid(const c2_grid& G, uint32_t i, uint32_t j, int d) {
    return (d == 1 || i + 1 < G.n) && (d == 0 || j + 1 < G.m);
}
// a NaN sample counts as +inf everywhere, so that counting and linking agree about which edges are crossed
__device__ __forceinline__ double c2_f(const c2_grid& G, int i, int j) {
    const float v = G.A[(size_t)i * G.m + (size_t)j];
    return (v == v) ? (double)v : (double)INFINITY;
}
// first index with values[idx] > x
__device__ __forceinline__ uint32_t c2_upper(const c2_grid& G, double x) {
    uint32_t lo = 0, hi = G.nvalues;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (G.values[mid] > x) hi = mid; else lo = mid + 1;
    }
    return lo;
}
// levels z with min(fa,fb) < z <= max(fa,fb): [start, end)
__device__ __forceinline__ void c2_levels(const c2_grid& G, double fa, double fb, uint32_t& start, uint32_t& end) {
    if (!(fa == fa) || !(fb == fb) || fa == fb) { start = end = 0; return; }
    start = c2_upper(G, fmin(fa, fb));
    end = c2_upper(G, fmax(fa, fb));
}
// id of the crossing of level lvl on the lattice edge between (ai,aj) and (bi,bj) (neighbours, either order)
__device__ __forceinline__ uint32_t c2_id_of(const c2_grid& G, int ai, int aj, double fa, int bi, int bj, double fb, uint32_t lvl) {
    if (bi < ai || bj < aj) {
        const int ti = ai, tj = aj; ai = bi; aj = bj; bi = ti; bj = tj;
    }
    const int d = (bi > ai) ? ((bj > aj) ? 2 : 0) : 1;
    const uint32_t e = 3u * ((uint32_t)ai * G.m + (uint32_t)aj) + (uint32_t)d;
    return G.base[e] + (lvl - c2_upper(G, fmin(fa, fb)));
}

__global__ void c2_k_count(c2_grid G, uint32_t* cnt) {
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= G.n * G.m) return;
    const uint32_t i = lin / G.m, j = lin - i * G.m;
    const double f0 = c2_f(G, i, j);
    for (int d = 0; d < 3; d++) {
        uint32_t c = 0;
        if (c2_edge_valid(G, i, j, d)) {
            uint32_t s, e;
            c2_levels(G, f0, c2_f(G, i + (d != 1), j + (d != 0)), s, e);
            c = e - s;
        }
        cnt[3u * lin + d] = c;
    }
}

// the segment of level z inside the counter-clockwise triangle (v0, v1, v2): links crossing `id`, which lies on the
// edge between triangle corners ea and eb, to the other crossing of the triangle
__device__ __forceinline__ void c2_link(const c2_grid& G, const int vi[3], const int vj[3], int ea, int eb, uint32_t lvl, double z, uint32_t id,
                                        uint32_t* succ, uint32_t* pred) {
    double f[3];
    int nlow = 0;
    bool low[3];
    for (int k = 0; k < 3; k++) {
        f[k] = c2_f(G, vi[k], vj[k]);
        low[k] = f[k] < z;
        nlow += low[k] ? 1 : 0;
    }
    if (nlow == 0 || nlow == 3) return;
    const bool s_low = (nlow == 1);
    int s = 0;
    for (int k = 0; k < 3; k++)
        if (low[k] == s_low) s = k;
    const int x = (s + 1) % 3, y = (s + 2) % 3;
    // low side on the left: alone corner low -> from edge (s,x) to edge (s,y); alone corner high -> the other way
    const int tail_other = s_low ? x : y, head_other = s_low ? y : x;
    const int mine_other = (ea == s) ? eb : ea;
    if (mine_other == tail_other)
        succ[id] = c2_id_of(G, vi[s], vj[s], f[s], vi[head_other], vj[head_other], f[head_other], lvl);
    else
        pred[id] = c2_id_of(G, vi[s], vj[s], f[s], vi[tail_other], vj[tail_other], f[tail_other], lvl);
}

__global__ void c2_k_emit(c2_grid G, double2* pts, c2_u64* keys, uint32_t* succ, uint32_t* pred, uint32_t* parent, uint32_t* cparent) {
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= G.n * G.m) return;
    const int i = (int)(lin / G.m), j = (int)(lin - (uint32_t)i * G.m);
    const double f0 = c2_f(G, i, j);
    for (int d = 0; d < 3; d++) {
        if (!c2_edge_valid(G, i, j, d)) continue;
        const int bi = i + (d != 1), bj = j + (d != 0);
        const double f1 = c2_f(G, bi, bj);
        uint32_t s, e;
        c2_levels(G, f0, f1, s, e);
        if (s == e) continue;
        const uint32_t id0 = G.base[3u * lin + d];
        // oriented pair (triangulated.py:339-353)
        const bool first_low = f0 < f1;
        const double flow = first_low ? f0 : f1, fhigh = first_low ? f1 : f0;
        const double li = first_low ? i : bi, lj = first_low ? j : bj, hi = first_low ? bi : i, hj = first_low ? bj : j;
        const double den = 1.0 * (fhigh - flow);
        // the two triangles of the edge, corners counter-clockwise in (i, j)
        int t1i[3], t1j[3], t2i[3], t2j[3], a1, b1, a2, b2;
        bool has1, has2;
        if (d == 0) {
            has1 = j + 1 < (int)G.m;  t1i[0] = i; t1j[0] = j; t1i[1] = i + 1; t1j[1] = j; t1i[2] = i + 1; t1j[2] = j + 1; a1 = 0; b1 = 1;
            has2 = j >= 1;            t2i[0] = i; t2j[0] = j - 1; t2i[1] = i + 1; t2j[1] = j; t2i[2] = i; t2j[2] = j; a2 = 2; b2 = 1;
        } else if (d == 1) {
            has1 = i + 1 < (int)G.n;  t1i[0] = i; t1j[0] = j; t1i[1] = i + 1; t1j[1] = j + 1; t1i[2] = i; t1j[2] = j + 1; a1 = 0; b1 = 2;
            has2 = i >= 1;            t2i[0] = i - 1; t2j[0] = j; t2i[1] = i; t2j[1] = j; t2i[2] = i; t2j[2] = j + 1; a2 = 1; b2 = 2;
        } else {
            has1 = true;              t1i[0] = i; t1j[0] = j; t1i[1] = i + 1; t1j[1] = j; t1i[2] = i + 1; t1j[2] = j + 1; a1 = 0; b1 = 2;
            has2 = true;              t2i[0] = i; t2j[0] = j; t2i[1] = i + 1; t2j[1] = j + 1; t2i[2] = i; t2j[2] = j + 1; a2 = 0; b2 = 1;
        }
        for (uint32_t lvl = s; lvl < e; lvl++) {
            const uint32_t id = id0 + (lvl - s);
            const double z = G.values[lvl];
            double ratio = 0.5;
            if (!(fabs(den) <= 1e-8)) ratio = (z - flow) / den;
            pts[id] = make_double2(li + ratio * (hi - li), lj + ratio * (hj - lj));
            keys[id] = ((c2_u64)(3u * lin + (uint32_t)d) << 16) | (c2_u64)lvl;
            succ[id] = C2_NIL;
            pred[id] = C2_NIL;
            parent[id] = id;
            cparent[id] = id;
            if (has1) c2_link(G, t1i, t1j, a1, b1, lvl, z, id, succ, pred);
            if (has2) c2_link(G, t2i, t2j, a2, b2, lvl, z, id, succ, pred);
        }
    }
}

// ---- lock-free union-find, root = smallest id ---------------------------------------------------------
__device__ __forceinline__ uint32_t c2_find(uint32_t* parent, uint32_t x) {
    for (;;) {
        const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        const uint32_t g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != p) atomicCAS(&parent[x], p, g);
        x = p;
    }
}
__device__ __forceinline__ void c2_union(uint32_t* parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = c2_find(parent, a);
        b = c2_find(parent, b);
        if (a == b) return;
        const uint32_t win = min(a, b), lose = max(a, b);
        if (atomicCAS(&parent[lose], lose, win) == lose) return;
    }
}
__global__ void c2_k_flatten(uint32_t* parent, uint32_t n) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        const uint32_t root = c2_find(parent, r);
        __hip_atomic_store(&parent[r], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __constant__ int c2_off_i[6] = {0, 1, 1, 0, -1, -1};   // adjacent_offsets (triangulated.py:10-12)
__device__ __constant__ int c2_off_j[6] = {1, 1, 0, -1, -1, 0};
// the first pair around point (i,j) in the given role at level lvl (role 0: the point is the low end), or C2_NIL
__device__ __forceinline__ uint32_t c2_first_around(const c2_grid& G, int i, int j, int role, uint32_t lvl) {
    if (i < 0 || j < 0 || i >= (int)G.n || j >= (int)G.m) return C2_NIL;
    const double z = G.values[lvl], f0 = c2_f(G, i, j);
    if (!(f0 == f0) || ((f0 < z) != (role == 0))) return C2_NIL;
    for (int k = 0; k < 6; k++) {
        const int qi = i + c2_off_i[k], qj = j + c2_off_j[k];
        if (qi < 0 || qj < 0 || qi >= (int)G.n || qj >= (int)G.m) continue;   // in_range (:333-334)
        const double fq = c2_f(G, qi, qj);
        if (!(fq == fq)) continue;
        if ((fq < z) == (role == 0)) continue;   // same side: no pair
        return c2_id_of(G, i, j, f0, qi, qj, fq, lvl);
    }
    return C2_NIL;
}
__device__ __forceinline__ void c2_decode(const c2_grid& G, c2_u64 key, int& i, int& j, int& bi, int& bj, uint32_t& lvl) {
    lvl = (uint32_t)(key & 0xFFFFull);
    const uint32_t e = (uint32_t)(key >> 16), lin = e / 3u, d = e - 3u * lin;
    i = (int)(lin / G.m);
    j = (int)(lin - (uint32_t)i * G.m);
    bi = i + (d != 1u);
    bj = j + (d != 0u);
}
// pairs that share an end point in the same role belong to one growth group (expand_contour_pairs :322-331)
__global__ void c2_k_group(c2_grid G, const c2_u64* keys, uint32_t nv, uint32_t* parent) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    int i, j, bi, bj;
    uint32_t lvl;
    c2_decode(G, keys[id], i, j, bi, bj, lvl);
    const bool first_low = c2_f(G, i, j) < G.values[lvl];
    const uint32_t ra = c2_first_around(G, i, j, first_low ? 0 : 1, lvl);
    const uint32_t rb = c2_first_around(G, bi, bj, first_low ? 1 : 0, lvl);
    if (ra != C2_NIL && ra != id) c2_union(parent, id, ra);
    if (rb != C2_NIL && rb != id) c2_union(parent, id, rb);
}
// first index with values[idx] >= x
__device__ __forceinline__ uint32_t c2_lower(const c2_grid& G, double x) {
    uint32_t lo = 0, hi = G.nvalues;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (G.values[mid] >= x) hi = mid; else lo = mid + 1;
    }
    return lo;
}
// marks the growth groups of the pairs around a seed point.  role 0: the reference took the point as the low end of
// its seed (f <= z); a point equal to z is a high point here, so its pairs are looked up in that role
__device__ __forceinline__ void c2_mark_seed(const c2_grid& G, int i, int j, int role, uint32_t lvl, uint32_t* parent, uint32_t* mark) {
    if (i < 0 || j < 0 || i >= (int)G.n || j >= (int)G.m) return;
    if (role == 0 && c2_f(G, i, j) == G.values[lvl]) role = 1;
    const uint32_t r = c2_first_around(G, i, j, role, lvl);
    if (r != C2_NIL) mark[c2_find(parent, r)] = 1u;
}
// seeds of the exhaustive search (search_grid :198-212): the axis edges from (i,j), i < n-1, j < m-1, with
// f(low) <= z <= f(high); both end points seed (find_initial_contour_pairs :316-317)
__global__ void c2_k_seed_search(c2_grid G, uint32_t* parent, uint32_t* mark) {
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= G.n * G.m) return;
    const int i = (int)(lin / G.m), j = (int)(lin - (uint32_t)i * G.m);
    if (i + 1 >= (int)G.n || j + 1 >= (int)G.m) return;
    const double f0 = c2_f(G, i, j);
    for (int d = 0; d < 2; d++) {
        const int bi = i + (d == 0), bj = j + (d == 1);
        const double f1 = c2_f(G, bi, bj);
        const bool fwd = f0 <= f1;
        const uint32_t s = c2_lower(G, fmin(f0, f1)), e = c2_upper(G, fmax(f0, f1));
        for (uint32_t lvl = s; lvl < e; lvl++) {
            c2_mark_seed(G, fwd ? i : bi, fwd ? j : bj, 0, lvl, parent, mark);
            c2_mark_seed(G, fwd ? bi : i, fwd ? bj : j, 1, lvl, parent, mark);
        }
    }
}
// explicit seeds: (i, j, role, level index) -- the pairs around the point in that role (find_initial_contour_pairs :316-317)
__global__ void c2_k_seed_points(c2_grid G, const int32_t* seeds, uint32_t nseeds, uint32_t* parent, uint32_t* mark) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseeds) return;
    const int32_t* q = seeds + 4 * (size_t)s;
    if (q[3] < 0 || q[3] >= (int)G.nvalues || (q[2] != 0 && q[2] != 1)) return;
    c2_mark_seed(G, q[0], q[1], q[2], (uint32_t)q[3], parent, mark);
}

// ---- chains ----------------------------------------------------------------------------------------------
__global__ void c2_k_chain_union(const uint32_t* succ, uint32_t nv, uint32_t* cparent) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    const uint32_t s = succ[id];
    if (s != C2_NIL) c2_union(cparent, id, s);
}
__global__ void c2_k_heads(const uint32_t* pred, const uint32_t* cparent, uint32_t nv, uint32_t* head) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    if (pred[id] == C2_NIL) head[cparent[id]] = id;   // an open chain has exactly one such element
}
// closed chains are opened in front of their smallest id (the root of the union-find)
__global__ void c2_k_cut(uint32_t* succ, uint32_t* pred, const uint32_t* cparent, uint32_t nv, uint32_t* head, uint32_t* cyc) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    cyc[id] = 0u;
    if (cparent[id] != id || head[id] != C2_NIL) return;
    head[id] = id;
    cyc[id] = 1u;
    const uint32_t p = pred[id];
    if (p != C2_NIL) succ[p] = C2_NIL;
    pred[id] = C2_NIL;
}
__global__ void c2_k_rank_init(const uint32_t* pred, uint32_t nv, uint32_t* ptr, uint32_t* dist) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    const uint32_t p = pred[id];
    ptr[id] = (p == C2_NIL) ? id : p;
    dist[id] = (p == C2_NIL) ? 0u : 1u;
}
__global__ void c2_k_rank_jump(const uint32_t* ptr, const uint32_t* dist, uint32_t nv, uint32_t* ptr2, uint32_t* dist2, uint32_t* changed) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    const uint32_t p = ptr[id], pp = ptr[p];
    ptr2[id] = pp;
    dist2[id] = dist[id] + dist[p];   // dist[p] is 0 once p is the head of its chain
    if (pp != p) *changed = 1u;
}
__global__ void c2_k_lengths(const uint32_t* succ, const uint32_t* cparent, const uint32_t* dist, uint32_t nv, uint32_t* len) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    if (succ[id] == C2_NIL) len[cparent[id]] = dist[id] + 1u;
}
// hflag[id] = 1 for the head of a chain that is kept
__global__ void c2_k_head_flags(const uint32_t* cparent, const uint32_t* head, uint32_t* parent, const uint32_t* mark, int all, uint32_t nv,
                                uint32_t* hflag) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    uint32_t f = 0;
    if (head[cparent[id]] == id) f = all ? 1u : (mark[c2_find(parent, id)] ? 1u : 0u);
    hflag[id] = f;
}
__global__ void c2_k_chain_table(const uint32_t* hflag, const uint32_t* cidx, const uint32_t* cparent, const uint32_t* len, uint32_t nv,
                                 uint32_t* chead, uint32_t* clen) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv || !hflag[id]) return;
    const uint32_t c = cidx[id];
    chead[c] = id;
    clen[c] = len[cparent[id]];
}
__global__ void c2_k_place(const double2* pts, const c2_u64* keys, const uint32_t* cparent, const uint32_t* head, const uint32_t* hflag,
                           const uint32_t* cidx, const uint32_t* coff, const uint32_t* dist, uint32_t nv, double2* opts, c2_u64* okeys,
                           uint32_t* ochain) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    const uint32_t h = head[cparent[id]];
    if (!hflag[h]) return;
    const uint32_t c = cidx[h], pos = coff[c] + dist[id];
    opts[pos] = pts[id];
    okeys[pos] = keys[id];
    ochain[pos] = c;
}
// np.allclose(a, b) for 2-vectors
__device__ __forceinline__ bool c2_close(const double2 a, const double2 b) {
    return fabs(a.x - b.x) <= 1e-8 + 1e-5 * fabs(b.x) && fabs(a.y - b.y) <= 1e-8 + 1e-5 * fabs(b.y);
}
__global__ void c2_k_keep(const double2* opts, const uint32_t* ochain, const uint32_t* coff, uint32_t nsel, int dedupe, uint32_t* keep) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nsel) return;
    uint32_t k = 1u;
    if (dedupe && pos != coff[ochain[pos]] && c2_close(opts[pos - 1], opts[pos])) k = 0u;
    keep[pos] = k;
}
__global__ void c2_k_final_points(const double2* opts, const c2_u64* okeys, const uint32_t* keep, const uint32_t* fidx, uint32_t nsel,
                                  const double* scal, double2* fpts, c2_u64* fkeys) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nsel || !keep[pos]) return;
    const double2 p = opts[pos];
    // from_grid_coordinates (grid_field.py:89-93): grid * delta + mins
    fpts[fidx[pos]] = make_double2(p.x * scal[2] + scal[0], p.y * scal[3] + scal[1]);
    fkeys[fidx[pos]] = okeys[pos];
}
__global__ void c2_k_final_chains(const double2* opts, const c2_u64* okeys, const uint32_t* keep, const uint32_t* fidx, const uint32_t* coff,
                                  const uint32_t* clen, const uint32_t* chead, const uint32_t* cyc, uint32_t nchains, uint32_t nfinal,
                                  cx_chain2d* chains) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    const uint32_t first = coff[c], n = clen[c];
    uint32_t last = first + n - 1u;
    while (last > first && !keep[last]) last--;
    cx_chain2d out;
    out.level = (int32_t)(okeys[first] & 0xFFFFull);
    out.first = fidx[first];
    out.count = ((c + 1u < nchains) ? fidx[coff[c + 1u]] : nfinal) - out.first;
    out.closed = (cyc[chead[c]] || c2_close(opts[first], opts[last])) ? 1 : 0;   // :293-294
    chains[c] = out;
}

static inline uint32_t c2_blocks(size_t n) { return (uint32_t)((n + 255) / 256); }

extern "C" int cx_contour2d_extract(cx_ctx* ctx, const float* samples, int on_device, int64_t n, int64_t m, const double* values, int32_t nvalues,
                                    const int32_t* seeds, int64_t nseeds, uint32_t flags, const double* mins_delta, cx_counts2d* out) {
    if (!ctx) return CX_ERR_INVALID;
    ctx->err.clear();
    if (!samples || !values || n < 2 || m < 2 || nvalues < 1 || nvalues > 65535 || (nseeds > 0 && !seeds)) {
        ctx->err = "cx_contour2d_extract: need samples, 2 <= n, m and 1 <= nvalues <= 65535";
        return CX_ERR_INVALID;
    }
    if ((uint64_t)n * (uint64_t)m > (1ull << 30)) {
        ctx->err = "cx_contour2d_extract: more than 2^30 samples";
        return CX_ERR_UNSUPPORTED;
    }
    for (int k = 0; k < nvalues; k++)
        if (!(values[k] == values[k]) || (k > 0 && !(values[k] > values[k - 1]))) {
            ctx->err = "cx_contour2d_extract: values must be finite, distinct and ascending";
            return CX_ERR_INVALID;
        }
    C2_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->s2) ctx->s2 = new cx_state2();
    cx_state2* S = ctx->s2;
    S->valid = false;
    hipStream_t st = ctx->stream;
    const size_t N = (size_t)n * (size_t)m;
    int rc;
    const float* A = samples;
    if (!on_device) {
        if ((rc = c2_reserve(ctx, S->grid, N * sizeof(float)))) return rc;
        C2_HIP(ctx, hipMemcpyAsync(S->grid.p, samples, N * sizeof(float), hipMemcpyHostToDevice, st));
        A = (const float*)S->grid.p;
    }
    if ((rc = c2_reserve(ctx, S->values, (size_t)nvalues * sizeof(double)))) return rc;
    C2_HIP(ctx, hipMemcpyAsync(S->values.p, values, (size_t)nvalues * sizeof(double), hipMemcpyHostToDevice, st));
    double scal_host[4] = {0.0, 0.0, 1.0, 1.0};
    if (mins_delta) memcpy(scal_host, mins_delta, sizeof(scal_host));
    if ((rc = c2_reserve(ctx, S->scal, 64))) return rc;
    C2_HIP(ctx, hipMemcpyAsync(S->scal.p, scal_host, sizeof(scal_host), hipMemcpyHostToDevice, st));
    uint32_t* scratch = (uint32_t*)((char*)S->scal.p + 32);   // [0] scan total, [1] changed flag
    const size_t E = 3 * N;
    if ((rc = c2_reserve(ctx, S->cnt, E * 4)) || (rc = c2_reserve(ctx, S->base, (E + 1) * 4)) || (rc = c2_reserve(ctx, S->sums, (E / 1024 + 4) * 4)))
        return rc;
    c2_grid G{A, (uint32_t)n, (uint32_t)m, (const double*)S->values.p, (uint32_t)nvalues, (const uint32_t*)S->base.p};
    hipLaunchKernelGGL(c2_k_count, dim3(c2_blocks(N)), dim3(256), 0, st, G, (uint32_t*)S->cnt.p);
    cx_scan_u32(ctx, (const uint32_t*)S->cnt.p, (uint32_t*)S->base.p, (uint32_t)E, (uint32_t*)S->sums.p, scratch);
    uint32_t nv = 0;
    C2_HIP(ctx, hipMemcpyAsync(&nv, scratch, 4, hipMemcpyDeviceToHost, st));
    C2_HIP(ctx, hipStreamSynchronize(st));
    S->counts = cx_counts2d{0, 0, nv, (uint32_t)nvalues};
    if (nv == 0) {
        S->valid = true;
        if (out) *out = S->counts;
        return CX_OK;
    }
    if (nv >= 0x7FFFFFFFu) {
        ctx->err = "cx_contour2d_extract: more than 2^31 crossings";
        return CX_ERR_UNSUPPORTED;
    }
    const size_t V = nv;
    c2_buf* u32s[] = {&S->succ, &S->pred, &S->parent, &S->cparent, &S->mark, &S->head, &S->cyc, &S->ptr[0], &S->ptr[1], &S->dist[0], &S->dist[1],
                      &S->len, &S->hflag, &S->cidx, &S->ochain, &S->keep, &S->fidx};
    for (c2_buf* b : u32s)
        if ((rc = c2_reserve(ctx, *b, (V + 1) * 4))) return rc;
    if ((rc = c2_reserve(ctx, S->pts, V * 16)) || (rc = c2_reserve(ctx, S->keys, V * 8)) || (rc = c2_reserve(ctx, S->opts, V * 16)) ||
        (rc = c2_reserve(ctx, S->okeys, V * 8)) || (rc = c2_reserve(ctx, S->fpts, V * 16)) || (rc = c2_reserve(ctx, S->fkeys, V * 8)))
        return rc;
    if (S->sums.cap < (V / 1024 + 4) * 4 && (rc = c2_reserve(ctx, S->sums, (V / 1024 + 4) * 4))) return rc;
    double2* pts = (double2*)S->pts.p;
    c2_u64* keys = (c2_u64*)S->keys.p;
    uint32_t *succ = (uint32_t*)S->succ.p, *pred = (uint32_t*)S->pred.p, *parent = (uint32_t*)S->parent.p, *cparent = (uint32_t*)S->cparent.p;
    uint32_t *mark = (uint32_t*)S->mark.p, *head = (uint32_t*)S->head.p, *cyc = (uint32_t*)S->cyc.p, *len = (uint32_t*)S->len.p;
    uint32_t *hflag = (uint32_t*)S->hflag.p, *cidx = (uint32_t*)S->cidx.p;
    const uint32_t gb = c2_blocks(V);
    hipLaunchKernelGGL(c2_k_emit, dim3(c2_blocks(N)), dim3(256), 0, st, G, pts, keys, succ, pred, parent, cparent);
    // growth groups and seeds
    const int all = (flags & CX2_ALL_CHAINS) ? 1 : 0;
    if (!all) {
        C2_HIP(ctx, hipMemsetAsync(mark, 0, V * 4, st));
        hipLaunchKernelGGL(c2_k_group, dim3(gb), dim3(256), 0, st, G, keys, nv, parent);
        if (nseeds > 0) {
            if ((rc = c2_reserve(ctx, S->seeds, (size_t)nseeds * 16))) return rc;
            C2_HIP(ctx, hipMemcpyAsync(S->seeds.p, seeds, (size_t)nseeds * 16, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(c2_k_seed_points, dim3(c2_blocks((size_t)nseeds)), dim3(256), 0, st, G, (const int32_t*)S->seeds.p, (uint32_t)nseeds,
                               parent, mark);
        }
        if (nseeds <= 0 || (flags & CX2_SEARCH_SEEDS)) hipLaunchKernelGGL(c2_k_seed_search, dim3(c2_blocks(N)), dim3(256), 0, st, G, parent, mark);
    }
    // chains: identity, heads, cuts, ranks
    hipLaunchKernelGGL(c2_k_chain_union, dim3(gb), dim3(256), 0, st, succ, nv, cparent);
    hipLaunchKernelGGL(c2_k_flatten, dim3(gb), dim3(256), 0, st, cparent, nv);
    C2_HIP(ctx, hipMemsetAsync(head, 0xFF, V * 4, st));
    hipLaunchKernelGGL(c2_k_heads, dim3(gb), dim3(256), 0, st, pred, cparent, nv, head);
    hipLaunchKernelGGL(c2_k_cut, dim3(gb), dim3(256), 0, st, succ, pred, cparent, nv, head, cyc);
    int cur = 0;
    hipLaunchKernelGGL(c2_k_rank_init, dim3(gb), dim3(256), 0, st, pred, nv, (uint32_t*)S->ptr[0].p, (uint32_t*)S->dist[0].p);
    for (int round = 0;; round++) {
        if (round >= 34) {
            ctx->err = "cx_contour2d_extract: chain ranking did not converge";
            return CX_ERR_HIP;
        }
        C2_HIP(ctx, hipMemsetAsync(scratch + 1, 0, 4, st));
        hipLaunchKernelGGL(c2_k_rank_jump, dim3(gb), dim3(256), 0, st, (const uint32_t*)S->ptr[cur].p, (const uint32_t*)S->dist[cur].p, nv,
                           (uint32_t*)S->ptr[1 - cur].p, (uint32_t*)S->dist[1 - cur].p, scratch + 1);
        cur = 1 - cur;
        uint32_t changed = 0;
        C2_HIP(ctx, hipMemcpyAsync(&changed, scratch + 1, 4, hipMemcpyDeviceToHost, st));
        C2_HIP(ctx, hipStreamSynchronize(st));
        if (!changed) break;
    }
    const uint32_t* dist = (const uint32_t*)S->dist[cur].p;
    hipLaunchKernelGGL(c2_k_lengths, dim3(gb), dim3(256), 0, st, succ, cparent, dist, nv, len);
    hipLaunchKernelGGL(c2_k_head_flags, dim3(gb), dim3(256), 0, st, cparent, head, parent, mark, all, nv, hflag);
    cx_scan_u32(ctx, hflag, cidx, nv, (uint32_t*)S->sums.p, scratch);
    uint32_t nchains = 0;
    C2_HIP(ctx, hipMemcpyAsync(&nchains, scratch, 4, hipMemcpyDeviceToHost, st));
    C2_HIP(ctx, hipStreamSynchronize(st));
    if (nchains == 0) {
        S->valid = true;
        if (out) *out = S->counts;
        return CX_OK;
    }
    if ((rc = c2_reserve(ctx, S->chead, (size_t)nchains * 4)) || (rc = c2_reserve(ctx, S->clen, (size_t)nchains * 4)) ||
        (rc = c2_reserve(ctx, S->coff, ((size_t)nchains + 1) * 4)) || (rc = c2_reserve(ctx, S->chains, (size_t)nchains * sizeof(cx_chain2d))))
        return rc;
    uint32_t *chead = (uint32_t*)S->chead.p, *clen = (uint32_t*)S->clen.p, *coff = (uint32_t*)S->coff.p;
    hipLaunchKernelGGL(c2_k_chain_table, dim3(gb), dim3(256), 0, st, hflag, cidx, cparent, len, nv, chead, clen);
    cx_scan_u32(ctx, clen, coff, nchains, (uint32_t*)S->sums.p, scratch);
    uint32_t nsel = 0;
    C2_HIP(ctx, hipMemcpyAsync(&nsel, scratch, 4, hipMemcpyDeviceToHost, st));
    C2_HIP(ctx, hipStreamSynchronize(st));
    if (nsel == 0 || nsel > nv) {
        ctx->err = "cx_contour2d_extract: inconsistent chain lengths";
        return CX_ERR_HIP;
    }
    double2* opts = (double2*)S->opts.p;
    c2_u64* okeys = (c2_u64*)S->okeys.p;
    uint32_t *ochain = (uint32_t*)S->ochain.p, *keep = (uint32_t*)S->keep.p, *fidx = (uint32_t*)S->fidx.p;
    hipLaunchKernelGGL(c2_k_place, dim3(gb), dim3(256), 0, st, pts, keys, cparent, head, hflag, cidx, coff, dist, nv, opts, okeys, ochain);
    hipLaunchKernelGGL(c2_k_keep, dim3(c2_blocks(nsel)), dim3(256), 0, st, opts, ochain, coff, nsel, (flags & CX2_NO_DEDUPE) ? 0 : 1, keep);
    cx_scan_u32(ctx, keep, fidx, nsel, (uint32_t*)S->sums.p, scratch);
    uint32_t nfinal = 0;
    C2_HIP(ctx, hipMemcpyAsync(&nfinal, scratch, 4, hipMemcpyDeviceToHost, st));
    C2_HIP(ctx, hipStreamSynchronize(st));
    hipLaunchKernelGGL(c2_k_final_points, dim3(c2_blocks(nsel)), dim3(256), 0, st, opts, okeys, keep, fidx, nsel, (const double*)S->scal.p,
                       (double2*)S->fpts.p, (c2_u64*)S->fkeys.p);
    hipLaunchKernelGGL(c2_k_final_chains, dim3(c2_blocks(nchains)), dim3(256), 0, st, opts, okeys, keep, fidx, coff, clen, chead, cyc, nchains, nfinal,
                       (cx_chain2d*)S->chains.p);
    C2_HIP(ctx, hipGetLastError());
    C2_HIP(ctx, hipStreamSynchronize(st));
    S->counts = cx_counts2d{nfinal, nchains, nv, (uint32_t)nvalues};
    S->valid = true;
    if (out) *out = S->counts;
    return CX_OK;
}

extern "C" int cx_contour2d_download(cx_ctx* ctx, double* points_xy, int64_t* keys, cx_chain2d* chains) {
    if (!ctx) return CX_ERR_INVALID;
    ctx->err.clear();
    cx_state2* S = ctx->s2;
    if (!S || !S->valid) {
        ctx->err = "cx_contour2d_download: no contour extraction yet";
        return CX_ERR_STATE;
    }
    C2_HIP(ctx, hipSetDevice(ctx->device));
    const size_t np = S->counts.n_points, nc = S->counts.n_chains;
    if (points_xy && np) C2_HIP(ctx, hipMemcpyAsync(points_xy, S->fpts.p, np * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (keys && np) C2_HIP(ctx, hipMemcpyAsync(keys, S->fkeys.p, np * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (chains && nc) C2_HIP(ctx, hipMemcpyAsync(chains, S->chains.p, nc * sizeof(cx_chain2d), hipMemcpyDeviceToHost, ctx->stream));
    C2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CX_OK;
}
