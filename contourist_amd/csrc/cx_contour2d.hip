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
// pointer jumping (no atomics); the growth groups of the seeded search are a lock-free union-find over whole polylines.
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
    c2_buf grid, values, cnt, base, sums, pts, keys, succ, pred, parent, mark, rmark, rep, rank, cyc, jst[2], alist, len, hflag, cidx,
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
    c2_buf* all[] = {&S->grid, &S->values, &S->cnt, &S->base, &S->sums, &S->pts, &S->keys, &S->succ, &S->pred, &S->parent,
                     &S->mark, &S->rmark, &S->rep, &S->rank, &S->cyc, &S->jst[0], &S->jst[1], &S->alist, &S->len, &S->hflag, &S->cidx, &S->chead,
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
// the sorted isovalues are searched several times per crossing: kernels copy them to LDS first (up to C2_SV of them;
// longer lists are searched in global memory)
#define C2_SV 1024
#define C2_STAGE_VALUES(G)                                                              \
    __shared__ double c2_sv[C2_SV];                                                     \
    if ((G).nvalues <= C2_SV) {                                                         \
        for (uint32_t k__ = threadIdx.x; k__ < (G).nvalues; k__ += blockDim.x) c2_sv[k__] = (G).values[k__]; \
        __syncthreads();                                                                \
        (G).values = c2_sv;                                                             \
    }
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

// *tie is set when a sample on a searched axis edge equals one of the isovalues (the seed search then needs the
// per-sample pass c2_k_seed_ties)
__global__ void c2_k_count(c2_grid G, uint32_t* cnt, uint32_t* tie) {
    C2_STAGE_VALUES(G)
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= G.n * G.m) return;
    const uint32_t i = lin / G.m, j = lin - i * G.m;
    // the point and its three forward neighbours are requested together, the neighbours at clamped positions whether their edge
    // exists or not, and the three counts are stored at the end (one edge after the other -- a conditional load, a binary search, a
    // store -- every load waited for the store before it: four round trips per lattice point)
    const uint32_t i1 = min(i + 1u, G.n - 1u), j1 = min(j + 1u, G.m - 1u);
    const float r0 = G.A[(size_t)i * G.m + j], ra = G.A[(size_t)i1 * G.m + j], rb = G.A[(size_t)i * G.m + j1], rc = G.A[(size_t)i1 * G.m + j1];
    const double f0 = (r0 == r0) ? (double)r0 : (double)INFINITY;
    const double fn[3] = {(ra == ra) ? (double)ra : (double)INFINITY, (rb == rb) ? (double)rb : (double)INFINITY, (rc == rc) ? (double)rc : (double)INFINITY};
    const uint32_t u0 = c2_upper(G, f0);
    if (u0 > 0 && G.values[u0 - 1] == f0) *tie = 1u;
    uint32_t c[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        if (c2_edge_valid(G, i, j, d) && f0 != fn[d]) {
            const uint32_t u1 = c2_upper(G, fn[d]);
            c[d] = (u1 > u0) ? (u1 - u0) : (u0 - u1);
        }
    }
    cnt[3u * lin] = c[0]; cnt[3u * lin + 1u] = c[1]; cnt[3u * lin + 2u] = c[2];
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

// A block looks at C2_EPB lattice edges, queues the crossed ones (about one in ten on smooth fields) in LDS and then
// works through the queue with all of its lanes, one edge per lane
#define C2_EPT 8u
#define C2_EPB (256u * C2_EPT)
__global__ __launch_bounds__(256) void c2_k_emit(c2_grid G, const uint32_t* cnt, double2* pts, c2_u64* keys, uint32_t* succ, uint32_t* pred) {
    __shared__ uint32_t s_q[C2_EPB];
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0u;
    C2_STAGE_VALUES(G)
    __syncthreads();
    const uint32_t nedges = 3u * G.n * G.m, e0 = blockIdx.x * C2_EPB;
    for (uint32_t k = 0; k < C2_EPT; k++) {
        const uint32_t e = e0 + k * 256u + threadIdx.x;
        if (e < nedges && cnt[e] != 0u) s_q[atomicAdd(&s_n, 1u)] = e;   // the order in the queue does not matter: ids come from base[]
    }
    __syncthreads();
    const uint32_t nq = s_n;
    for (uint32_t q = threadIdx.x; q < nq; q += 256u) {
    const uint32_t eidx = s_q[q];
    const uint32_t lin = eidx / 3u;
    const int d = (int)(eidx - 3u * lin);
    const int i = (int)(lin / G.m), j = (int)(lin - (uint32_t)i * G.m);
    const double f0 = c2_f(G, i, j);
    const int bi = i + (d != 1), bj = j + (d != 0);
    const double f1 = c2_f(G, bi, bj);
    uint32_t s, e;
    c2_levels(G, f0, f1, s, e);
    const uint32_t id0 = G.base[eidx];
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
        keys[id] = ((c2_u64)eidx << 16) | (c2_u64)lvl;
        succ[id] = C2_NIL;
        pred[id] = C2_NIL;
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
// Growth groups (expand_contour_pairs :322-331): pairs that share an end point in the same role grow together.  The
// pairs of one chain already do; what is left is to unite CHAINS that pass the same lattice point in the same role
// (a saddle), which is rare: the union-find runs over chain representatives and most threads do no atomic at all.
// search != 0: the pair also is a seed of the exhaustive search (search_grid :198-212: a crossed axis edge from (i,j)
// with i < n-1, j < m-1) -- a strictly crossed edge lies in the growth group of both of its end points.
__global__ void c2_k_group(c2_grid G, const c2_u64* keys, const uint32_t* rep, uint32_t nv, int search, uint32_t* parent, uint32_t* mark) {
    C2_STAGE_VALUES(G)
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    int i, j, bi, bj;
    uint32_t lvl;
    c2_decode(G, keys[id], i, j, bi, bj, lvl);
    const bool first_low = c2_f(G, i, j) < G.values[lvl];
    const uint32_t ra = c2_first_around(G, i, j, first_low ? 0 : 1, lvl);
    const uint32_t rb = c2_first_around(G, bi, bj, first_low ? 1 : 0, lvl);
    const uint32_t me = rep[id];
    if (ra != C2_NIL && ra != id) {
        const uint32_t o = rep[ra];
        if (o != me) c2_union(parent, me, o);
    }
    if (rb != C2_NIL && rb != id) {
        const uint32_t o = rep[rb];
        if (o != me) c2_union(parent, me, o);
    }
    if (search && !((bi != i) && (bj != j)) && i + 1 < (int)G.n && j + 1 < (int)G.m) mark[me] = 1u;
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
// marks the chain of the first pair around a seed point.  role 0: the reference took the point as the low end of
// its seed (f <= z); a point equal to z is a high point here, so its pairs are looked up in that role
__device__ __forceinline__ void c2_mark_seed(const c2_grid& G, int i, int j, int role, uint32_t lvl, const uint32_t* rep, uint32_t* mark) {
    if (i < 0 || j < 0 || i >= (int)G.n || j >= (int)G.m) return;
    if (role == 0 && c2_f(G, i, j) == G.values[lvl]) role = 1;
    const uint32_t r = c2_first_around(G, i, j, role, lvl);
    if (r != C2_NIL) mark[rep[r]] = 1u;
}
// an axis edge whose lower sample EQUALS the isovalue is no pair here but a seed of the reference's search
// (f(low) <= z <= f(high)): its end points seed in the role they have under the build's rule (run only when
// c2_k_count saw such a sample)
__global__ void c2_k_seed_ties(c2_grid G, const uint32_t* rep, uint32_t* mark) {
    C2_STAGE_VALUES(G)
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= G.n * G.m) return;
    const int i = (int)(lin / G.m), j = (int)(lin - (uint32_t)i * G.m);
    if (i + 1 >= (int)G.n || j + 1 >= (int)G.m) return;
    const double f0 = c2_f(G, i, j);
    for (int d = 0; d < 2; d++) {
        const int bi = i + (d == 0), bj = j + (d == 1);
        const double f1 = c2_f(G, bi, bj);
        const bool fwd = f0 <= f1;
        const double lo = fmin(f0, f1);
        const uint32_t s = c2_lower(G, lo), e = c2_upper(G, lo);   // the levels equal to the lower sample
        for (uint32_t lvl = s; lvl < e; lvl++) {
            c2_mark_seed(G, fwd ? i : bi, fwd ? j : bj, 0, lvl, rep, mark);
            c2_mark_seed(G, fwd ? bi : i, fwd ? bj : j, 1, lvl, rep, mark);
        }
    }
}
// explicit seeds: (i, j, role, level index) -- the pairs around the point in that role (find_initial_contour_pairs :316-317)
__global__ void c2_k_seed_points(c2_grid G, const int32_t* seeds, uint32_t nseeds, const uint32_t* rep, uint32_t* mark) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseeds) return;
    const int32_t* q = seeds + 4 * (size_t)s;
    if (q[3] < 0 || q[3] >= (int)G.nvalues || (q[2] != 0 && q[2] != 1)) return;
    c2_mark_seed(G, q[0], q[1], q[2], (uint32_t)q[3], rep, mark);
}
// a seeded chain seeds its whole growth group
__global__ void c2_k_mark_roots(const uint32_t* rep, const uint32_t* mark, uint32_t nv, uint32_t* parent, uint32_t* rmark) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv || rep[id] != id || !mark[id]) return;
    rmark[c2_find(parent, id)] = 1u;
}

// ---- chains ----------------------------------------------------------------------------------------------
// Pointer jumping along the predecessor links, all chains at once, no atomics.  State of element i after k rounds:
//   x = ptr   the element reached after y steps back from i (an open chain's head points to itself)
//   y = hop   number of steps to ptr; the window of i is the y elements i, pred(i), ... in front of ptr
//   z = mn    smallest label in the window; label = 0 for the head of an open chain, id + 1 otherwise
//   w = off   steps from i back to the nearest element with that label
// Merging the window of i with the window of ptr doubles it.  Open chain: done when the window holds the head
// (mn == 0): off is the distance to the head.  Closed chain: when i and ptr report the same mn their windows overlap,
// i.e. together they cover the whole cycle: mn is the smallest id of the cycle and off the distance back to it, which
// is the rank of i when the cycle is opened in front of its smallest id.
// The first C2_WALK steps are walked one by one (the predecessors of a crossing are its neighbours in memory, and
// three rounds over all elements cost more than eight short dependent loads): the state starts with a window of up to
// C2_WALK elements instead of one.
#define C2_WALK 8
#define C2_FIN 0x80000000u
__global__ void c2_k_jump_init(const uint32_t* pred, uint32_t nv, uint4* st) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    uint32_t p = pred[id];
    if (p == C2_NIL) {   // the head of an open chain: label 0, final
        st[id] = make_uint4(id, C2_FIN, 0u, 0u);
        return;
    }
    uint32_t hop = 1, mn = id + 1u, off = 0;   // window {id}, ptr = p
    bool fin = false;
    for (int k = 1; k < C2_WALK; k++) {
        if (p == id) { fin = true; break; }            // a short cycle: the window is the whole cycle
        const uint32_t q = pred[p];
        if (q == C2_NIL) {                             // p is the head: take it in and stop there
            mn = 0u;
            off = hop;
            fin = true;
            break;
        }
        if (p + 1u < mn) { mn = p + 1u; off = hop; }   // p joins the window
        hop++;
        p = q;
    }
    st[id] = make_uint4(p, hop | (fin ? C2_FIN : 0u), mn, off);
}
// Bit 31 of hop marks an element whose state is final; it is copied from then on.  Most polylines are short and
// the rounds are paid by the longest: after C2_FULL_ROUNDS rounds over all elements the unfinished ones are listed
// (c2_k_jump_flags + scan + c2_k_jump_list) and the remaining rounds run over that list only.  Elements outside the
// list hold the same final state in both buffers (one full round is run after the round the list is taken from).
#define C2_FULL_ROUNDS 4
__device__ __forceinline__ void c2_jump_one(const uint4* in, uint4* out, uint32_t id, uint32_t* changed) {
    uint4 a = in[id];
    if (a.y & C2_FIN) {
        out[id] = a;
        return;
    }
    if (a.x == id) {   // a head, or a cycle whose length divides hop: final
        a.y |= C2_FIN;
        out[id] = a;
        return;
    }
    const uint4 b = in[a.x];
    uint4 r;
    r.x = b.x;
    r.y = a.y + (b.y & ~C2_FIN);
    if (b.z < a.z) {
        r.z = b.z;
        r.w = a.y + b.w;
    } else {
        r.z = a.z;
        r.w = a.w;
    }
    if (r.z == 0u || a.z == b.z) r.y |= C2_FIN; else *changed = 1u;
    out[id] = r;
}
__global__ void c2_k_jump(const uint4* in, uint4* out, uint32_t nv, uint32_t* changed) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < nv) c2_jump_one(in, out, id, changed);
}
__global__ void c2_k_jump_listed(const uint4* in, uint4* out, const uint32_t* list, uint32_t n, uint32_t* changed) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) c2_jump_one(in, out, list[k], changed);
}
__global__ void c2_k_jump_flags(const uint4* st, uint32_t nv, uint32_t* flag) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < nv) flag[id] = (st[id].y & C2_FIN) ? 0u : 1u;
}
__global__ void c2_k_jump_list(const uint32_t* flag, const uint32_t* pos, uint32_t nv, uint32_t* list) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < nv && flag[id]) list[pos[id]] = id;
}
// rep = first element of the chain (head, or smallest id of a cycle), rank = position in the chain, len[rep], cyc[rep]
__global__ void c2_k_chain_finish(const uint4* st, const uint32_t* succ, uint32_t nv, uint32_t* rep, uint32_t* rank, uint32_t* len, uint32_t* cyc,
                                  uint32_t* parent) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    const uint4 a = st[id];   // (hop is not needed any more)
    const uint32_t r = (a.z == 0u) ? a.x : a.z - 1u;
    rep[id] = r;
    rank[id] = a.w;
    if (r == id) cyc[id] = (a.z != 0u) ? 1u : 0u;
    parent[id] = id;
    const uint32_t s = succ[id];
    if (s == C2_NIL || s == r) len[r] = a.w + 1u;   // the last element of the chain
}
// hflag[id] = 1 for the head of a chain that is kept
__global__ void c2_k_head_flags(const uint32_t* rep, uint32_t* parent, const uint32_t* mark, int all, uint32_t nv, uint32_t* hflag) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    uint32_t f = 0;
    if (rep[id] == id) f = all ? 1u : (mark[c2_find(parent, id)] ? 1u : 0u);
    hflag[id] = f;
}
__global__ void c2_k_chain_table(const uint32_t* hflag, const uint32_t* cidx, const uint32_t* len, uint32_t nv, uint32_t* chead, uint32_t* clen) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv || !hflag[id]) return;
    const uint32_t c = cidx[id];
    chead[c] = id;
    clen[c] = len[id];
}
__global__ void c2_k_place(const double2* pts, const c2_u64* keys, const uint32_t* rep, const uint32_t* hflag, const uint32_t* cidx,
                           const uint32_t* coff, const uint32_t* rank, uint32_t nv, double2* opts, c2_u64* okeys, uint32_t* ochain) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nv) return;
    const uint32_t h = rep[id];
    if (!hflag[h]) return;
    const uint32_t c = cidx[h], pos = coff[c] + rank[id];
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
    uint32_t* scratch = (uint32_t*)((char*)S->scal.p + 32);   // [0] scan total, [1] changed flag, [2] tie flag, [4..5] 64-bit total
    C2_HIP(ctx, hipMemsetAsync(scratch, 0, 24, st));
    const size_t E = 3 * N;
    if ((rc = c2_reserve(ctx, S->cnt, E * 4)) || (rc = c2_reserve(ctx, S->base, (E + 1) * 4)) || (rc = c2_reserve(ctx, S->sums, (E / 1024 + 4) * 4)))
        return rc;
    c2_grid G{A, (uint32_t)n, (uint32_t)m, (const double*)S->values.p, (uint32_t)nvalues, (const uint32_t*)S->base.p};
    hipLaunchKernelGGL(c2_k_count, dim3(c2_blocks(N)), dim3(256), 0, st, G, (uint32_t*)S->cnt.p, scratch + 2);
    // (the scan also returns the total in 64 bits: a 32-bit total cannot show an overflow)
    cx_scan_u32(ctx, (const uint32_t*)S->cnt.p, (uint32_t*)S->base.p, (uint32_t)E, (uint32_t*)S->sums.p, scratch, (unsigned long long*)(scratch + 4));
    uint32_t head6[6] = {0, 0, 0, 0, 0, 0};
    C2_HIP(ctx, hipMemcpyAsync(head6, scratch, 24, hipMemcpyDeviceToHost, st));
    C2_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t nv = head6[0];
    const bool ties = head6[2] != 0;
    const unsigned long long total64 = ((unsigned long long)head6[5] << 32) | head6[4];
    if (total64 >= 0x20000000ull) {   // ids, and three times the longest chain, must stay below 2^31
        ctx->err = "cx_contour2d_extract: " + std::to_string(total64) + " crossings (more than 2^29): contour fewer levels per call";
        return CX_ERR_UNSUPPORTED;
    }
    S->counts = cx_counts2d{0, 0, nv, (uint32_t)nvalues};
    if (nv == 0) {
        S->valid = true;
        if (out) *out = S->counts;
        return CX_OK;
    }
    const size_t V = nv;
    c2_buf* u32s[] = {&S->succ, &S->pred, &S->parent, &S->mark, &S->rmark, &S->rep, &S->rank, &S->cyc, &S->len, &S->hflag, &S->cidx, &S->ochain, &S->keep,
                      &S->fidx};
    for (c2_buf* b : u32s)
        if ((rc = c2_reserve(ctx, *b, (V + 1) * 4))) return rc;
    if ((rc = c2_reserve(ctx, S->pts, V * 16)) || (rc = c2_reserve(ctx, S->keys, V * 8)) || (rc = c2_reserve(ctx, S->opts, V * 16)) ||
        (rc = c2_reserve(ctx, S->okeys, V * 8)) || (rc = c2_reserve(ctx, S->fpts, V * 16)) || (rc = c2_reserve(ctx, S->fkeys, V * 8)))
        return rc;
    if (S->sums.cap < (V / 1024 + 4) * 4 && (rc = c2_reserve(ctx, S->sums, (V / 1024 + 4) * 4))) return rc;
    double2* pts = (double2*)S->pts.p;
    c2_u64* keys = (c2_u64*)S->keys.p;
    uint32_t *succ = (uint32_t*)S->succ.p, *pred = (uint32_t*)S->pred.p, *parent = (uint32_t*)S->parent.p, *rep = (uint32_t*)S->rep.p;
    uint32_t *mark = (uint32_t*)S->mark.p, *rmark = (uint32_t*)S->rmark.p, *rank = (uint32_t*)S->rank.p, *cyc = (uint32_t*)S->cyc.p, *len = (uint32_t*)S->len.p;
    uint32_t *hflag = (uint32_t*)S->hflag.p, *cidx = (uint32_t*)S->cidx.p;
    const uint32_t gb = c2_blocks(V);
    hipLaunchKernelGGL(c2_k_emit, dim3((uint32_t)((E + C2_EPB - 1) / C2_EPB)), dim3(256), 0, st, G, (const uint32_t*)S->cnt.p, pts, keys, succ, pred);
    const int all = (flags & CX2_ALL_CHAINS) ? 1 : 0;
    // chains: first element, rank, length
    if ((rc = c2_reserve(ctx, S->jst[0], V * 16)) || (rc = c2_reserve(ctx, S->jst[1], V * 16))) return rc;
    int cur = 0;
    hipLaunchKernelGGL(c2_k_jump_init, dim3(gb), dim3(256), 0, st, pred, nv, (uint4*)S->jst[0].p);
    uint32_t nlist = 0;           // 0: rounds over all elements
    const uint32_t* list = nullptr;
    for (int round = 0;; round++) {
        if (round >= 48) {
            ctx->err = "cx_contour2d_extract: chain ranking did not converge";
            return CX_ERR_HIP;
        }
        C2_HIP(ctx, hipMemsetAsync(scratch + 1, 0, 4, st));
        if (nlist)
            hipLaunchKernelGGL(c2_k_jump_listed, dim3(c2_blocks(nlist)), dim3(256), 0, st, (const uint4*)S->jst[cur].p, (uint4*)S->jst[1 - cur].p, list,
                               nlist, scratch + 1);
        else
            hipLaunchKernelGGL(c2_k_jump, dim3(gb), dim3(256), 0, st, (const uint4*)S->jst[cur].p, (uint4*)S->jst[1 - cur].p, nv, scratch + 1);
        cur = 1 - cur;
        uint32_t changed = 0;
        C2_HIP(ctx, hipMemcpyAsync(&changed, scratch + 1, 4, hipMemcpyDeviceToHost, st));
        C2_HIP(ctx, hipStreamSynchronize(st));
        if (!changed) break;
        if (round == C2_FULL_ROUNDS && nv > (1u << 20)) {
            // the list of elements that were unfinished BEFORE this round (buffer 1 - cur): whatever finished earlier has
            // just been copied, i.e. is the same in both buffers, and can be left alone from now on
            if ((rc = c2_reserve(ctx, S->alist, V * 4))) return rc;
            hipLaunchKernelGGL(c2_k_jump_flags, dim3(gb), dim3(256), 0, st, (const uint4*)S->jst[1 - cur].p, nv, hflag);
            cx_scan_u32(ctx, hflag, cidx, nv, (uint32_t*)S->sums.p, scratch);
            hipLaunchKernelGGL(c2_k_jump_list, dim3(gb), dim3(256), 0, st, hflag, cidx, nv, (uint32_t*)S->alist.p);
            C2_HIP(ctx, hipMemcpyAsync(&nlist, scratch, 4, hipMemcpyDeviceToHost, st));
            C2_HIP(ctx, hipStreamSynchronize(st));
            list = (const uint32_t*)S->alist.p;
            if (nlist == 0) break;   // (cannot happen while `changed` is set; defensive)
        }
    }
    hipLaunchKernelGGL(c2_k_chain_finish, dim3(gb), dim3(256), 0, st, (const uint4*)S->jst[cur].p, succ, nv, rep, rank, len, cyc, parent);
    // growth groups of chains and their seeds
    if (!all) {
        const bool search = nseeds <= 0 || (flags & CX2_SEARCH_SEEDS);
        C2_HIP(ctx, hipMemsetAsync(mark, 0, V * 4, st));
        C2_HIP(ctx, hipMemsetAsync(rmark, 0, V * 4, st));
        hipLaunchKernelGGL(c2_k_group, dim3(gb), dim3(256), 0, st, G, keys, rep, nv, search ? 1 : 0, parent, mark);
        if (search && ties) hipLaunchKernelGGL(c2_k_seed_ties, dim3(c2_blocks(N)), dim3(256), 0, st, G, rep, mark);
        if (nseeds > 0) {
            if ((rc = c2_reserve(ctx, S->seeds, (size_t)nseeds * 16))) return rc;
            C2_HIP(ctx, hipMemcpyAsync(S->seeds.p, seeds, (size_t)nseeds * 16, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(c2_k_seed_points, dim3(c2_blocks((size_t)nseeds)), dim3(256), 0, st, G, (const int32_t*)S->seeds.p, (uint32_t)nseeds,
                               rep, mark);
        }
        hipLaunchKernelGGL(c2_k_mark_roots, dim3(gb), dim3(256), 0, st, rep, mark, nv, parent, rmark);
    }
    hipLaunchKernelGGL(c2_k_head_flags, dim3(gb), dim3(256), 0, st, rep, parent, rmark, all, nv, hflag);
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
    hipLaunchKernelGGL(c2_k_chain_table, dim3(gb), dim3(256), 0, st, hflag, cidx, len, nv, chead, clen);
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
    hipLaunchKernelGGL(c2_k_place, dim3(gb), dim3(256), 0, st, pts, keys, rep, hflag, cidx, coff, rank, nv, opts, okeys, ochain);
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
