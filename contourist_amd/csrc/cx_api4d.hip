// cx_api4d.hip -- C ABI of the 4-D (pentatope) march: grid binding, kernel sequencing, download.
#include <cmath>
#include <cstring>
#include <new>
#include <string>

#include "cx_ctx.h"

#include "cx_state4.h"

#define CX4_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return (e__ == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;                      \
        }                                                                                        \
    } while (0)

void cx_state4_free(cx_ctx* ctx) {
    cx_state4* S = ctx->s4;
    if (!S) return;
    cx_release(S->grid_owned, S->grid_owned_bytes); cx_release(S->items, S->items_cap); cx_release(S->info, S->info_cap);
    cx_release(S->verts, S->vcap); cx_release(S->vkeys, S->vkeys_cap); cx_release(S->cells, S->ccap); cx_release(S->tets, S->tcap);
    cx_release(S->hash_xyz, S->hash_cap); cx_release(S->signbits, S->signbits_cap); cx_release(S->tet_keep, S->keep_cap);
    cx_release(S->queue, S->qcap); cx_release(S->rounds, S->rounds_cap);
    delete S;
    ctx->s4 = nullptr;
}

static int state4(cx_ctx* ctx, cx_state4** out) {
    if (!ctx->s4) ctx->s4 = new (std::nothrow) cx_state4();
    if (!ctx->s4) return CX_ERR_NOMEM;
    *out = ctx->s4;
    return CX_OK;
}

static int set_dims4(cx_ctx* ctx, cx_state4* S, int64_t n0, int64_t n1, int64_t n2, int64_t n3) {
    if (n0 < 2 || n1 < 2 || n2 < 2 || n3 < 2) { ctx->err = "4-D grid needs at least 2 samples per axis"; return CX_ERR_INVALID; }
    const int64_t N = n0 * n1 * n2 * n3;
    if (N > (1LL << 28)) { ctx->err = "more than 2^28 samples in one 4-D grid: partition into slabs"; return CX_ERR_UNSUPPORTED; }
    S->n[0] = n0; S->n[1] = n1; S->n[2] = n2; S->n[3] = n3;
    S->extracted = false;
    S->keep_valid = false;
    return CX_OK;
}

extern "C" int cx_grid4d_upload(cx_ctx* ctx, const float* host, int64_t n0, int64_t n1, int64_t n2, int64_t n3) {
    if (!ctx || !host) return CX_ERR_INVALID;
    CX4_HIP(ctx, hipSetDevice(ctx->device));
    cx_state4* S;
    int rc = state4(ctx, &S);
    if (rc) return rc;
    if ((rc = set_dims4(ctx, S, n0, n1, n2, n3))) return rc;
    const size_t bytes = (size_t)(n0 * n1 * n2 * n3) * sizeof(float);
    {
        size_t have = S->grid_owned_bytes / sizeof(float);
        rc = cx_grow(ctx, S->grid_owned, have, bytes / sizeof(float));
        S->grid_owned_bytes = have * sizeof(float);
        if (rc) return rc;
    }
    CX4_HIP(ctx, hipMemcpyAsync(S->grid_owned, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    CX4_HIP(ctx, hipStreamSynchronize(ctx->stream));
    S->grid = S->grid_owned;
    return CX_OK;
}

extern "C" int cx_grid4d_adopt_device(cx_ctx* ctx, const void* device_ptr, int64_t n0, int64_t n1, int64_t n2, int64_t n3) {
    if (!ctx || !device_ptr) return CX_ERR_INVALID;
    CX4_HIP(ctx, hipSetDevice(ctx->device));
    cx_state4* S;
    int rc = state4(ctx, &S);
    if (rc) return rc;
    if ((rc = set_dims4(ctx, S, n0, n1, n2, n3))) return rc;
    S->grid = (const float*)device_ptr;
    return CX_OK;
}

static int reserve4(cx_ctx* ctx, cx_state4* S, int64_t nc, int64_t nv, int64_t nt, int64_t nq) {
    if (nq > 0xFFFFFFF0LL || nc > 0xFFFFFFF0LL || nv > 0xFFFFFFF0LL || nt > 0x7FFFFFF0LL) { ctx->err = "capacity beyond 32-bit indices"; return CX_ERR_UNSUPPORTED; }
    int rc;
    if ((rc = cx_grow(ctx, S->queue, S->qcap, (size_t)nq))) return rc;
    if ((rc = cx_grow(ctx, S->rounds, S->rounds_cap, (size_t)S->qcap / 64 + 8))) return rc;
    if ((rc = cx_grow(ctx, S->info, S->info_cap, (size_t)S->qcap + 64u))) return rc;
    if ((rc = cx_grow(ctx, S->cells, S->ccap, (size_t)nc))) return rc;
    if ((rc = cx_grow(ctx, S->verts, S->vcap, (size_t)nv))) return rc;
    if ((rc = cx_grow(ctx, S->vkeys, S->vkeys_cap, (size_t)S->vcap))) return rc;
    {
        size_t t4 = (size_t)S->tcap * 4u;
        if ((rc = cx_grow(ctx, S->tets, t4, (size_t)nt * 4u))) { S->tcap = 0; return rc; }
        S->tcap = (uint32_t)(t4 / 4u);
    }
    return CX_OK;
}

// one attempt with the buffers as they are: parameters, the four kernels and the copy of the counters into pinned host memory, all
// enqueued on the context's stream -- nothing waited for
static int enqueue4(cx_ctx* ctx, cx_state4* S, double value, uint32_t flags) {
    int rc;
    const int64_t N = S->n[0] * S->n[1] * S->n[2] * S->n[3];
    cx_params4 P;
    memset(&P, 0, sizeof(P));
    P.grid = S->grid;
    P.n0 = (uint32_t)S->n[0]; P.n1 = (uint32_t)S->n[1]; P.n2 = (uint32_t)S->n[2]; P.n3 = (uint32_t)S->n[3];
    P.nsamples = (uint32_t)N;
    P.div3 = cx_fdiv_make(P.n1 * P.n2 * P.n3);
    P.div2 = cx_fdiv_make(P.n2 * P.n3);
    P.div1 = cx_fdiv_make(P.n3);
    float t = (float)value;
    if ((double)t < value) t = std::nextafterf(t, INFINITY);
    P.vcmp = t;
    P.near_abs = std::nextafterf((float)(2.2e-5 * std::fabs(value) + 4e-8), INFINITY);
    P.vhi = (float)value;
    P.vlo = (float)(value - (double)P.vhi);
    P.value = value;
    P.tol_value = 1e-8 + 1e-5 * std::fabs(value);
    P.flags = flags;
    for (int d = 0; d < 4; d++) P.org[d] = (uint32_t)S->origin[d];
    P.info = S->info; P.verts = S->verts; P.vkeys = S->vkeys; P.cells = S->cells; P.tets = S->tets;
    P.vcap = S->vcap; P.ccap = S->ccap; P.tcap = S->tcap;
    P.queue = S->queue; P.qcap = S->qcap; P.rounds = S->rounds;
    P.counters = ctx->counters + CX_CNT_WORDS;   // the 4-D march's own block (cx_ctx_create)
    P.counters_tb = reinterpret_cast<unsigned long long*>(ctx->counters + 1024);
    P.lut = cx_pent_lut_device();
    if (!P.lut) { ctx->err = "pentatope table symbol not found"; return CX_ERR_HIP; }
    if (flags & CX_DIAG_CPYTHON310) {
        const int64_t key[7] = {S->n[0], S->n[1], S->n[2], S->origin[0], S->origin[1], S->origin[2], 1};
        if (memcmp(key, S->hash_key, sizeof(key)) != 0) {
            const size_t need = (size_t)(S->n[0] * S->n[1] * S->n[2]);
            if ((rc = cx_grow(ctx, S->hash_xyz, S->hash_cap, need))) return rc;
            cx_launch_hash_xyz(S->hash_xyz, P.n0, P.n1, P.n2, P.org, ctx->stream);
            memcpy(S->hash_key, key, sizeof(key));
        }
    }
    P.hash_xyz = S->hash_xyz;
    P.nw3 = (P.n3 + 31u) / 32u;
    P.nrows = P.n0 * P.n1 * P.n2;
    P.div_w = cx_fdiv_make(P.nw3);
    P.div_r2 = cx_fdiv_make(P.n1 * P.n2);
    P.div_r1 = cx_fdiv_make(P.n2);
    {
        const size_t need = (size_t)P.nrows * P.nw3 + 64u;
        if ((rc = cx_grow(ctx, S->signbits, S->signbits_cap, need))) return rc;
        P.signbits = S->signbits;
        if ((rc = cx_grow(ctx, S->items, S->items_cap, need))) return rc;
        P.items = S->items;
    }
    CX4_HIP(ctx, hipMemsetAsync(ctx->counters + CX_CNT_WORDS, 0, CX_CNT_WORDS * sizeof(uint32_t), ctx->stream));
    cx_launch_signbits4d(P, ctx->stream);
    cx_launch_classify4d(P, ctx->stream);
    cx_launch_emit_tets(P, ctx->stream);
    CX4_HIP(ctx, hipGetLastError());
    CX4_HIP(ctx, hipMemcpyAsync(ctx->counters_host + CX_CNT_WORDS, ctx->counters + CX_CNT_WORDS, CX_CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    return CX_OK;
}
// the counters of the attempt that was enqueued last: 0 = the extraction stands, 1 = a buffer was too small and has been grown (run
// the attempt again), < 0: error
static int settle4(cx_ctx* ctx, cx_state4* S, double value, cx_counts* out) {
    int rc;
    CX4_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cx_counts c;
    c.n_cells = ctx->counters_host[CX_CNT_WORDS + CX_CNT_CELLS];
    c.n_vertices = ctx->counters_host[CX_CNT_WORDS + CX_CNT_VERTS];
    c.n_triangles = ctx->counters_host[CX_CNT_WORDS + CX_CNT_TRIS];   // tetrahedra
    c.n_border_voxels = ctx->counters_host[CX_CNT_WORDS + CX_CNT_BORDER];
    S->counts = c;
    if (out) *out = c;
    const uint32_t nq = ctx->counters_host[CX_CNT_WORDS + CX4_CNT_QUEUE];
    if (nq > S->qcap) {   // nothing was classified: only the queue length is known
        if ((rc = reserve4(ctx, S, 0, 0, 0, (int64_t)nq + nq / 20 + 1024))) return rc;
        return 1;
    }
    if (c.n_cells <= S->ccap && c.n_vertices <= S->vcap && c.n_triangles <= S->tcap) {
        S->extracted = true;
        S->post_valid = false;
        S->keep_valid = false;
        S->value = value;
        return 0;
    }
    if ((rc = reserve4(ctx, S, c.n_cells + c.n_cells / 20 + 1024, c.n_vertices + c.n_vertices / 20 + 1024,
                       c.n_triangles + c.n_triangles / 20 + 1024, 0)))
        return rc;
    return 1;
}
static int begin4(cx_ctx* ctx, cx_state4** Sout, double value) {
    if (!ctx) return CX_ERR_INVALID;
    CX4_HIP(ctx, hipSetDevice(ctx->device));
    cx_state4* S;
    int rc = state4(ctx, &S);
    if (rc) return rc;
    if (!S->grid) { ctx->err = "no 4-D grid: call cx_grid4d_upload or cx_grid4d_adopt_device first"; return CX_ERR_STATE; }
    if (!(value == value)) { ctx->err = "isovalue is NaN"; return CX_ERR_INVALID; }
    const int64_t N = S->n[0] * S->n[1] * S->n[2] * S->n[3];
    if (!S->cells) {
        if ((rc = reserve4(ctx, S, N / 8 + 4096, N / 2 + 4096, 4 * N + 4096, N / 8 + 4096))) return rc;
    }
    for (int d = 0; d < 4; d++) S->origin[d] = ctx->origin4[d];
    S->extracted = false;
    *Sout = S;
    return CX_OK;
}

extern "C" int cx_extract4d(cx_ctx* ctx, double value, uint32_t flags, cx_counts* out) {
    cx_state4* S;
    int rc = begin4(ctx, &S, value);
    if (rc) return rc;
    S->pending = false;
    for (int attempt = 0; attempt < 4; attempt++) {
        if ((rc = enqueue4(ctx, S, value, flags))) return rc;
        rc = settle4(ctx, S, value, out);
        if (rc == 0) return CX_OK;
        if (rc < 0) return rc;
    }
    ctx->err = "4-D output buffers still too small after growing";
    return CX_ERR_CAPACITY;
}
// The same march, enqueued and not waited for: a caller with several contexts (one volume after the other, or the same one at several
// isovalues) keeps one extraction in flight per context, as cx_extract3d_async / cx_counts_get do for the 3-D path; cx_counts4d_get waits,
// validates the counters and -- if a buffer was too small -- grows it and runs the march again, synchronously.
extern "C" int cx_extract4d_async(cx_ctx* ctx, double value, uint32_t flags) {
    cx_state4* S;
    int rc = begin4(ctx, &S, value);
    if (rc) return rc;
    S->pending = false;
    if ((rc = enqueue4(ctx, S, value, flags))) return rc;
    S->pending = true; S->pending_value = value; S->pending_flags = flags;
    return CX_OK;
}
extern "C" int cx_counts4d_get(cx_ctx* ctx, cx_counts* out) {
    if (!ctx) return CX_ERR_INVALID;
    cx_state4* S = ctx->s4;
    if (!S || !S->pending) { if (ctx) ctx->err = "cx_counts4d_get: no cx_extract4d_async in flight"; return CX_ERR_STATE; }
    CX4_HIP(ctx, hipSetDevice(ctx->device));
    S->pending = false;
    const int rc = settle4(ctx, S, S->pending_value, out);
    if (rc == 0) return CX_OK;
    if (rc < 0) return rc;
    return cx_extract4d(ctx, S->pending_value, S->pending_flags, out);     // (buffers grown by settle4)
}

extern "C" int cx_level0_4d_download(cx_ctx* ctx, float* verts_xyzt, uint32_t* edge_ids, int32_t* tets) {
    if (!ctx) return CX_ERR_INVALID;
    cx_state4* S = ctx->s4;
    if (!S || !S->extracted) { ctx->err = "no valid 4-D extraction"; return CX_ERR_STATE; }
    CX4_HIP(ctx, hipSetDevice(ctx->device));
    if (verts_xyzt && S->counts.n_vertices)
        CX4_HIP(ctx, hipMemcpyAsync(verts_xyzt, S->verts, (size_t)S->counts.n_vertices * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (edge_ids && S->counts.n_vertices)
        CX4_HIP(ctx, hipMemcpyAsync(edge_ids, S->vkeys, (size_t)S->counts.n_vertices * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (tets && S->counts.n_triangles)
        CX4_HIP(ctx, hipMemcpyAsync(tets, S->tets, (size_t)S->counts.n_triangles * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    CX4_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CX_OK;
}

extern "C" int cx_set_origin4d(cx_ctx* ctx, int64_t o0, int64_t o1, int64_t o2, int64_t o3) {
    // negative: the array starts before the reference's grid (a rim of extra samples)
    if (!ctx || o0 < -1024 || o1 < -1024 || o2 < -1024 || o3 < -1024) return CX_ERR_INVALID;
    ctx->origin4[0] = o0; ctx->origin4[1] = o1; ctx->origin4[2] = o2; ctx->origin4[3] = o3;
    return CX_OK;
}
