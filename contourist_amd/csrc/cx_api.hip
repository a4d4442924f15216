// cx_api.hip -- C ABI (include/contourist_hip.h) of the gfx950 isosurface extractor: context,
// device memory, kernel sequencing.  No torch types, no global state.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "cx_ctx.h"

#define CX_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return (e__ == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;                      \
        }                                                                                        \
    } while (0)

static int fail(cx_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->err = msg;
    return code;
}

extern "C" const char* cx_version(void) { return "contourist_hip gfx950 " __DATE__; }

extern "C" int cx_ctx_create(int device_id, cx_ctx** out) {
    if (!out) return CX_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return CX_ERR_HIP;
    cx_ctx* ctx = new (std::nothrow) cx_ctx();
    if (!ctx) return CX_ERR_NOMEM;
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return CX_ERR_HIP;
    }
    ctx->own_stream = ctx->stream;
    // two counter blocks: [0, CX_CNT_WORDS) the 3-D march, [CX_CNT_WORDS, 2 CX_CNT_WORDS) the 4-D march -- a 4-D extraction on the
    // same context must not disturb what a later relaunch of the 3-D vertex stage (cx_ensure_cell_records) reads on the device
    // (+ a second page: the 4-D cells kernel's (tetrahedra | border voxels) word sits 4 KB away from its (cells | vertices) word --
    // two same-line atomics per workgroup executed one after the other at the memory side, 38 us of that kernel's 63)
    if (hipMalloc(&ctx->counters, 2048 * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc(&ctx->counters_host, 2 * CX_CNT_WORDS * sizeof(uint32_t)) != hipSuccess) {
        cx_ctx_destroy(ctx);
        return CX_ERR_NOMEM;
    }
    *out = ctx;
    return CX_OK;
}

static void free_outputs(cx_ctx* ctx) {
    cx_release(ctx->verts, ctx->vcap);
    cx_release(ctx->verts_xyz, ctx->verts_xyz_cap);
    cx_release(ctx->cells, ctx->ccap);
    cx_release(ctx->tris, ctx->tcap);
}

extern "C" int cx_ctx_destroy(cx_ctx* ctx) {
    if (!ctx) return CX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    cx_rccl_comm_free(ctx);
    cx_levels_free(ctx);
    cx_xfer_free(ctx);
    cx_post_free(ctx);
    cx_state4_free(ctx);
    cx_state2_free(ctx);
    free_outputs(ctx);
    cx_release(ctx->grid_owned, ctx->grid_owned_bytes);
    cx_release(ctx->grid64, ctx->grid64_cap);
    cx_release(ctx->celltab, ctx->tables_for);
    cx_release(ctx->queue, ctx->queue_cap);
    cx_release(ctx->wsum, ctx->wsum_cap);
    cx_release(ctx->wbase, ctx->wbase_cap);
    cx_release(ctx->tri_keep, ctx->keep_cap);
    for (int k = 0; k < 8; k++) cx_release(ctx->seed_buf[k], ctx->seed_cap[k]);
    cx_release(ctx->brec, ctx->brec_cap);
    cx_release(ctx->flat, ctx->flat_cap);
    cx_release(ctx->hash_xy, ctx->hash_xy_cap);
    cx_release(ctx->qa, ctx->qa_cap);
    cx_release(ctx->info, ctx->info_cap);
    cx_release(ctx->info64, ctx->info64_cap);
    cx_release(ctx->tq, ctx->tq_cap);
    cx_release(ctx->chunksum, ctx->chunksum_cap);
    cx_release(ctx->rstart, ctx->rstart_cap);
    cx_release(ctx->kstart, ctx->kstart_cap);
    cx_release(ctx->hbytes, ctx->hbytes_cap);
    cx_release(ctx->fj, ctx->fj_cap);
    cx_release(ctx->fk, ctx->fk_cap);
    cx_release(ctx->bnd, ctx->bnd_cap);
    cx_release(ctx->bndn, ctx->bndn_cap);
    cx_release(ctx->torder, ctx->torder_cap);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->counters_host) (void)hipHostFree(ctx->counters_host);
    for (auto& ev : ctx->events)
        for (int n = 0; n < 5; n++)
            if (ev.e[n]) (void)hipEventDestroy(ev.e[n]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return CX_OK;
}

extern "C" const char* cx_last_error(cx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int cx_set_stream(cx_ctx* ctx, void* hip_stream) {
    if (!ctx) return CX_ERR_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return CX_OK;
}

extern "C" int cx_synchronize(cx_ctx* ctx) {
    if (!ctx) return CX_ERR_INVALID;
    CX_HIP(ctx, hipSetDevice(ctx->device));
    CX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CX_OK;
}

static int set_grid_dims(cx_ctx* ctx, int64_t n0, int64_t n1, int64_t n2) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return fail(ctx, CX_ERR_INVALID, "grid must have at least 2 samples per axis");
    const int64_t N = n0 * n1 * n2;
    if (N > (1LL << 29)) return fail(ctx, CX_ERR_UNSUPPORTED, "more than 2^29 samples in one grid: partition into slabs");
    ctx->n0 = n0; ctx->n1 = n1; ctx->n2 = n2;
    ctx->grid64_valid = false;   // the float64 originals belonged to the previous grid
    cx_levels_invalidate(ctx);
    ctx->extracted = false;
    ctx->post_valid = false;
    return CX_OK;
}

extern "C" int cx_grid_upload(cx_ctx* ctx, const float* host, int64_t n0, int64_t n1, int64_t n2) {
    if (!ctx || !host) return CX_ERR_INVALID;
    CX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = set_grid_dims(ctx, n0, n1, n2);
    if (rc) return rc;
    const size_t bytes = (size_t)(n0 * n1 * n2) * sizeof(float);
    {
        size_t have = ctx->grid_owned_bytes / sizeof(float);
        rc = cx_grow(ctx, ctx->grid_owned, have, (size_t)(n0 * n1 * n2));
        ctx->grid_owned_bytes = have * sizeof(float);
        if (rc) return rc;
    }
    CX_HIP(ctx, hipMemcpyAsync(ctx->grid_owned, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    CX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->grid = ctx->grid_owned;
    return CX_OK;
}

extern "C" int cx_grid_adopt_device(cx_ctx* ctx, const void* device_ptr, int64_t n0, int64_t n1, int64_t n2) {
    if (!ctx || !device_ptr) return CX_ERR_INVALID;
    CX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = set_grid_dims(ctx, n0, n1, n2);
    if (rc) return rc;
    ctx->grid = (const float*)device_ptr;
    return CX_OK;
}

// The reference evaluates a callable field in float64 and interpolates its crossings on those values
// (tetrahedral.py:471-487); the march runs on their fp32 roundings.  With the float64 originals bound here Level 1
// (cx_postprocess3d, cx_level0_points_f64) interpolates on them, so a crossing next to a weld-bucket boundary
// (surface_geometry.py:30-48) lands on the reference's side of it.  NULL drops them.  8 bytes per sample.
extern "C" int cx_grid_shadow_f64(cx_ctx* ctx, const double* host, int64_t n0, int64_t n1, int64_t n2) {
    if (!ctx) return CX_ERR_INVALID;
    ctx->grid64_valid = false;
    ctx->post_valid = false;
    if (!host) return CX_OK;
    if (!ctx->grid || n0 != ctx->n0 || n1 != ctx->n1 || n2 != ctx->n2)
        return fail(ctx, CX_ERR_STATE, "cx_grid_shadow_f64: bind the fp32 grid of the same dimensions first");
    CX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)(n0 * n1 * n2);
    int rc = cx_grow(ctx, ctx->grid64, ctx->grid64_cap, N);
    if (rc) return rc;
    CX_HIP(ctx, hipMemcpyAsync(ctx->grid64, host, N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->grid64_valid = true;
    return CX_OK;
}

extern "C" int cx_set_origin(cx_ctx* ctx, int64_t o0, int64_t o1, int64_t o2) {
    // (negative: an array with a rim of samples around the reference's grid; the hash order of the diagonals and
    // cx_level0_points_f64 use lattice point + origin)
    const int64_t lim = 0x3FFFFFFFLL;
    if (!ctx || o0 < -lim || o1 < -lim || o2 < -lim || o0 > lim || o1 > lim || o2 > lim) return CX_ERR_INVALID;
    ctx->origin[0] = o0; ctx->origin[1] = o1; ctx->origin[2] = o2;
    return CX_OK;
}

extern "C" int cx_set_reference_corner(cx_ctx* ctx, int64_t c0, int64_t c1, int64_t c2) {
    if (!ctx || c0 < 0 || c1 < 0 || c2 < 0) return CX_ERR_INVALID;
    ctx->corner_ref[0] = c0; ctx->corner_ref[1] = c1; ctx->corner_ref[2] = c2;
    ctx->post_valid = false;
    return CX_OK;
}

extern "C" int cx_reserve(cx_ctx* ctx, int64_t max_cells, int64_t max_vertices, int64_t max_triangles) {
    if (!ctx || max_cells < 0 || max_vertices < 0 || max_triangles < 0) return CX_ERR_INVALID;
    if (max_cells > 0xFFFFFFF0LL || max_vertices > 0xFFFFFFF0LL || max_triangles > 0x7FFFFFF0LL)
        return fail(ctx, CX_ERR_UNSUPPORTED, "capacity beyond 32-bit indices");
    CX_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = cx_grow(ctx, ctx->cells, ctx->ccap, (size_t)max_cells))) return rc;
    if ((rc = cx_grow(ctx, ctx->verts, ctx->vcap, (size_t)max_vertices))) return rc;
    {
        size_t t3 = (size_t)ctx->tcap * 3u;
        if ((rc = cx_grow(ctx, ctx->tris, t3, (size_t)max_triangles * 3u))) { ctx->tcap = 0; return rc; }
        ctx->tcap = (uint32_t)(t3 / 3u);
    }
    ctx->extracted = false;
    ctx->post_valid = false;
    return CX_OK;
}

static float fp32_threshold(double value) {
    // smallest fp32 t with t >= value, so that for every fp32 sample f:  (double)f < value  <=>  f < t
    float t = (float)value;
    if ((double)t < value) t = std::nextafterf(t, INFINITY);
    return t;
}

// CPython tuple-hash prefixes per (i,j) column (CX_DIAG_CPYTHON310), rebuilt when the shape or the origin changed
int cx_ensure_hash_xy(cx_ctx* ctx, uint32_t flags) {
    if (!(flags & CX_DIAG_CPYTHON310)) return CX_OK;
    if (ctx->hash_xy_n0 == ctx->n0 && ctx->hash_xy_n1 == ctx->n1 && ctx->hash_xy_o0 == ctx->origin[0] && ctx->hash_xy_o1 == ctx->origin[1])
        return CX_OK;
    const size_t need = (size_t)(ctx->n0 * ctx->n1);
    {
        const int rc = cx_grow(ctx, ctx->hash_xy, ctx->hash_xy_cap, need);
        if (rc) return rc;
    }
    cx_launch_hash_xy(ctx->hash_xy, (uint32_t)ctx->n0, (uint32_t)ctx->n1, (uint32_t)ctx->origin[0], (uint32_t)ctx->origin[1], ctx->stream);
    ctx->hash_xy_n0 = ctx->n0; ctx->hash_xy_n1 = ctx->n1;
    ctx->hash_xy_o0 = ctx->origin[0]; ctx->hash_xy_o1 = ctx->origin[1];
    ctx->hbytes_valid = false;
    return CX_OK;
}

// the value-dependent fields of one extraction
void cx_fill_value_params(cx_params& P, double value) {
    P.vcmp = fp32_threshold(value);
    // The stream kernel takes "f < vcmp" from the SIGN BIT of f - vcmp.  With vcmp == +0.0 a sample of -0.0 gives -0.0 - 0.0 = -0.0:
    // sign bit set, "below the isovalue" -- where the reference's (and every other kernel's) comparison -0.0 < 0.0 is false; a volume
    // holding negative zeros (np.round of small negative numbers) then got inconsistent meshes at isovalue 0 (found by tools/fuzz_gpu.py,
    // round 4: present since round 1).  As -0.0 the threshold compares the same and subtracts right: (+-0.0) - (-0.0) = +0.0.
    if (P.vcmp == 0.0f) P.vcmp = -0.0f;
    // |f-v| <= 1e-8 + 1e-5*max(|f|,|v|)  =>  |f - vcmp| <= 2.2e-5*|v| + 4e-8  (vcmp within 1 ulp of v)
    P.near_abs = std::nextafterf((float)(2.2e-5 * std::fabs(value) + 4e-8), INFINITY);
    P.vhi = (float)value;
    P.vlo = (float)(value - (double)P.vhi);
    P.value = value;
    P.tol_value = 1e-8 + 1e-5 * std::fabs(value);
}

static int enqueue_extract(cx_ctx* ctx, double value, uint32_t flags) {
    if (!ctx->grid) return fail(ctx, CX_ERR_STATE, "no grid: call cx_grid_upload or cx_grid_adopt_device first");
    if (!(value == value)) return fail(ctx, CX_ERR_INVALID, "isovalue is NaN");
    // the public flags are the documented ones; the ablation bits (CX_DBG_*) give partial meshes and are only honoured
    // in a process started with CX_DEBUG=1 (tools/)
    if ((flags & ~(uint32_t)(CX_DIAG_CPYTHON310 | CX_KERNEL_GENERIC | CX_KERNEL_STAGED | CX_KERNEL_FUSED | CX_KERNEL_TILED)) != 0u && !cx_debug_enabled())
        return fail(ctx, CX_ERR_INVALID, "unknown flag bits in cx_extract3d");
    const int64_t N = ctx->n0 * ctx->n1 * ctx->n2;
    if (!ctx->cells || !ctx->verts || !ctx->tris) {
        int rc = cx_reserve(ctx, N / 16 + 4096, N / 8 + 4096, N / 4 + 4096);
        if (rc) return rc;
    }
    cx_params P;
    memset(&P, 0, sizeof(P));
    P.grid = ctx->grid;
    P.n0 = (uint32_t)ctx->n0; P.n1 = (uint32_t)ctx->n1; P.n2 = (uint32_t)ctx->n2;
    P.nsamples = (uint32_t)N;
    P.div_plane = cx_fdiv_make(P.n1 * P.n2);
    P.div_row = cx_fdiv_make(P.n2);
    cx_fill_value_params(P, value);
    P.flags = flags;
    const bool staged = !(flags & CX_KERNEL_GENERIC) && cx_fast_classify_supported_dims(ctx->n2, ctx->grid);
    // fused emit kernel (no per-cell table, no cell records) only on request: measured slower than the staged kernels
    // (DESIGN.md section 4); an extraction that meets the tolerance path is sent through the staged kernels by cx_counts_get
    const bool fused = staged && (flags & CX_KERNEL_FUSED) && !(flags & CX_KERNEL_STAGED);
    // tile emit (cx_tile3d.h): vertex records and triangles tile by tile, the hand-over between them in LDS
    // On request (CX_KERNEL_TILED; CX_DEBUG=1 CX_TILED=1 makes it the default of a tools/ process): measured 4 % faster than the staged
    // kernels with one extraction in flight, equal with two, 0.35 GB less traffic -- and a smooth sheet lying flat in a half tile
    // (a quarter of its cells active) is beyond its LDS words, which sends the whole extraction through the staged kernels
    const bool tiled = staged && !fused && !(flags & CX_KERNEL_STAGED) && ((flags & CX_KERNEL_TILED) || cx_debug_knob("CX_TILED", 0u));
    if (!staged) {
        // the per-cell table of the generic emit path (one 8-byte entry per sample): only when that path runs
        const int rc = cx_grow(ctx, ctx->celltab, ctx->tables_for, (size_t)N + 64u);
        if (rc) return rc;
    }
    P.celltab = ctx->celltab;
    P.fused = fused ? 1u : 0u;
    P.verts = ctx->verts; P.cells = ctx->cells; P.tris = ctx->tris;
    P.vcap = ctx->vcap; P.ccap = ctx->ccap; P.tcap = ctx->tcap;
    P.counters = ctx->counters;
    P.stamps = ctx->stamps;
    ctx->last = P;
    P.org0 = (uint32_t)ctx->origin[0]; P.org1 = (uint32_t)ctx->origin[1]; P.org2 = (uint32_t)ctx->origin[2];
    ctx->last = P;
    {
        const int rch = cx_ensure_hash_xy(ctx, flags);
        if (rch) return rch;
    }
    P.hbytes = nullptr;
    if ((flags & CX_DIAG_CPYTHON310) && (fused || cx_debug_knob("CX_HBYTES", 0u))) {   // staged kernels: measured slower than the hash arithmetic (byte gathers from a table of one byte per sample)
        // one byte per lattice point: its slot (and alternative slot) in CPython's 8-slot set, built once per shape / origin
        if (!ctx->hbytes_valid || ctx->hbytes_n2 != ctx->n2 || ctx->hbytes_o2 != ctx->origin[2]) {
            {
                const int rc = cx_grow(ctx, ctx->hbytes, ctx->hbytes_cap, (size_t)N + 64u);
                if (rc) return rc;
            }
            cx_launch_hash_bytes(ctx->hbytes, ctx->hash_xy, P.n0, P.n1, P.n2, P.org2, ctx->stream);
            ctx->hbytes_valid = true; ctx->hbytes_n2 = ctx->n2; ctx->hbytes_o2 = ctx->origin[2];
        }
        P.hbytes = ctx->hbytes;
    }
    cx_task T;
    memset(&T, 0, sizeof(T));
    if (staged) {
        T = cx_fast_task(P.n0, P.n1, P.n2);
        const size_t nw = (size_t)T.nblocks * 4u, need = nw * T.wcap;
        int rc;
        if ((rc = cx_grow(ctx, ctx->queue, ctx->queue_cap, need))) return rc;
        if ((rc = cx_grow(ctx, ctx->wsum, ctx->wsum_cap, nw))) return rc;
        if ((rc = cx_grow(ctx, ctx->wbase, ctx->wbase_cap, nw))) return rc;
        if ((rc = cx_grow(ctx, ctx->brec, ctx->brec_cap, nw * T.bcap))) return rc;
        // batches: at most one short batch per streaming wave plus one per CX_BATCH_MIN (>= 128) queued cells; queued
        // cells = cell records + array-boundary cells without vertices.  Sized from the cell capacity, so a
        // surface that fits the cell capacity fits here (cx_counts_get reports CX_ERR_CAPACITY otherwise).
        const size_t boundary = (size_t)(ctx->n0 * ctx->n1 + ctx->n0 * ctx->n2 + ctx->n1 * ctx->n2);
        const size_t nflat = nw + (size_t)ctx->ccap / 64u + boundary / 128u + 4096u;
        if ((rc = cx_grow(ctx, ctx->flat, ctx->flat_cap, nflat))) return rc;
        if ((rc = cx_grow(ctx, ctx->qa, ctx->qa_cap, nw * CX_SWP * 64u + 64u))) return rc;
        const size_t nchunk_words = ((nw + 255u) / 256u) * 8u;
        if ((rc = cx_grow(ctx, ctx->chunksum, ctx->chunksum_cap, nchunk_words))) return rc;
        CX_HIP(ctx, hipMemsetAsync(ctx->chunksum, 0, nchunk_words * sizeof(uint32_t), ctx->stream));
        if (!fused && (rc = cx_grow(ctx, ctx->info64, ctx->info64_cap, need))) return rc;   // staged kernels: (first vertex, crossing mask) per queue entry
        if (fused && (rc = cx_grow(ctx, ctx->info, ctx->info_cap, need))) return rc;        // fused kernel: one word per queue entry
        P.queue = ctx->queue; P.wsum = ctx->wsum; P.wbase = ctx->wbase; P.brec = ctx->brec;
        P.flat = ctx->flat; P.fcap = (uint32_t)nflat;
        P.qa = ctx->qa; P.info = ctx->info; P.info64 = ctx->info64; P.chunksum = ctx->chunksum;
        P.div_ci = cx_fdiv_make(T.ci);
        P.qlimit = T.wcap;
        // The stream of interpolation fractions between the stream kernel and the vertex stage (round 4): the stream
        // kernel computes the fractions where the samples are, the vertex stage reads them in order instead of gathering samples from
        // half of the grid's cache lines.  Bit-identical meshes and 185 MB less HBM traffic per 512^3 extraction -- but MEASURED SLOWER:
        // the vertex stage drops from 0.105 to 0.072 ms while the stream kernel, whose every wave is on the critical path of its
        // own loads, goes from 0.131 to 0.176 ms (~125 instructions per step that queued cells, at three waves per SIMD).  So it is
        // built, tested (tests/test_gpu_level0.py::test_fraction_stream) and OFF; CX_DEBUG=1 CX_TQ=1 turns it on.
        P.tq = nullptr; P.tlimit = 0;
        if (!fused && cx_debug_knob("CX_TQ", 0u)) {
            if ((rc = cx_grow(ctx, ctx->tq, ctx->tq_cap, need))) return rc;
            P.tq = ctx->tq; P.tlimit = T.wcap;
        }
        // the triangle stage walks the vertex stage's cell records.  The kernel that walks queue entries instead (no records: 80 MB
        // less HBM traffic per 512^3 extraction) is built and bit-identical, and measured no faster at 512^3 and slower on thin slabs
        // (DESIGN.md section 4): it runs on request (CX_DEBUG=1 CX_K2_ENTRIES=1)
        P.write_records = (!fused && !tiled && !cx_debug_knob("CX_K2_ENTRIES", 0u)) ? 1u : 0u;
        if (tiled) {
            P.qa = nullptr;       // nobody gathers the queue words: the stream kernel does not store them
            if ((rc = cx_grow(ctx, ctx->fj, ctx->fj_cap, (size_t)P.n0 * (4u * T.njg) * P.n2 + 64u))) return rc;
            if ((rc = cx_grow(ctx, ctx->fk, ctx->fk_cap, (size_t)P.n0 * P.n1 * (2u * T.nks) + 64u))) return rc;
            if ((rc = cx_grow(ctx, ctx->bnd, ctx->bnd_cap, (size_t)T.nblocks * 2u * T.bndcap))) return rc;
            if ((rc = cx_grow(ctx, ctx->bndn, ctx->bndn_cap, (size_t)T.nblocks * 2u))) return rc;
            if ((rc = cx_grow(ctx, ctx->torder, ctx->torder_cap, (size_t)T.nblocks * 6u))) return rc;
            P.fj = ctx->fj; P.fk = ctx->fk; P.bnd = ctx->bnd; P.bndn = ctx->bndn; P.torder = ctx->torder;
            P.tile_cap = cx_debug_knob("CX_TILE_CAP", cx_tile_cap_default());
        }
        P.nvw = cx_vertex_stage_waves(P);
        if ((rc = cx_grow(ctx, ctx->rstart, ctx->rstart_cap, (size_t)P.nvw + 1u))) return rc;
        P.rstart = ctx->rstart;
        P.nkw = cx_triangle_stage_waves(P);
        if ((rc = cx_grow(ctx, ctx->kstart, ctx->kstart_cap, (size_t)P.nkw + 1u))) return rc;
        P.kstart = ctx->kstart;
        ctx->last = P;
    }
    ctx->last_task = T;
    ctx->last_flags = flags;
    ctx->path = staged ? (fused ? 2 : (tiled ? 3 : 1)) : 0;
    ctx->records_valid = staged ? (P.write_records != 0u) : true;   // the generic path always writes them
    cx_ctx::evset* ev = nullptr;
    if (ctx->timing) {
        if (ctx->nevents < (int)(sizeof(ctx->events) / sizeof(ctx->events[0]))) {
            ev = &ctx->events[ctx->nevents++];
            for (int n = 0; n < 5; n++)
                if (!ev->e[n]) CX_HIP(ctx, hipEventCreate(&ev->e[n]));
        }
    }
    // the generic kernel accumulates into the counters; the staged pipeline's scan kernel writes all of them
    if (!staged || (flags & CX_DBG_PHASE_A_ONLY))
        CX_HIP(ctx, hipMemsetAsync(ctx->counters, 0, CX_CNT_WORDS * sizeof(uint32_t), ctx->stream));
    if (ev) CX_HIP(ctx, hipEventRecord(ev->e[0], ctx->stream));
    if (staged) {
        cx_launch_stream(P, T, ctx->stream);
        if (ev) CX_HIP(ctx, hipEventRecord(ev->e[1], ctx->stream));
        if (!(flags & CX_DBG_PHASE_A_ONLY)) cx_launch_scan_waves(P, T, ctx->stream);
        if (ev) CX_HIP(ctx, hipEventRecord(ev->e[2], ctx->stream));
        if (fused) cx_launch_emit_mesh(P, T, ctx->stream);
        else if (tiled) { if (!(flags & (CX_DBG_PHASE_A_ONLY | CX_DBG_COUNT_ONLY))) cx_launch_tile_emit(P, T, ctx->hash_xy, ctx->stream); }
        else if (!(flags & (CX_DBG_PHASE_A_ONLY | CX_DBG_COUNT_ONLY))) cx_launch_emit_vertices(P, T, ctx->stream);
    } else {
        cx_launch_classify_generic(P, ctx->stream);
        if (ev) CX_HIP(ctx, hipEventRecord(ev->e[1], ctx->stream));
        if (ev) CX_HIP(ctx, hipEventRecord(ev->e[2], ctx->stream));
    }
    if (ev) CX_HIP(ctx, hipEventRecord(ev->e[3], ctx->stream));
    if (!fused && !tiled && !(flags & (CX_DBG_NO_EMIT | CX_DBG_PHASE_A_ONLY | CX_DBG_COUNT_ONLY))) {
        if (staged && P.write_records) cx_launch_emit_triangles_q(P, T, ctx->hash_xy, ctx->stream);
        else if (staged) cx_launch_emit_triangles_e(P, T, ctx->hash_xy, ctx->stream);   // (A/B: CX_K2_ENTRIES)
        else cx_launch_emit_triangles(P, ctx->hash_xy, ctx->stream);
    }
    if (ev) CX_HIP(ctx, hipEventRecord(ev->e[4], ctx->stream));
    CX_HIP(ctx, hipGetLastError());
    ctx->extracted = true;
    ctx->counts_fetched = false;
    ctx->post_valid = false;
    ctx->keep_valid = false;
    cx_levels_invalidate(ctx); // the context's output buffers hold this extraction now, not a level of cx_extract3d_levels
    return CX_OK;
}

extern "C" int cx_extract3d_async(cx_ctx* ctx, double value, uint32_t flags) {
    if (!ctx) return CX_ERR_INVALID;
    CX_HIP(ctx, hipSetDevice(ctx->device));
    return enqueue_extract(ctx, value, flags);
}

extern "C" int cx_counts_get(cx_ctx* ctx, cx_counts* out) {
    if (!ctx || !out) return CX_ERR_INVALID;
    if (!ctx->extracted) return fail(ctx, CX_ERR_STATE, "no extraction enqueued");
    if (ctx->counts_fetched) { *out = ctx->counts; return CX_OK; }   // the device counters may belong to a later 4-D march
    CX_HIP(ctx, hipSetDevice(ctx->device));
    CX_HIP(ctx, hipMemcpyAsync(ctx->counters_host, ctx->counters, CX_CNT_WORDS * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, ctx->stream));
    CX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->n_cells = ctx->counters_host[CX_CNT_CELLS];
    out->n_vertices = ctx->counters_host[CX_CNT_VERTS];
    out->n_triangles = ctx->counters_host[CX_CNT_TRIS];
    out->n_border_voxels = ctx->counters_host[CX_CNT_BORDER];
    ctx->counts = *out;
    if (out->n_cells > ctx->ccap || out->n_vertices > ctx->vcap || out->n_triangles > ctx->tcap ||
        (ctx->last.flat && ctx->counters_host[CX_CNT_BATCHES] > ctx->last.fcap)) {
        ctx->extracted = false;
        return fail(ctx, CX_ERR_CAPACITY, "output buffers too small for this isosurface");
    }
    if (ctx->path == 3 && (ctx->counters_host[CX_CNT_NEAR] != 0u || ctx->counters_host[CX_CNT_TILEOVF] != 0u)) {
        // a wave on the tolerance path, or a tile with more surface cells than a workgroup's LDS words: the tile kernels have
        // written nothing (or not everything) -- the same extraction again through the staged kernels
        int rc = enqueue_extract(ctx, ctx->last.value, (ctx->last_flags & ~(uint32_t)CX_KERNEL_TILED) | CX_KERNEL_STAGED);
        if (rc) return rc;
        return cx_counts_get(ctx, out);
    }
    if (ctx->path == 2 && ctx->counters_host[CX_CNT_NEAR] != 0u) {
        // a sample within the reference's np.allclose tolerances of the isovalue: its rules may drop vertices, the fused
        // kernel has written nothing -- the same extraction again through the staged kernels (exact per-cell path)
        int rc = enqueue_extract(ctx, ctx->last.value, (ctx->last_flags & ~(uint32_t)CX_KERNEL_FUSED) | CX_KERNEL_STAGED);
        if (rc) return rc;
        return cx_counts_get(ctx, out);
    }
    ctx->counts_fetched = true;
    return CX_OK;
}

extern "C" int cx_extract3d(cx_ctx* ctx, double value, uint32_t flags, cx_counts* out) {
    if (!ctx) return CX_ERR_INVALID;
    CX_HIP(ctx, hipSetDevice(ctx->device));
    cx_counts c;
    for (int attempt = 0; attempt < 3; attempt++) {
        int rc = enqueue_extract(ctx, value, flags);
        if (rc) return rc;
        rc = cx_counts_get(ctx, &c);
        if (out) *out = c;
        if (rc != CX_ERR_CAPACITY) return rc;
        // grow to what this surface needs (+5 %) and run again
        rc = cx_reserve(ctx, c.n_cells + c.n_cells / 20 + 1024, c.n_vertices + c.n_vertices / 20 + 1024,
                        c.n_triangles + c.n_triangles / 20 + 1024);
        if (rc) return rc;
    }
    return fail(ctx, CX_ERR_CAPACITY, "output buffers still too small after growing");
}

// cell records {lin, sign|tetskip<<8|ntri<<16|emask<<24, first triangle, first vertex} of the last extraction: the fused emit
// kernel does not write them; callers that walk the surface voxels (seeded selection) get them from the vertex stage of the
// staged kernels run over the same queues with its other stores switched off
int cx_ensure_cell_records(cx_ctx* ctx) {
    if (!ctx->extracted) return fail(ctx, CX_ERR_STATE, "no valid extraction");
    if (ctx->records_valid) return CX_OK;
    cx_params P = ctx->last;
    P.flags |= CX_DBG_NO_VERTS | CX_DBG_NO_CELLTAB;
    P.write_records = 1u;
    P.ccap = ctx->ccap; P.cells = ctx->cells;
    cx_launch_emit_vertices(P, ctx->last_task, ctx->stream);
    CX_HIP(ctx, hipGetLastError());
    ctx->records_valid = true;
    return CX_OK;
}

extern "C" int cx_level0_path(cx_ctx* ctx, int* path) {
    if (!ctx || !path) return CX_ERR_INVALID;
    *path = ctx->path;
    return CX_OK;
}

// the vertex records of the current extraction expanded to float4 {x, y, z, bits(edge id)} in a buffer of the context
int cx_level0_expanded(cx_ctx* ctx, float4** out) {
    const size_t nv = (size_t)ctx->counts.n_vertices;
    if (ctx->verts_xyz_cap < nv) {
        const int rc = cx_grow(ctx, ctx->verts_xyz, ctx->verts_xyz_cap, nv + nv / 16 + 64);
        if (rc) return rc;
    }
    cx_launch_expand_verts(ctx->verts, ctx->verts_xyz, (uint32_t)nv, (uint32_t)ctx->n1, (uint32_t)ctx->n2, ctx->stream);
    CX_HIP(ctx, hipGetLastError());
    *out = ctx->verts_xyz;
    return CX_OK;
}

extern "C" int cx_level0_download(cx_ctx* ctx, float* verts_xyzk, int32_t* tris) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->extracted) return fail(ctx, CX_ERR_STATE, "no valid extraction");
    CX_HIP(ctx, hipSetDevice(ctx->device));
    float4* xyz = nullptr;
    if (verts_xyzk && ctx->counts.n_vertices) {
        const int rc = cx_level0_expanded(ctx, &xyz);
        if (rc) return rc;
    }
    void* d[2] = {xyz ? (void*)verts_xyzk : nullptr, (tris && ctx->counts.n_triangles) ? (void*)tris : nullptr};
    const void* sp[2] = {xyz, ctx->tris};
    const size_t nb[2] = {(size_t)ctx->counts.n_vertices * sizeof(float4), (size_t)ctx->counts.n_triangles * 3 * sizeof(int32_t)};
    return cx_copy_to_host(ctx, 2, d, sp, nb);
}

extern "C" int cx_level0_device_ptrs(cx_ctx* ctx, void** verts_xyzk, void** tris) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->extracted) return fail(ctx, CX_ERR_STATE, "no valid extraction");
    CX_HIP(ctx, hipSetDevice(ctx->device));
    if (verts_xyzk) {
        cx_counts c;
        int rc = cx_counts_get(ctx, &c);
        if (rc) return rc;
        float4* xyz = nullptr;
        if ((rc = cx_level0_expanded(ctx, &xyz))) return rc;
        *verts_xyzk = xyz;
    }
    if (tris) *tris = ctx->tris;
    return CX_OK;
}

extern "C" int cx_level0_download_records(cx_ctx* ctx, uint32_t* vertex_records, int32_t* tris) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->extracted) return fail(ctx, CX_ERR_STATE, "no valid extraction");
    CX_HIP(ctx, hipSetDevice(ctx->device));
    cx_counts c;
    const int rc = cx_counts_get(ctx, &c);
    if (rc) return rc;
    void* d[2] = {(vertex_records && c.n_vertices) ? (void*)vertex_records : nullptr, (tris && c.n_triangles) ? (void*)tris : nullptr};
    const void* sp[2] = {ctx->verts, ctx->tris};
    const size_t nb[2] = {(size_t)c.n_vertices * sizeof(cx_vrec), (size_t)c.n_triangles * 3 * sizeof(int32_t)};
    return cx_copy_to_host(ctx, 2, d, sp, nb);
}

extern "C" int cx_level0_device_records(cx_ctx* ctx, void** vertex_records, void** tris) {
    if (!ctx) return CX_ERR_INVALID;
    if (!ctx->extracted) return fail(ctx, CX_ERR_STATE, "no valid extraction");
    if (vertex_records) *vertex_records = ctx->verts;
    if (tris) *tris = ctx->tris;
    return CX_OK;
}

// diagnostic: allocate (words > 0) or drop (0) the per-wave stamp buffer; copy it out with host != NULL
extern "C" int cx_debug_stamps(cx_ctx* ctx, int64_t words, unsigned long long* host) {
    if (!ctx) return CX_ERR_INVALID;
    CX_HIP(ctx, hipSetDevice(ctx->device));
    CX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (host && ctx->stamps) {
        CX_HIP(ctx, hipMemcpy(host, ctx->stamps, ctx->stamps_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return CX_OK;
    }
    if (ctx->stamps) (void)hipFree(ctx->stamps);
    ctx->stamps = nullptr; ctx->stamps_words = 0;
    if (words > 0) {
        CX_HIP(ctx, hipMalloc(&ctx->stamps, (size_t)words * sizeof(unsigned long long)));
        CX_HIP(ctx, hipMemset(ctx->stamps, 0, (size_t)words * sizeof(unsigned long long)));
        ctx->stamps_words = (size_t)words;
    }
    return CX_OK;
}

// ---- measurement: what a plain streaming read of `bytes` bytes gets on this device (16-byte loads, grid-stride, 8 in flight per
// lane, result folded into one word per workgroup so that nothing is optimised away).  The ceiling the stream kernel is compared
// with in bench.py (`roofline.measured_peak_GBps`), measured in the process that runs the bench.
__global__ __launch_bounds__(256) void cx_k_read_bw(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 256u;
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 7u * stride < n16; i += 8u * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[i + (size_t)u * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9E3779B9u) sink[blockIdx.x & 255u] = acc;    // (practically never: keeps the loads alive)
}
extern "C" int cx_measure_read_bandwidth(cx_ctx* ctx, const void* device_ptr, int64_t bytes, int reps, double* out_GBps) {
    if (!ctx || !device_ptr || bytes < (1 << 20) || reps < 1 || !out_GBps) return ctx ? fail(ctx, CX_ERR_INVALID, "cx_measure_read_bandwidth: bad argument") : CX_ERR_INVALID;
    CX_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->counters) return fail(ctx, CX_ERR_STATE, "context has no scratch words yet (extract once first)");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    CX_HIP(ctx, hipEventCreate(&e0));
    CX_HIP(ctx, hipEventCreate(&e1));
    const size_t n16 = (size_t)bytes / 16u;
    uint32_t* sink = nullptr;
    CX_HIP(ctx, hipMalloc(&sink, 256 * sizeof(uint32_t)));
    hipLaunchKernelGGL(cx_k_read_bw, dim3(256 * 8), dim3(256), 0, ctx->stream, static_cast<const uint4*>(device_ptr), n16, sink);   // warm
    double best = 0.0;
    const uint32_t grids[3] = {256u * 8u, 256u * 16u, 256u * 32u};     // workgroups: 8, 16, 32 per CU's worth (the best one counts)
    for (int r = 0; r < reps; r++)
        for (uint32_t g : grids) {
            CX_HIP(ctx, hipEventRecord(e0, ctx->stream));
            hipLaunchKernelGGL(cx_k_read_bw, dim3(g), dim3(256), 0, ctx->stream, static_cast<const uint4*>(device_ptr), n16, sink);
            CX_HIP(ctx, hipEventRecord(e1, ctx->stream));
            CX_HIP(ctx, hipEventSynchronize(e1));
            float ms = 0.f;
            CX_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            if (ms > 0.f) best = fmax(best, (double)n16 * 16.0 / (ms * 1e-3) / 1e9);
        }
    (void)hipFree(sink);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *out_GBps = best;
    return CX_OK;
}

extern "C" int cx_timing_enable(cx_ctx* ctx, int on) {
    if (!ctx) return CX_ERR_INVALID;
    ctx->timing = on != 0;
    ctx->nevents = 0;
    return CX_OK;
}

extern "C" int cx_timing_read(cx_ctx* ctx, double ms[8], int* n) {
    if (!ctx || !ms) return CX_ERR_INVALID;
    CX_HIP(ctx, hipSetDevice(ctx->device));
    CX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 8; k++) ms[k] = 0.0;
    for (int i = 0; i < ctx->nevents; i++) {
        float d[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; k++) CX_HIP(ctx, hipEventElapsedTime(&d[k], ctx->events[i].e[k], ctx->events[i].e[k + 1]));
        ms[0] += d[0] + d[1] + d[2]; ms[1] += d[3]; ms[2] += d[0] + d[1] + d[2] + d[3];
        ms[3] += d[0]; ms[4] += d[1]; ms[5] += d[2];
    }
    if (n) *n = ctx->nevents;
    ctx->nevents = 0;
    return CX_OK;
}
