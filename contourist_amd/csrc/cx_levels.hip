// cx_levels.hip -- several isovalues of ONE grid in one call (BASELINE.json config 5; the reference has the idea only in
// 2-D: multiple_2d_contour.Multiple2DContourGrid classifies an edge against all sorted levels at once,
// multiple_2d_contour.py:48-59).
//
// The stream kernel -- the pass over the samples -- runs ONCE for all levels: its workgroups are numbered (tile, level) with
// the level running fastest inside an XCD's sequence, so the workgroups that stream one tile for the different levels run
// next to each other on one XCD and the tile comes from HBM once (cx_k_stream_levels).  Every level has its own queues and
// side tables; scan, vertex stage and triangle stage then run per level exactly as for a single extraction, so each level's
// mesh is bit for bit what cx_extract3d gives for that isovalue (tests/test_gpu_levels.py).
// cx_levels_select makes one level the context's current extraction: download, post-passes and seeded selection act on it.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "cx_ctx.h"

#define CXL_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return (e__ == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;                      \
        }                                                                                        \
    } while (0)

struct cx_level_slot {
    double value = 0.0;
    cx_params P;                       // parameters of the level (valid after cx_extract3d_levels)
    cx_counts counts = {0, 0, 0, 0};
    // per-level side tables of the staged pipeline
    uint32_t* queue = nullptr;
    size_t queue_cap = 0;
    cx_wsum* wsum = nullptr;
    size_t wsum_cap = 0;
    cx_wbase* wbase = nullptr;
    size_t wbase_cap = 0;
    cx_brec* brec = nullptr;
    size_t brec_cap = 0;
    cx_bdesc* flat = nullptr;
    size_t flat_cap = 0;
    uint32_t* qa = nullptr;
    size_t qa_cap = 0;
    uint32_t* counters = nullptr;      // a slice of cx_levels_state::counters_all (not owned)
    uint32_t* chunksum = nullptr;      // a slice of cx_levels_state::chunk_all (not owned)
    uint32_t* rstart = nullptr;        // vertex stage: first batch of every wave's share of the rounds
    size_t rstart_cap = 0;
    uint32_t* kstart = nullptr;        // triangle stage: the same
    size_t kstart_cap = 0;
    // Level-0 outputs of the level (swapped with the context's while the level is selected)
    cx_vrec* verts = nullptr;
    uint4* cells = nullptr;
    int32_t* tris = nullptr;
    uint32_t vcap = 0, ccap = 0, tcap = 0;
};

#ifndef CXL_SIDES
#define CXL_SIDES 3      // side streams the emit stages CAN use (knob); one is used by default
#endif
struct cx_levels_state {
    std::vector<cx_level_slot> slots;
    int nvalid = 0;                    // levels of the last cx_extract3d_levels
    cx_params* dparams = nullptr;      // device copy of the levels' parameters
    size_t dparams_cap = 0;
    uint32_t* hcounters = nullptr;     // pinned: CX_CNT_WORDS per level
    size_t hcounters_cap = 0;
    // the levels' counters and chunk totals side by side: ONE memset and ONE copy back per call (a memset and a copy per level were
    // 16 small stream operations in front of and behind the scans: ~0.1 ms of a 2.3 ms call)
    uint32_t* counters_all = nullptr;
    size_t counters_all_cap = 0;
    uint32_t* chunk_all = nullptr;
    size_t chunk_all_cap = 0;
    cx_params* hparams = nullptr;      // pinned staging of the levels' parameters (the upload needs no wait)
    size_t hparams_cap = 0;
    cx_task T;
    uint32_t flags = 0;
    // the emit stages of two levels run side by side (each kernel alone leaves part of the chip idle): a second stream, and a
    // second set of info words for the levels that run on it
    // the emit stages of the levels run on CXL_SIDES + 1 streams (the context's and these), level l on stream l % (CXL_SIDES + 1)
    hipStream_t side[CXL_SIDES] = {};
    hipEvent_t ev_fork = nullptr, ev_join[CXL_SIDES] = {};
    uint64_t* info_side[CXL_SIDES] = {};       // info words of the levels in flight on the side streams (unpooled queues)
    size_t info_side_cap[CXL_SIDES] = {};
    // ONE pool of queue entries for all levels (every level a slice of every streaming wave's region); false after a call
    // whose surface overflowed a slice: that grid then gets full-size regions per level, as in round 2
    bool pooled = true;
    uint32_t* qpool = nullptr;
    size_t qpool_cap = 0;
    int64_t pooled_off_for[3] = {0, 0, 0};   // the grid shape `pooled == false` was decided for
};

static void free_slot(cx_level_slot& S) {
    void* all[] = {S.queue, S.wsum, S.wbase, S.brec, S.flat, S.qa, S.rstart, S.kstart, S.verts, S.cells, S.tris};
    for (void* p : all)
        if (p) (void)hipFree(p);
    S = cx_level_slot();
}

void cx_levels_free(cx_ctx* ctx) {
    cx_levels_state* L = ctx->lv;
    if (!L) return;
    // the buffers of the selected level sit in the context (and the context's own in that slot): give them back first so that
    // both sides free what they own
    if (ctx->lv_current >= 0 && ctx->lv_current < (int)L->slots.size()) {
        cx_level_slot& S = L->slots[ctx->lv_current];
        std::swap(ctx->verts, S.verts); std::swap(ctx->cells, S.cells); std::swap(ctx->tris, S.tris);
        std::swap(ctx->vcap, S.vcap); std::swap(ctx->ccap, S.ccap); std::swap(ctx->tcap, S.tcap);
        ctx->lv_current = -1;
    }
    for (auto& S : L->slots) free_slot(S);
    cx_release(L->dparams, L->dparams_cap);
    if (L->hcounters) (void)hipHostFree(L->hcounters);
    if (L->hparams) (void)hipHostFree(L->hparams);
    cx_release(L->counters_all, L->counters_all_cap);
    cx_release(L->chunk_all, L->chunk_all_cap);
    for (int k = 0; k < CXL_SIDES; k++) cx_release(L->info_side[k], L->info_side_cap[k]);
    cx_release(L->qpool, L->qpool_cap);
    for (int k = 0; k < CXL_SIDES; k++) {
        if (L->side[k]) (void)hipStreamDestroy(L->side[k]);
        if (L->ev_join[k]) (void)hipEventDestroy(L->ev_join[k]);
    }
    if (L->ev_fork) (void)hipEventDestroy(L->ev_fork);
    delete L;
    ctx->lv = nullptr;
}

template <typename Tp, typename C>
static int grow(cx_ctx* ctx, Tp*& ptr, C& cap, size_t need) { return cx_grow(ctx, ptr, cap, need); }

// a single-level extraction is about to overwrite the context's output buffers: the levels are gone
void cx_levels_invalidate(cx_ctx* ctx) {
    if (ctx->lv) ctx->lv->nvalid = 0;
    ctx->lv_current = -1;
}

static void unselect(cx_ctx* ctx) {
    cx_levels_state* L = ctx->lv;
    if (!L || ctx->lv_current < 0 || ctx->lv_current >= (int)L->slots.size()) { ctx->lv_current = -1; return; }
    cx_level_slot& S = L->slots[ctx->lv_current];
    std::swap(ctx->verts, S.verts); std::swap(ctx->cells, S.cells); std::swap(ctx->tris, S.tris);
    std::swap(ctx->vcap, S.vcap); std::swap(ctx->ccap, S.ccap); std::swap(ctx->tcap, S.tcap);
    ctx->lv_current = -1;
}

extern "C" int cx_levels_select(cx_ctx* ctx, int32_t index) {
    if (!ctx) return CX_ERR_INVALID;
    cx_levels_state* L = ctx->lv;
    if (!L || index < 0 || index >= L->nvalid) { ctx->err = "cx_levels_select: no such level (call cx_extract3d_levels first)"; return CX_ERR_STATE; }
    CXL_HIP(ctx, hipSetDevice(ctx->device));
    CXL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unselect(ctx);
    cx_level_slot& S = L->slots[index];
    std::swap(ctx->verts, S.verts); std::swap(ctx->cells, S.cells); std::swap(ctx->tris, S.tris);
    std::swap(ctx->vcap, S.vcap); std::swap(ctx->ccap, S.ccap); std::swap(ctx->tcap, S.tcap);
    ctx->lv_current = index;
    ctx->last = S.P;
    ctx->last_task = L->T;
    ctx->last_flags = L->flags;
    ctx->counts = S.counts;
    ctx->extracted = true;
    ctx->counts_fetched = true;
    ctx->post_valid = false;
    ctx->keep_valid = false;
    ctx->records_valid = true;
    ctx->path = 1;
    return CX_OK;
}

extern "C" int cx_extract3d_levels(cx_ctx* ctx, const double* values, int32_t nlevels, uint32_t flags, cx_counts* out_counts) {
    if (!ctx || !values || nlevels < 1 || nlevels > 64) return CX_ERR_INVALID;
    if (!ctx->grid) { ctx->err = "no grid: call cx_grid_upload or cx_grid_adopt_device first"; return CX_ERR_STATE; }
    if ((flags & ~(uint32_t)(CX_DIAG_CPYTHON310)) != 0u) { ctx->err = "cx_extract3d_levels: only the diagonal flag is accepted"; return CX_ERR_INVALID; }
    for (int l = 0; l < nlevels; l++)
        if (!(values[l] == values[l])) { ctx->err = "isovalue is NaN"; return CX_ERR_INVALID; }
    CXL_HIP(ctx, hipSetDevice(ctx->device));
    if (!cx_fast_classify_supported_dims(ctx->n2, ctx->grid)) {
        ctx->err = "cx_extract3d_levels needs rows of at least 4 samples: extract the levels one by one with cx_extract3d";
        return CX_ERR_UNSUPPORTED;
    }
    if (!ctx->lv) ctx->lv = new cx_levels_state();
    cx_levels_state* L = ctx->lv;
    unselect(ctx);
    L->nvalid = 0;
    // nothing is selected from here on: an error return below must not leave the context describing a level that is gone
    ctx->extracted = false; ctx->counts_fetched = false; ctx->post_valid = false; ctx->keep_valid = false;
    if ((int)L->slots.size() < nlevels) L->slots.resize(nlevels);
    const int64_t N = ctx->n0 * ctx->n1 * ctx->n2;
    int rc = cx_ensure_hash_xy(ctx, flags);
    if (rc) return rc;
    const cx_task T = cx_fast_task((uint32_t)ctx->n0, (uint32_t)ctx->n1, (uint32_t)ctx->n2);
    L->T = T;
    L->flags = flags;
    const size_t nw = (size_t)T.nblocks * 4u, need = nw * T.wcap;
    const size_t boundary = (size_t)(ctx->n0 * ctx->n1 + ctx->n0 * ctx->n2 + ctx->n1 * ctx->n2);
    // batches: one short batch per streaming wave + one per CX_BATCH_MIN queued cells; the cell count is not known yet, so
    // room for a surface through a quarter of all cells (more: CX_ERR_CAPACITY, extract such levels one by one)
    const size_t nflat = nw + (size_t)N / 4u / 512u + boundary / 128u + 4096u;
    // Queue entries: worst case one per sample and level.  Pooled (the default): the levels share ONE region of `need` entries,
    // level l owning entries [l sub, l sub + sub) of every streaming wave's stretch of T.wcap -- 0.54 GB + 1.07 GB of info words at
    // 512^3 whatever the number of levels (round 2: 0.6 GB per level + 2 x 1.07 GB).  A wave whose cells do not fit its slice
    // raises the overflow flag; the call is then repeated with full-size regions per level, and that stays so for this grid shape.
    if (!L->pooled && (L->pooled_off_for[0] != ctx->n0 || L->pooled_off_for[1] != ctx->n1 || L->pooled_off_for[2] != ctx->n2)) L->pooled = true;
    uint32_t sub = (T.wcap / (uint32_t)nlevels) & ~63u;
    if (cx_debug_knob("CX_LEVELS_SLICE", 0u)) sub = cx_debug_knob("CX_LEVELS_SLICE", 0u) & ~63u;   // tests: force small slices
    // Measured on the bench field (512^3, smooth): a sheet that runs along a wave's 4-row tile passes through up to half of the
    // wave's cells, so slices of a quarter (4 levels) or an eighth (8 levels) of the region overflow and the call falls back --
    // the pool is only tried with up to 3 levels (2 levels: 2.5 GB held instead of 3.7), more levels take full-size regions at once.
    bool pooled = L->pooled && sub >= 64u && (nlevels <= 3 || cx_debug_knob("CX_LEVELS_SLICE", 0u)) && !cx_debug_knob("CX_LEVELS_NO_POOL", 0u);
    if (pooled) {
        if ((rc = grow(ctx, L->qpool, L->qpool_cap, need))) return rc;
        for (auto& S : L->slots) cx_release(S.queue, S.queue_cap);      // full-size regions of an earlier call
        for (int k = 0; k < CXL_SIDES; k++) cx_release(L->info_side[k], L->info_side_cap[k]);
    } else {
        cx_release(L->qpool, L->qpool_cap);
    }
    const size_t chunk_words = (((nw + 255u) / 256u) * 8u + 63u) & ~(size_t)63u;      // per level, a multiple of 256 bytes
    if ((rc = grow(ctx, L->counters_all, L->counters_all_cap, (size_t)nlevels * CX_CNT_WORDS))) return rc;
    if ((rc = grow(ctx, L->chunk_all, L->chunk_all_cap, (size_t)nlevels * chunk_words))) return rc;
    CXL_HIP(ctx, hipMemsetAsync(L->chunk_all, 0, (size_t)nlevels * chunk_words * sizeof(uint32_t), ctx->stream));
    for (int l = 0; l < nlevels; l++) {
        cx_level_slot& S = L->slots[l];
        S.value = values[l];
        if (!pooled && (rc = grow(ctx, S.queue, S.queue_cap, need))) return rc;
        if ((rc = grow(ctx, S.wsum, S.wsum_cap, nw))) return rc;
        if ((rc = grow(ctx, S.wbase, S.wbase_cap, nw))) return rc;
        if ((rc = grow(ctx, S.brec, S.brec_cap, nw * T.bcap))) return rc;
        if ((rc = grow(ctx, S.flat, S.flat_cap, nflat))) return rc;
        if ((rc = grow(ctx, S.qa, S.qa_cap, nw * CX_SWP * 64u + 64u))) return rc;
        S.counters = L->counters_all + (size_t)l * CX_CNT_WORDS;
        S.chunksum = L->chunk_all + (size_t)l * chunk_words;
        cx_params& P = S.P;
        memset(&P, 0, sizeof(P));
        P.grid = ctx->grid;
        P.n0 = (uint32_t)ctx->n0; P.n1 = (uint32_t)ctx->n1; P.n2 = (uint32_t)ctx->n2;
        P.nsamples = (uint32_t)N;
        P.div_plane = cx_fdiv_make(P.n1 * P.n2);
        P.div_row = cx_fdiv_make(P.n2);
        P.div_ci = cx_fdiv_make(T.ci);
        cx_fill_value_params(P, values[l]);
        P.flags = flags;
        P.org0 = (uint32_t)ctx->origin[0]; P.org1 = (uint32_t)ctx->origin[1]; P.org2 = (uint32_t)ctx->origin[2];
        P.counters = S.counters;
        P.queue = pooled ? L->qpool + (size_t)l * sub : S.queue;
        P.qlimit = pooled ? sub : T.wcap;
        P.write_records = 1u;
        P.wsum = S.wsum; P.wbase = S.wbase; P.brec = S.brec; P.flat = S.flat; P.fcap = (uint32_t)nflat;
        P.qa = S.qa;
        P.chunksum = S.chunksum;
        P.nvw = cx_vertex_stage_waves(P);
        if ((rc = grow(ctx, S.rstart, S.rstart_cap, (size_t)P.nvw + 1u))) return rc;
        P.rstart = S.rstart;
        P.nkw = cx_triangle_stage_waves(P);
        if ((rc = grow(ctx, S.kstart, S.kstart_cap, (size_t)P.nkw + 1u))) return rc;
        P.kstart = S.kstart;
        P.verts = S.verts; P.cells = S.cells; P.tris = S.tris;
        P.vcap = S.vcap; P.ccap = S.ccap; P.tcap = S.tcap;
    }
    // the staged kernels' word per queue entry is shared: the levels' vertex and triangle stages run one level after the other
    if ((rc = grow(ctx, ctx->info64, ctx->info64_cap, need))) return rc;
    if ((rc = grow(ctx, L->dparams, L->dparams_cap, (size_t)nlevels))) return rc;
    if (L->hcounters_cap < (size_t)nlevels) {
        if (L->hcounters) (void)hipHostFree(L->hcounters);
        L->hcounters = nullptr; L->hcounters_cap = 0;
        CXL_HIP(ctx, hipHostMalloc(&L->hcounters, (size_t)nlevels * CX_CNT_WORDS * sizeof(uint32_t)));
        L->hcounters_cap = (size_t)nlevels;
    }
    if (L->hparams_cap < (size_t)nlevels) {
        if (L->hparams) (void)hipHostFree(L->hparams);
        L->hparams = nullptr; L->hparams_cap = 0;
        CXL_HIP(ctx, hipHostMalloc(&L->hparams, (size_t)nlevels * sizeof(cx_params)));
        L->hparams_cap = (size_t)nlevels;
    }
    {
        cx_params* hp = L->hparams;       // pinned: the copy below is asynchronous and nothing has to wait for it (every call ends synchronised)
        for (int l = 0; l < nlevels; l++) {
            cx_level_slot& S = L->slots[l];
            S.P.verts = S.verts; S.P.cells = S.cells; S.P.tris = S.tris;
            S.P.vcap = S.vcap; S.P.ccap = S.ccap; S.P.tcap = S.tcap;
            S.P.info64 = pooled ? ctx->info64 + (size_t)l * sub : ctx->info64;
            hp[l] = S.P;
        }
        CXL_HIP(ctx, hipMemcpyAsync(L->dparams, hp, (size_t)nlevels * sizeof(cx_params), hipMemcpyHostToDevice, ctx->stream));
        // ONE pass over the samples for all levels, then the scans
        cx_launch_stream_levels(hp, T, (uint32_t)nlevels, ctx->stream);
        cx_launch_scan_levels(L->dparams, T, (uint32_t)nlevels, ctx->stream);
        CXL_HIP(ctx, hipMemcpyAsync(L->hcounters, L->counters_all, (size_t)nlevels * CX_CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        CXL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // output buffers of every level, sized by what its surface needs
        for (int l = 0; l < nlevels; l++) {
            cx_level_slot& S = L->slots[l];
            const uint32_t* hc = L->hcounters + (size_t)l * CX_CNT_WORDS;
            S.counts.n_cells = hc[CX_CNT_CELLS]; S.counts.n_vertices = hc[CX_CNT_VERTS];
            S.counts.n_triangles = hc[CX_CNT_TRIS]; S.counts.n_border_voxels = hc[CX_CNT_BORDER];
            if (pooled && hc[CX_CNT_OVERFLOW]) {
                // a level's surface is denser than a slice of the pool holds somewhere: the same call with full-size regions
                L->pooled = false;
                L->pooled_off_for[0] = ctx->n0; L->pooled_off_for[1] = ctx->n1; L->pooled_off_for[2] = ctx->n2;
                return cx_extract3d_levels(ctx, values, nlevels, flags, out_counts);
            }
            if (hc[CX_CNT_BATCHES] > nflat) {
                ctx->err = "cx_extract3d_levels: a level's surface passes through too many cells for the batch list: extract it with cx_extract3d";
                return CX_ERR_CAPACITY;
            }
            if ((uint64_t)S.counts.n_vertices > 0xFFFFFFF0ULL || (uint64_t)S.counts.n_triangles > 0x7FFFFFF0ULL) {
                ctx->err = "capacity beyond 32-bit indices";
                return CX_ERR_UNSUPPORTED;
            }
            size_t c = S.ccap, v = S.vcap, t = S.tcap;
            if ((rc = grow(ctx, S.cells, c, (size_t)S.counts.n_cells + 64u))) return rc;
            if ((rc = grow(ctx, S.verts, v, (size_t)S.counts.n_vertices + 64u))) return rc;
            {
                size_t t3 = (size_t)S.tcap * 3u;
                if ((rc = grow(ctx, S.tris, t3, ((size_t)S.counts.n_triangles + 64u) * 3u))) return rc;
                t = t3 / 3u;
            }
            S.ccap = (uint32_t)c; S.vcap = (uint32_t)v; S.tcap = (uint32_t)t;
            S.P.verts = S.verts; S.P.cells = S.cells; S.P.tris = S.tris;
            S.P.vcap = S.vcap; S.P.ccap = S.ccap; S.P.tcap = S.tcap;
            S.P.info64 = pooled ? ctx->info64 + (size_t)l * sub : ctx->info64;
        }
        // vertex and triangle stages, several levels side by side: level l on stream l % nstr (0 = the context's stream), the side
        // streams with their own info words (the staged kernels of one level hand over through them)
        // (Also measured and dropped: the levels in two groups, the second group's pass over the samples on its own stream while the
        // first group is in its emit stages -- 2.59 against 2.35 ms: the stream kernel works the samples once per level, 1.03 ms for 8
        // levels, bound by its own instructions as much as by HBM, and the emit stages are issue-bound too: nothing to overlap.)
        // Two by default: three and four were measured (CX_DEBUG=1 CX_LEVELS_STREAMS=n, tools/levels_streams.py) and change nothing
        // (2.33-2.42 ms for 8 levels of the bench grid whatever n): two levels in flight already fill the chip.
        int nstr = (nlevels > 1 && !cx_debug_knob("CX_LEVELS_ONE_STREAM", 0)) ? 2 : 1;
        if (nstr > 1) {
            const int want = cx_debug_knob("CX_LEVELS_STREAMS", 0);
            if (want >= 2 && want <= CXL_SIDES + 1) nstr = std::min(nlevels, want);
        }
        if (nstr > 1) {
            if (!L->ev_fork) CXL_HIP(ctx, hipEventCreateWithFlags(&L->ev_fork, hipEventDisableTiming));
            for (int k = 0; k + 1 < nstr; k++) {
                if (!L->side[k]) CXL_HIP(ctx, hipStreamCreateWithFlags(&L->side[k], hipStreamNonBlocking));
                if (!L->ev_join[k]) CXL_HIP(ctx, hipEventCreateWithFlags(&L->ev_join[k], hipEventDisableTiming));
                if (!pooled && (rc = grow(ctx, L->info_side[k], L->info_side_cap[k], ctx->info64_cap))) return rc;   // pooled: every level has its own slice of the info words
            }
            CXL_HIP(ctx, hipEventRecord(L->ev_fork, ctx->stream));
            for (int k = 0; k + 1 < nstr; k++) CXL_HIP(ctx, hipStreamWaitEvent(L->side[k], L->ev_fork, 0));
        }
        for (int l = 0; l < nlevels; l++) {
            cx_level_slot& S = L->slots[l];
            const int k = l % nstr;
            hipStream_t st = k ? L->side[k - 1] : ctx->stream;
            if (!pooled) S.P.info64 = k ? L->info_side[k - 1] : ctx->info64;
            cx_launch_emit_vertices(S.P, T, st);
            cx_launch_emit_triangles_q(S.P, T, ctx->hash_xy, st);
        }
        {
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                for (int k = 0; k + 1 < nstr; k++) (void)hipStreamSynchronize(L->side[k]);   // no forked stream is left running behind an error return
                ctx->err = std::string("cx_extract3d_levels: ") + hipGetErrorString(le);
                return CX_ERR_HIP;
            }
        }
        for (int k = 0; k + 1 < nstr; k++) {
            CXL_HIP(ctx, hipEventRecord(L->ev_join[k], L->side[k]));
            CXL_HIP(ctx, hipStreamWaitEvent(ctx->stream, L->ev_join[k], 0));
        }
        CXL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    L->nvalid = nlevels;
    if (out_counts)
        for (int l = 0; l < nlevels; l++) out_counts[l] = L->slots[l].counts;
    return cx_levels_select(ctx, 0);
}
