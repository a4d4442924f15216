// cx_ctx.h -- the context object behind the C ABI (host side only).
#pragma once
#include <string>

#include "cx_common.h"

struct cx_post_state;  // Level-1 buffers (cx_post.hip)
struct cx_state4;       // 4-D march state (cx_api4d.hip)
struct cx_state2;       // 2-D contour lines (cx_contour2d.hip)
struct cx_levels_state; // several isovalues of one grid (cx_levels.hip)

struct cx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // stream in use
    hipStream_t own_stream = nullptr;  // stream created by the context
    std::string err;
    // sampled field
    const float* grid = nullptr;
    float* grid_owned = nullptr;
    size_t grid_owned_bytes = 0;
    int64_t n0 = 0, n1 = 0, n2 = 0;
    double* grid64 = nullptr;          // float64 originals of the samples (cx_grid_shadow_f64): Level 1 interpolates on these
    size_t grid64_cap = 0;
    bool grid64_valid = false;
    // side tables of the march
    uint64_t* celltab = nullptr;
    size_t tables_for = 0;
    uint32_t* queue = nullptr;         // staged pipeline: per-wave queues, totals and offsets
    size_t queue_cap = 0;
    cx_wsum* wsum = nullptr;
    size_t wsum_cap = 0;
    cx_wbase* wbase = nullptr;
    size_t wbase_cap = 0;
    cx_brec* brec = nullptr;
    size_t brec_cap = 0;
    cx_bdesc* flat = nullptr;
    size_t flat_cap = 0;
    uint32_t* qa = nullptr;            // fused emit: queue positions / active cells per plane step and lane of the streaming waves
    size_t qa_cap = 0;
    uint32_t* info = nullptr;          // fused emit: per queue entry, first vertex of the cell in its wave | crossing mask
    size_t info_cap = 0;
    float* tq = nullptr;               // staged kernels: the stream kernel's interpolation fractions, one region per streaming wave (cx_params::tq)
    size_t tq_cap = 0;
    uint64_t* info64 = nullptr;        // staged kernels: per queue entry, (crossing mask << 32) | first vertex
    size_t info64_cap = 0;
    uint32_t* chunksum = nullptr;      // totals of every 256 streaming waves (cx_params::chunksum)
    size_t chunksum_cap = 0;
    uint32_t* rstart = nullptr;        // vertex stage: first batch of every wave's share of the rounds (cx_params::rstart)
    size_t rstart_cap = 0;
    uint32_t* kstart = nullptr;        // the same for the triangle stage (cx_params::kstart)
    size_t kstart_cap = 0;
    uint64_t* fj = nullptr;            // tile emit path: face words (cx_params::fj, fk), boundary records and their counts per tile
    size_t fj_cap = 0;
    uint64_t* fk = nullptr;
    size_t fk_cap = 0;
    uint4* bnd = nullptr;
    size_t bnd_cap = 0;
    uint32_t* bndn = nullptr;
    size_t bndn_cap = 0;
    uint32_t* torder = nullptr;
    size_t torder_cap = 0;
    uint8_t* hbytes = nullptr;         // fused emit: CPython set-order code per lattice point (valid for hash_xy's shape and origin)
    size_t hbytes_cap = 0;
    bool hbytes_valid = false;
    int64_t hbytes_n2 = 0, hbytes_o2 = 0;
    cx_task last_task = {};
    uint32_t last_flags = 0;
    int path = 0;                      // kernels of the last extraction: 0 generic, 1 staged, 2 fused, 3 tile emit
    bool records_valid = false;        // ctx->cells holds the cell records of the last extraction
    uint64_t* hash_xy = nullptr;       // CPython tuple-hash prefix per (i,j), for CX_DIAG_CPYTHON310
    size_t hash_xy_cap = 0;
    int64_t hash_xy_n0 = 0, hash_xy_n1 = 0, hash_xy_o0 = -1, hash_xy_o1 = -1;
    int64_t origin[3] = {0, 0, 0};
    int64_t corner_ref[3] = {0, 0, 0};   // > 0: the reference's corner for the Level-1 scales (cx_set_reference_corner)
    int64_t origin4[4] = {0, 0, 0, 0};
    cx_state4* s4 = nullptr;
    cx_state2* s2 = nullptr;
    cx_levels_state* lv = nullptr;
    int lv_current = -1;               // level of cx_extract3d_levels whose mesh the context's output buffers hold (-1: none)
    // Level-0 outputs
    cx_vrec* verts = nullptr;          // 8-byte vertex records {edge id, fp32 fraction}
    float4* verts_xyz = nullptr;       // {x, y, z, bits(edge id)} expanded from the records on request (cx_level0_expanded)
    size_t verts_xyz_cap = 0;
    uint4* cells = nullptr;
    int32_t* tris = nullptr;
    uint32_t vcap = 0, ccap = 0, tcap = 0;
    uint32_t* counters = nullptr;
    uint32_t* counters_host = nullptr;
    bool extracted = false;
    bool counts_fetched = false;       // ctx->counts holds the counters of the last 3-D extraction (cx_counts_get)
    cx_counts counts = {0, 0, 0, 0};
    cx_params last;
    // seeded selection (cx_select_seeded3d): triangle mask followed by vertex mask, valid until the next extraction
    uint8_t* tri_keep = nullptr;
    size_t keep_cap = 0;
    // scratch of cx_select_seeded3d_ex (bytes), kept between calls: map per sample, union-find, bitmap, flags, seeds, counters, end
    // points, visited set -- a selection allocated and freed them every time (0.8 of 2.9 ms on the 512^3 bench field)
    uint8_t* seed_buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t seed_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int seed_mode = 0;      // how the last seeded selection (3-D or 4-D) ran its end points: 0 sequential (the reference's shared visited set), 1 one thread per pair
    bool keep_valid = false;
    // Level-1
    cx_post_state* post = nullptr;
    bool post_valid = false;
    unsigned long long* stamps = nullptr;   // diagnostic stamps (cx_debug_stamps)
    size_t stamps_words = 0;
    // the context's own RCCL communicator (cx_rccl_comm_init, cx_halo.hip), or null
    void* rccl_comm = nullptr;
    bool rccl_owned = true;              // false: the communicator belongs to another context of this rank (cx_rccl_comm_share)
    int rccl_rank = 0, rccl_world = 1;
    // device -> host copies of mesh-sized buffers (cx_xfer.hip): two pinned staging buffers and their events
    void* xfer_stage[2] = {nullptr, nullptr};
    hipEvent_t xfer_ev[2] = {nullptr, nullptr};
    // timing
    struct evset { hipEvent_t e[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; };
    bool timing = false;
    int nevents = 0;
    evset events[256];
};

// THE place where a device buffer of a context is freed and allocated again.  A buffer is a (pointer, capacity) pair that always
// travels together through this function: a regrow branch cannot free a neighbour's pointer or leave a capacity describing a
// buffer that is gone (rounds 1 and 2 each had a stray hipFree in a hand-written regrow block).  Waits for the context's stream
// first (kernels enqueued on it may still use the old buffer).  `need` in elements; grows to exactly `need`.
template <typename T, typename C>
int cx_grow(cx_ctx* ctx, T*& ptr, C& cap, size_t need) {
    if ((size_t)cap >= need && ptr) return CX_OK;
    if (need == 0) return CX_OK;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && ptr) e = hipFree(ptr);
    ptr = nullptr; cap = 0;
    void* fresh = nullptr;
    if (e == hipSuccess) e = hipMalloc(&fresh, need * sizeof(T));
    if (e != hipSuccess) {
        ctx->err = std::string("device buffer (") + std::to_string(need * sizeof(T)) + " bytes): " + hipGetErrorString(e);
        return (e == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;
    }
    ptr = static_cast<T*>(fresh);
    cap = (C)need;
    return CX_OK;
}
template <typename T, typename C>
void cx_release(T*& ptr, C& cap) {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr; cap = 0;
}

// cx_api.hip
int cx_ensure_cell_records(cx_ctx* ctx);
int cx_ensure_hash_xy(cx_ctx* ctx, uint32_t flags);
int cx_level0_expanded(cx_ctx* ctx, float4** out);   // the records of the current extraction as float4 {x,y,z,id} (device, enqueued on the stream)
void cx_fill_value_params(cx_params& P, double value);
// cx_halo.hip
void cx_rccl_comm_free(cx_ctx* ctx);
// cx_xfer.hip
int cx_copy_to_host(cx_ctx* ctx, int nparts, void* const* dst, const void* const* src, const size_t* bytes);
int cx_copy_to_host1(cx_ctx* ctx, void* dst, const void* src, size_t bytes);
void cx_xfer_free(cx_ctx* ctx);
// cx_levels.hip
void cx_levels_free(cx_ctx* ctx);
void cx_levels_invalidate(cx_ctx* ctx);
// cx_post.hip
void cx_post_free(cx_ctx* ctx);
int cx_scan_u32(cx_ctx* ctx, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums_tmp, uint32_t* total_dev,
                unsigned long long* total64_dev = nullptr);   // total64: the total without the wrap at 2^32
// cx_api4d.hip
void cx_state4_free(cx_ctx* ctx);
// cx_contour2d.hip
void cx_state2_free(cx_ctx* ctx);
