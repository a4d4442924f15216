// cx_halo.hip -- the one exchange step of the slab-partitioned march in the C ABI.
//
// A volume split into slabs along array axis 0: rank r owns planes [i0, i1) and marches them together with ONE halo plane,
// the first plane of rank r+1 (voxels are independent given their corners; edge ids are global by formula).  The exchange is
// a send of the rank's first owned plane to rank r-1 and a receive of its halo plane from rank r+1, fused in one RCCL group
// and enqueued on the context's stream, so that cx_extract3d* calls issued afterwards are ordered behind it.
//
// RCCL is not linked.  The calls go to the RCCL copy the calling process has ALREADY loaded (a PyTorch process carries its
// own librccl, loaded RTLD_LOCAL: dlopen with RTLD_NOLOAD finds exactly that copy and never maps a second one -- a
// communicator handed to another copy of the library would be garbage there).  The communicator is either the caller's
// (rccl_comm) or one the context owns: cx_rccl_unique_id on one rank, the 128 bytes carried to the others by whatever the host
// uses (torch.distributed broadcast in contourist_amd/distributed.py), cx_rccl_comm_init on every rank.
//
// cx_slab_step is the whole per-volume step of one rank as ONE call: adopt the device buffer, exchange the halo, enqueue the
// extraction.  At 64 planes per rank (512^3 on 8 GPUs) an extraction is ~50 us of GPU time; a host that spends a Python
// batch_isend_irecv plus several FFI calls per step cannot keep the GPU fed.
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "cx_ctx.h"

namespace {
struct cx_nccl_uid { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value
typedef int (*cx_nccl_group_fn)(void);
typedef int (*cx_nccl_p2p_fn)(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream);
typedef int (*cx_nccl_uid_fn)(cx_nccl_uid* out);
typedef int (*cx_nccl_init_fn)(void** comm, int nranks, cx_nccl_uid id, int rank);
typedef int (*cx_nccl_destroy_fn)(void* comm);
struct cx_rccl {
    cx_nccl_group_fn group_start = nullptr, group_end = nullptr;
    cx_nccl_p2p_fn send = nullptr, recv = nullptr;   // (ncclSend takes a const buffer: same ABI)
    cx_nccl_uid_fn get_uid = nullptr;
    cx_nccl_init_fn init_rank = nullptr;
    cx_nccl_destroy_fn destroy = nullptr;
    bool ok = false;
};
cx_rccl g_rccl;
std::once_flag g_rccl_once;
const int CX_NCCL_FLOAT32 = 7;   // ncclFloat32 (rccl.h)

bool cx_rccl_resolve() {
    std::call_once(g_rccl_once, [] {
        // only objects that are already part of the process
        void* h = nullptr;
        if (dlsym(RTLD_DEFAULT, "ncclSend")) h = RTLD_DEFAULT;
        if (!h) h = dlopen("librccl.so.1", RTLD_NOLOAD | RTLD_NOW);
        if (!h) h = dlopen("librccl.so", RTLD_NOLOAD | RTLD_NOW);
        if (!h) return;
        g_rccl.group_start = (cx_nccl_group_fn)dlsym(h, "ncclGroupStart");
        g_rccl.group_end = (cx_nccl_group_fn)dlsym(h, "ncclGroupEnd");
        g_rccl.send = (cx_nccl_p2p_fn)dlsym(h, "ncclSend");
        g_rccl.recv = (cx_nccl_p2p_fn)dlsym(h, "ncclRecv");
        g_rccl.get_uid = (cx_nccl_uid_fn)dlsym(h, "ncclGetUniqueId");
        g_rccl.init_rank = (cx_nccl_init_fn)dlsym(h, "ncclCommInitRank");
        g_rccl.destroy = (cx_nccl_destroy_fn)dlsym(h, "ncclCommDestroy");
        g_rccl.ok = g_rccl.group_start && g_rccl.group_end && g_rccl.send && g_rccl.recv;
    });
    return g_rccl.ok;
}
const char* CX_NO_RCCL = "no RCCL loaded in this process (librccl.so with ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd)";
}   // namespace

// can this process run the C-side exchange at all?  Purely local (no collective): a host calls it on every rank and agrees on the
// answer BEFORE anybody enters the collective cx_rccl_comm_init -- a rank that found no RCCL would otherwise leave the others
// waiting inside ncclCommInitRank for ever.
extern "C" int cx_rccl_available(void) {
    if (!cx_rccl_resolve() || !g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.destroy) return CX_ERR_UNSUPPORTED;
    return CX_OK;
}

extern "C" int cx_rccl_unique_id(uint8_t* out128) {
    if (!out128) return CX_ERR_INVALID;
    if (!cx_rccl_resolve() || !g_rccl.get_uid) return CX_ERR_UNSUPPORTED;
    cx_nccl_uid id;
    if (g_rccl.get_uid(&id) != 0) return CX_ERR_HIP;
    memcpy(out128, id.internal, sizeof(id.internal));
    return CX_OK;
}

extern "C" int cx_rccl_comm_init(cx_ctx* ctx, const uint8_t* id128, int rank, int world) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return CX_ERR_INVALID;
    if (!cx_rccl_resolve() || !g_rccl.init_rank || !g_rccl.destroy) { ctx->err = CX_NO_RCCL; return CX_ERR_UNSUPPORTED; }
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "cx_rccl_comm_init: hipSetDevice failed"; return CX_ERR_HIP; }
    cx_rccl_comm_free(ctx);
    cx_nccl_uid id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void* comm = nullptr;
    const int rc = g_rccl.init_rank(&comm, world, id, rank);
    if (rc != 0 || !comm) { ctx->err = "cx_rccl_comm_init: ncclCommInitRank returned " + std::to_string(rc); return CX_ERR_HIP; }
    ctx->rccl_comm = comm;
    ctx->rccl_owned = true;
    ctx->rccl_rank = rank; ctx->rccl_world = world;
    return CX_OK;
}

// the contexts of ONE rank (its extractions in flight, one HIP stream each) share one communicator: every exchange is a group on
// the calling context's stream, issued by the host's single thread in volume order -- the same order on every rank.  (Round 3 gave
// every context a communicator of its own: exchanges of different communicators may then run side by side on the device, which
// RCCL only promises to survive when every rank issues them in one order AND the device can hold both at once; one communicator
// leaves nothing to promise.  RCCL orders the exchanges of a communicator behind each other across streams; the extractions
// behind them still overlap.)
extern "C" int cx_rccl_comm_share(cx_ctx* ctx, cx_ctx* owner) {
    if (!ctx || !owner || ctx == owner) return CX_ERR_INVALID;
    if (!owner->rccl_comm) { ctx->err = "cx_rccl_comm_share: the owner has no communicator"; return CX_ERR_STATE; }
    if (ctx->device != owner->device) { ctx->err = "cx_rccl_comm_share: contexts on different devices"; return CX_ERR_INVALID; }
    cx_rccl_comm_free(ctx);
    ctx->rccl_comm = owner->rccl_comm;
    ctx->rccl_owned = false;
    ctx->rccl_rank = owner->rccl_rank; ctx->rccl_world = owner->rccl_world;
    return CX_OK;
}

void cx_rccl_comm_free(cx_ctx* ctx) {
    if (ctx->rccl_comm && ctx->rccl_owned && g_rccl.destroy) (void)g_rccl.destroy(ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
    ctx->rccl_owned = true;
}
extern "C" int cx_rccl_comm_destroy(cx_ctx* ctx) {
    if (!ctx) return CX_ERR_INVALID;
    cx_rccl_comm_free(ctx);
    return CX_OK;
}

extern "C" int cx_halo_exchange(cx_ctx* ctx, void* rccl_comm, int rank, int world, float* local_planes, int64_t n_own,
                                int64_t plane_samples) {
    if (!ctx || world < 1 || rank < 0 || rank >= world || n_own < 1 || plane_samples < 1) return CX_ERR_INVALID;
    if (world == 1) return CX_OK;   // nothing to exchange
    if (!rccl_comm) rccl_comm = ctx->rccl_comm;   // the context's own communicator (cx_rccl_comm_init)
    if (!rccl_comm || !local_planes) return CX_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "cx_halo_exchange: hipSetDevice failed"; return CX_ERR_HIP; }
    if (!cx_rccl_resolve()) { ctx->err = std::string("cx_halo_exchange: ") + CX_NO_RCCL; return CX_ERR_UNSUPPORTED; }
    int rc = g_rccl.group_start();
    if (rc == 0 && rank > 0) rc = g_rccl.send(local_planes, (size_t)plane_samples, CX_NCCL_FLOAT32, rank - 1, rccl_comm, ctx->stream);
    if (rc == 0 && rank + 1 < world)
        rc = g_rccl.recv(local_planes + (size_t)n_own * (size_t)plane_samples, (size_t)plane_samples, CX_NCCL_FLOAT32, rank + 1, rccl_comm, ctx->stream);
    const int rc_end = g_rccl.group_end();
    if (rc != 0 || rc_end != 0) {
        ctx->err = "cx_halo_exchange: RCCL returned " + std::to_string(rc != 0 ? rc : rc_end);
        return CX_ERR_HIP;
    }
    return CX_OK;
}

// one rank's whole step for one volume: local_planes = its n_own planes followed (unless it is the last rank) by room for the
// halo plane.  Halo exchange on the context's stream with the context's own communicator, then the extraction behind it.
extern "C" int cx_slab_step(cx_ctx* ctx, float* local_planes, int64_t n_own, int64_t n1, int64_t n2, int rank, int world, double value,
                            uint32_t flags) {
    if (!ctx || !local_planes || n_own < 1 || world < 1 || rank < 0 || rank >= world) return CX_ERR_INVALID;
    const int64_t n0 = n_own + ((rank + 1 < world) ? 1 : 0);
    int rc = cx_grid_adopt_device(ctx, local_planes, n0, n1, n2);
    if (rc) return rc;
    if (world > 1) {
        if (!ctx->rccl_comm) { ctx->err = "cx_slab_step: no communicator (cx_rccl_comm_init)"; return CX_ERR_STATE; }
        if ((rc = cx_halo_exchange(ctx, nullptr, rank, world, local_planes, n_own, n1 * n2))) return rc;
    }
    return cx_extract3d_async(ctx, value, flags);
}
