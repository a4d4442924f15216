// cx_halo.hip -- the one exchange step of the slab-partitioned march in the C ABI.
//
// A volume split into slabs along array axis 0: rank r owns planes [i0, i1) and marches them together with ONE halo plane,
// the first plane of rank r+1 (voxels are independent given their corners; edge ids are global by formula).  The exchange is
// a send of the rank's first owned plane to rank r-1 and a receive of its halo plane from rank r+1, fused in one RCCL group
// and enqueued on the context's stream, so that cx_extract3d* calls issued afterwards are ordered behind it.
//
// RCCL is not linked: the communicator belongs to the caller, so the calls must go to the RCCL the CALLER's process has
// loaded (a PyTorch process carries its own copy).  The four entry points are looked up at the first call.
#include <dlfcn.h>

#include <string>

#include "cx_ctx.h"

namespace {
typedef int (*cx_nccl_group_fn)(void);
typedef int (*cx_nccl_p2p_fn)(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream);
struct cx_rccl {
    cx_nccl_group_fn group_start = nullptr, group_end = nullptr;
    cx_nccl_p2p_fn send = nullptr, recv = nullptr;   // (ncclSend takes a const buffer: same ABI)
    bool tried = false;
};
cx_rccl g_rccl;
const int CX_NCCL_FLOAT32 = 7;   // ncclFloat32 (rccl.h)

bool cx_rccl_resolve() {
    if (g_rccl.tried) return g_rccl.send != nullptr;
    g_rccl.tried = true;
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclSend")) {
        h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return false;
    }
    g_rccl.group_start = (cx_nccl_group_fn)dlsym(h, "ncclGroupStart");
    g_rccl.group_end = (cx_nccl_group_fn)dlsym(h, "ncclGroupEnd");
    g_rccl.send = (cx_nccl_p2p_fn)dlsym(h, "ncclSend");
    g_rccl.recv = (cx_nccl_p2p_fn)dlsym(h, "ncclRecv");
    if (!g_rccl.group_start || !g_rccl.group_end || !g_rccl.send || !g_rccl.recv) g_rccl.send = nullptr;
    return g_rccl.send != nullptr;
}
}   // namespace

extern "C" int cx_halo_exchange(cx_ctx* ctx, void* rccl_comm, int rank, int world, float* local_planes, int64_t n_own,
                                int64_t plane_samples) {
    if (!ctx || world < 1 || rank < 0 || rank >= world || n_own < 1 || plane_samples < 1) return CX_ERR_INVALID;
    if (world == 1) return CX_OK;   // nothing to exchange
    if (!rccl_comm || !local_planes) return CX_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "cx_halo_exchange: hipSetDevice failed"; return CX_ERR_HIP; }
    if (!cx_rccl_resolve()) { ctx->err = "cx_halo_exchange: no RCCL (ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd) in this process"; return CX_ERR_UNSUPPORTED; }
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
