// cx_xfer.hip -- device -> host copies of mesh-sized buffers into the CALLER's (pageable) memory.
//
// hipMemcpy into pageable memory stages through one internal buffer with one host thread: 16-17 GB/s measured on the MI355X
// boxes (540 MB of Level-1 mesh: 32 ms, three times the post-pass itself).  Here the copy goes through two pinned staging
// buffers of the context: the DMA of chunk i+1 (PCIe 5 x16: ~55 GB/s into pinned memory) runs while chunk i is copied into
// the destination by several host threads, which also spread the first-touch page faults of a freshly allocated destination.
// What the reference returns from get_points_and_triangles() are host arrays (tetrahedral.py:604-621), so the copy is part of
// the API path; callers that stay on the device use cx_level1_device_ptrs instead.
#include <algorithm>
#include <string>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "cx_ctx.h"

#define CXX_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return (e__ == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;                      \
        }                                                                                        \
    } while (0)

static const size_t CX_XFER_CHUNK = (size_t)32 << 20;   // bytes per staging buffer

void cx_xfer_free(cx_ctx* ctx) {
    for (int b = 0; b < 2; b++) {
        if (ctx->xfer_stage[b]) (void)hipHostFree(ctx->xfer_stage[b]);
        if (ctx->xfer_ev[b]) (void)hipEventDestroy(ctx->xfer_ev[b]);
        ctx->xfer_stage[b] = nullptr; ctx->xfer_ev[b] = nullptr;
    }
}

static int xfer_threads() {
    static const int n = [] {
        int want = 8;
        if (const char* e = getenv("CX_COPY_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) want = v; }
        const unsigned hc = std::thread::hardware_concurrency();
        if (hc && (unsigned)want > hc) want = (int)hc;
        return want < 1 ? 1 : want;
    }();
    return n;
}

// several device buffers -> host buffers, back to back through the same pipeline (one pipeline fill for a whole mesh).
// Ordered after everything enqueued on the context's stream; returns when the host buffers are complete.
// The host side: nt - 1 worker threads live for the duration of the call and sleep on a condition variable between pieces; piece k
// is announced through `ready`, every thread copies its page-aligned slice of it, `done[k]` counts the slices.
int cx_copy_to_host(cx_ctx* ctx, int nparts, void* const* dst, const void* const* src, const size_t* bytes) {
    size_t total = 0;
    for (int p = 0; p < nparts; p++) total += (dst[p] && src[p]) ? bytes[p] : 0;
    if (total == 0) { CXX_HIP(ctx, hipStreamSynchronize(ctx->stream)); return CX_OK; }
    if (total < ((size_t)4 << 20)) {   // small: the runtime's own path
        for (int p = 0; p < nparts; p++)
            if (dst[p] && src[p] && bytes[p]) CXX_HIP(ctx, hipMemcpyAsync(dst[p], src[p], bytes[p], hipMemcpyDeviceToHost, ctx->stream));
        CXX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return CX_OK;
    }
    for (int b = 0; b < 2; b++) {
        if (!ctx->xfer_stage[b]) CXX_HIP(ctx, hipHostMalloc(&ctx->xfer_stage[b], CX_XFER_CHUNK));
        if (!ctx->xfer_ev[b]) CXX_HIP(ctx, hipEventCreateWithFlags(&ctx->xfer_ev[b], hipEventDisableTiming));
    }
    struct piece { char* dst; const char* src; size_t n; };
    std::vector<piece> pieces;
    for (int p = 0; p < nparts; p++) {
        if (!dst[p] || !src[p]) continue;
        for (size_t off = 0; off < bytes[p]; off += CX_XFER_CHUNK)
            pieces.push_back({(char*)dst[p] + off, (const char*)src[p] + off, std::min(CX_XFER_CHUNK, bytes[p] - off)});
    }
    // host threads: no more than the bytes justify (a worker per 16 MB of mesh, at most xfer_threads()), and they SLEEP while they
    // wait (condition variables): with several ranks or contexts per node a pool of spinning threads per download would take the
    // cores the ranks' own host threads run on (round 3 spun in yield loops for the whole duration of every DMA chunk)
    int nt = std::max(1, std::min(xfer_threads(), (int)(total / ((size_t)16 << 20)) + 1));
    const size_t np = pieces.size();
    std::mutex mtx;
    std::condition_variable cv_ready, cv_done;
    long ready = -1;                                // pieces 0..ready sit in their staging buffers      (guarded by mtx)
    bool go = false, abort_flag = false;            //                                                   (guarded by mtx)
    std::vector<int> done(np, 0);                   // slices of piece k copied out                      (guarded by mtx)
    void* const stage0 = ctx->xfer_stage[0];
    void* const stage1 = ctx->xfer_stage[1];
    auto slice = [&](size_t k, int t, int nthreads) {
        const piece& pc = pieces[k];
        const size_t per = ((pc.n + (size_t)nthreads - 1) / (size_t)nthreads + 4095u) & ~(size_t)4095u;
        const size_t off = per * (size_t)t;
        if (off < pc.n) memcpy(pc.dst + off, (const char*)((k & 1) ? stage1 : stage0) + off, std::min(per, pc.n - off));
        {
            std::lock_guard<std::mutex> lk(mtx);
            done[k]++;
        }
        cv_done.notify_one();
    };
    // the workers wait for `go` before they look at nt: if the system refuses a thread, the copy runs with those it granted
    std::vector<std::thread> workers;
    workers.reserve((size_t)nt);
    try {
        for (int t = 1; t < nt; t++)
            workers.emplace_back([&, t] {
                int nthreads = 0;
                {
                    std::unique_lock<std::mutex> lk(mtx);
                    cv_ready.wait(lk, [&] { return go || abort_flag; });
                    if (abort_flag) return;
                    nthreads = nt;
                }
                for (size_t k = 0; k < np; k++) {
                    {
                        std::unique_lock<std::mutex> lk(mtx);
                        cv_ready.wait(lk, [&] { return ready >= (long)k || abort_flag; });
                        if (abort_flag) return;
                    }
                    slice(k, t, nthreads);
                }
            });
    } catch (...) {
    }
    {
        std::lock_guard<std::mutex> lk(mtx);
        nt = (int)workers.size() + 1;      // slices are cut for the threads that exist (read by the workers only after `go`)
        go = true;
    }
    cv_ready.notify_all();
    int rc = CX_OK;
    // piece i travels through staging buffer i & 1: its DMA is enqueued before piece i-1 is copied out on the host
    for (size_t i = 0; i <= np && rc == CX_OK; i++) {
        hipError_t e = hipSuccess;
        if (i < np) {
            e = hipMemcpyAsync(ctx->xfer_stage[i & 1], pieces[i].src, pieces[i].n, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(ctx->xfer_ev[i & 1], ctx->stream);
        }
        if (e == hipSuccess && i >= 1) {
            e = hipEventSynchronize(ctx->xfer_ev[(i - 1) & 1]);
            if (e == hipSuccess) {
                {
                    std::lock_guard<std::mutex> lk(mtx);
                    ready = (long)(i - 1);
                }
                cv_ready.notify_all();
                slice(i - 1, 0, nt);
                std::unique_lock<std::mutex> lk(mtx);
                cv_done.wait(lk, [&] { return done[i - 1] >= nt; });   // the buffer is free again
            }
        }
        if (e != hipSuccess) {
            ctx->err = std::string("cx_copy_to_host: ") + hipGetErrorString(e);
            rc = (e == hipErrorOutOfMemory) ? CX_ERR_NOMEM : CX_ERR_HIP;
        }
    }
    if (rc != CX_OK) {
        {
            std::lock_guard<std::mutex> lk(mtx);
            abort_flag = true;
        }
        cv_ready.notify_all();
    }
    for (auto& w : workers) w.join();
    return rc;
}
int cx_copy_to_host1(cx_ctx* ctx, void* dst, const void* src, size_t bytes) {
    void* d[1] = {dst}; const void* s[1] = {src}; size_t n[1] = {bytes};
    return cx_copy_to_host(ctx, 1, d, s, n);
}
