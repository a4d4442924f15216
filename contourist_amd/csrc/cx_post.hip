// cx_post.hip -- Level-1 mesh post-passes (placeholder).
#include "cx_ctx.h"

void cx_post_free(cx_ctx*) {}

extern "C" int cx_postprocess3d(cx_ctx* ctx, uint32_t, int64_t*) {
    if (ctx) ctx->err = "cx_postprocess3d: not built yet";
    return CX_ERR_UNSUPPORTED;
}
extern "C" int cx_level1_download(cx_ctx* ctx, double*, int32_t*) {
    if (ctx) ctx->err = "cx_level1_download: not built yet";
    return CX_ERR_UNSUPPORTED;
}
extern "C" int cx_surface_geometry(cx_ctx* ctx, double*, int64_t*, int32_t*, int64_t*, int) {
    if (ctx) ctx->err = "cx_surface_geometry: not built yet";
    return CX_ERR_UNSUPPORTED;
}
