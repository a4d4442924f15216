// cx_march3d_fast.hip -- shape-specialised classify kernel (placeholder until the tiled kernel lands).
#include "cx_ctx.h"

bool cx_fast_classify_supported(const cx_params&) { return false; }
void cx_launch_classify_fast(const cx_params& P, hipStream_t s) { cx_launch_classify_generic(P, s); }
