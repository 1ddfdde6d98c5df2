// gemm_lp256.hip -- placeholder until the 256x256 deep-pipelined kernel lands.
#include "gemm_common.hpp"
namespace mi355 {
bool gemm_lp256_supports(const mi355_gemm_desc &, const void *, const void *, const void *) { return false; }
int32_t launch_gemm_lp256(mi355_ctx *ctx, hipStream_t, const mi355_gemm_desc &, const void *, const void *, void *)
{ return fail(ctx, MI355_E_UNSUPPORTED, "lp256 GEMM kernel not built"); }
}  // namespace mi355
