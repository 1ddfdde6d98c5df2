// Host stand-ins for the few COMPUTE entry points bench.py's N > 1 control flow touches (TEST INFRASTRUCTURE ONLY, like the rest of
// tests/fake_hip/): with the library's real host runtime (runtime.cpp, pool.cpp, comm.cpp) on the fake HIP runtime they let
// tests/test_bench_cpu.py run `bench.py --gpus 2` to completion on a box without a device -- launcher rendezvous, the native
// barrier / max-over-ranks, the C4 exchange (all-reduce + all-gather + combine), the one JSON line.  "Device memory" is host
// memory here, so the stand-ins are plain loops; the GEMM does nothing (its figures mean nothing in such a run).  Every other entry
// point of the header is a generated weak stub that returns MI355_E_UNSUPPORTED (tests/test_bench_cpu.py).
#include "../../cubecl_amd/csrc/internal.hpp"

#include <cmath>

namespace {
uint32_t key_of(float v)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;
    if (u == 0x80000000u) u = 0;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
}  // namespace

MI355_API int32_t mi355_fill_uniform(mi355_ctx *ctx, mi355_stream, void *dst, int32_t dtype, uint64_t n, uint64_t, uint64_t tensor, float lo, float hi)
{
    MI355_REQUIRE_CTX(ctx);
    if (dtype == MI355_DTYPE_F32) {
        float *p = static_cast<float *>(dst);
        for (uint64_t i = 0; i < n; ++i) p[i] = lo + (hi - lo) * (float)((i * 2654435761ull + tensor * 97) % 1000) / 1000.0f;
    } else {
        memset(dst, 0, n * mi355::dtype_size(dtype));      // 16-bit operands of the GEMM stand-in: never read
    }
    return MI355_OK;
}
MI355_API int32_t mi355_gemm_select(mi355_ctx *ctx, const mi355_gemm_desc *, int32_t *out_algo)
{
    if (!ctx || !out_algo) return MI355_E_INVALID_ARGUMENT;
    *out_algo = MI355_GEMM_ALGO_LP_256W4;
    return MI355_OK;
}
MI355_API int32_t mi355_gemm(mi355_ctx *ctx, mi355_stream, const mi355_gemm_desc *, const void *, const void *, void *) { return ctx ? MI355_OK : MI355_E_INVALID_ARGUMENT; }
MI355_API int32_t mi355_probe_clock(mi355_ctx *ctx, mi355_stream, uint64_t *) { return ctx ? MI355_OK : MI355_E_INVALID_ARGUMENT; }
MI355_API int32_t mi355_reduce_workspace_bytes(mi355_ctx *, uint64_t, uint64_t *out_bytes) { *out_bytes = 4096 * 16 + 256; return MI355_OK; }

MI355_API int32_t mi355_sum_argmax_f32(mi355_ctx *ctx, mi355_stream, const float *in, uint64_t n, float *out_sum, float *out_val, uint64_t *out_idx,
                                       void *, uint64_t)
{
    MI355_REQUIRE_CTX(ctx);
    double s = 0.0;
    uint64_t best = 0;
    uint32_t bk = 0;
    for (uint64_t i = 0; i < n; ++i) {
        s += in[i];
        const uint32_t k = key_of(in[i]);
        if (k > bk) { bk = k; best = i; }
    }
    if (out_sum) *out_sum = (float)s;
    if (out_val) *out_val = n ? in[best] : -INFINITY;
    if (out_idx) *out_idx = n ? best : 0;
    return MI355_OK;
}
MI355_API int32_t mi355_reduce_sum_f32(mi355_ctx *ctx, mi355_stream s, const float *in, uint64_t n, float *out, void *ws, uint64_t wsb)
{ return mi355_sum_argmax_f32(ctx, s, in, n, out, nullptr, nullptr, ws, wsb); }
MI355_API int32_t mi355_argmax_f32(mi355_ctx *ctx, mi355_stream s, const float *in, uint64_t n, float *out_val, uint64_t *out_idx, void *ws, uint64_t wsb)
{ return mi355_sum_argmax_f32(ctx, s, in, n, nullptr, out_val, out_idx, ws, wsb); }

// the rule of reduce.hip's argmax_combine_kernel, restated on the host
MI355_API int32_t mi355_argmax_combine_f32(mi355_ctx *ctx, mi355_stream, const void *records, uint32_t count, const uint64_t *index_base, float *out_val,
                                           uint64_t *out_idx)
{
    MI355_REQUIRE_CTX(ctx);
    const uint32_t *rec = static_cast<const uint32_t *>(records);
    uint32_t key = 0, bits = 0xFF800000u;
    uint64_t idx = ~0ull;
    for (uint32_t r = 0; r < count; ++r) {
        const uint64_t li = ((uint64_t)rec[r * 4 + 3] << 32) | rec[r * 4 + 2];
        if (li == ~0ull) continue;
        float v;
        memcpy(&v, &rec[r * 4], 4);
        const uint32_t k = key_of(v);
        const uint64_t gi = (index_base ? index_base[r] : 0) + li;
        if (k > key || (k == key && gi < idx)) { key = k; idx = gi; bits = rec[r * 4]; }
    }
    if (out_val) memcpy(out_val, &bits, 4);
    if (out_idx) *out_idx = (idx == ~0ull) ? 0 : idx;
    return MI355_OK;
}

// ... and of its SUMS form: the records' second words added in rank order from +0.0, in f32
MI355_API int32_t mi355_sum_argmax_combine_f32(mi355_ctx *ctx, mi355_stream s, const void *records, uint32_t count, const uint64_t *index_base,
                                               float *out_sum, float *out_val, uint64_t *out_idx)
{
    MI355_REQUIRE_CTX(ctx);
    const uint32_t *rec = static_cast<const uint32_t *>(records);
    float total = 0.f;
    for (uint32_t r = 0; r < count; ++r) {
        float p;
        memcpy(&p, &rec[r * 4 + 1], 4);
        total += p;
    }
    if (out_sum) *out_sum = total;
    if (out_val || out_idx) return mi355_argmax_combine_f32(ctx, s, records, count, index_base, out_val, out_idx);
    return MI355_OK;
}

// all-gather + fence + combine of the sharded sum + argmax in one call (the product's lives in reduce.hip beside the combine kernel)
MI355_API int32_t mi355_sum_argmax_exchange(mi355_ctx *ctx, mi355_comm *comm, mi355_stream stream, const void *record, void *gathered,
                                            const uint64_t *index_base, float *out_sum, float *out_val, uint64_t *out_idx)
{
    MI355_REQUIRE_CTX(ctx);
    if (!comm) return mi355::fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_sum_argmax_exchange: communicator is NULL (call mi355_comm_init)");
    if (!record || !gathered) return mi355::fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_sum_argmax_exchange: NULL record buffer");
    int32_t rc = mi355_all_gather(ctx, comm, stream, record, gathered, 2, MI355_DTYPE_U64);
    if (rc == MI355_OK) rc = mi355_sync_collective(ctx, stream);
    if (rc == MI355_OK) rc = mi355_sum_argmax_combine_f32(ctx, stream, gathered, (uint32_t)mi355::comm_world_size(comm), index_base, out_sum, out_val, out_idx);
    return rc;
}
