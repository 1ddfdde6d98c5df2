// runtime.cpp -- host side of libmi355cube.so: device contexts, storage, streams/events, IO,
// generic module launch, error queue, profiling.  Mirrors the operations of the reference's
// HipServer / HipContext / GpuStorage (crates/cubecl-hip/src/compute/*) one to one; see
// include/mi355cube.h for the per-function citations.
#include "internal.hpp"

#include <algorithm>

namespace {
std::mutex g_global_mutex;
std::string g_global_error;

void set_global_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lock(g_global_mutex);
    g_global_error = buf;
}

uint64_t next_pow2(uint64_t v)
{
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
}  // namespace

namespace mi355 {

int32_t fail(mi355_ctx *ctx, int32_t code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    else set_global_error("%s", buf);
    return code;
}

void queue_error(mi355_ctx *ctx, int32_t code, uint64_t requested, uint64_t max, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx->last_error = buf;
    ctx->errors.push_back({code, requested, max, std::string(buf)});
}

int32_t map_hip_error(hipError_t e)
{
    switch (e) {
    case hipSuccess: return MI355_OK;
    case hipErrorOutOfMemory: return MI355_E_OUT_OF_MEMORY;
    case hipErrorInvalidValue: return MI355_E_INVALID_ARGUMENT;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice: return MI355_E_NO_DEVICE;
    case hipErrorNotFound: return MI355_E_NOT_FOUND;
    case hipErrorLaunchFailure:
    case hipErrorLaunchOutOfResources: return MI355_E_LAUNCH;
    default: return MI355_E_EXECUTION;
    }
}

void check_launch(mi355_ctx *ctx, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        queue_error(ctx, MI355_E_LAUNCH, 0, 0, "%s: launch failed: %s", what, hipGetErrorString(e));
}

}  // namespace mi355

using namespace mi355;

static int32_t report_queue(mi355_ctx *ctx)
{
    if (ctx->errors.empty()) return MI355_OK;
    ctx->last_error = "server unhealthy: " + std::to_string(ctx->errors.size()) +
                      " queued error(s); first: " + ctx->errors.front().message;
    return MI355_E_SERVER_UNHEALTHY;
}

/* =================================== Runtime ============================================= */

MI355_API int32_t mi355_abi_version(void) { return MI355_ABI_VERSION; }

MI355_API const char *mi355_last_global_error(void)
{
    std::lock_guard<std::mutex> lock(g_global_mutex);
    static thread_local std::string copy;
    copy = g_global_error;
    return copy.c_str();
}

MI355_API const char *mi355_last_error(mi355_ctx *ctx)
{
    if (!ctx) return mi355_last_global_error();
    return ctx->last_error.c_str();
}

MI355_API int32_t mi355_device_count(int32_t *out_count)
{
    if (!out_count) return MI355_E_INVALID_ARGUMENT;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        return fail(nullptr, MI355_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *out_count = n;
    return MI355_OK;
}

static void add_mma(mi355_device_props_t &p, uint32_t m, uint32_t n, uint32_t k, int32_t ab, int32_t cd)
{
    if (p.num_mma_configs >= 16) return;
    mi355_mma_config &c = p.mma_configs[p.num_mma_configs++];
    c.m = m; c.n = n; c.k = k; c.a_type = ab; c.b_type = ab; c.cd_type = cd;
}


MI355_API int32_t mi355_ctx_create(int32_t device_index, mi355_ctx **out_ctx)
{
    if (!out_ctx) return MI355_E_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, MI355_E_NO_DEVICE, "no HIP device visible (hipGetDeviceCount: %s, count %d)",
                    hipGetErrorString(e), n);
    if (device_index < 0 || device_index >= n)
        return fail(nullptr, MI355_E_NO_DEVICE, "device index %d out of range (%d devices)", device_index, n);

    hipDeviceProp_t dp;
    e = hipGetDeviceProperties(&dp, device_index);
    if (e != hipSuccess)
        return fail(nullptr, MI355_E_NO_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    // The reference asserts warpSize against its arch table and dies on gfx950
    // (crates/cubecl-hip/src/runtime.rs:118-124).  This library ships gfx950 code objects only.
    if (dp.warpSize != 64 || strncmp(dp.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MI355_E_NO_DEVICE,
                    "device %d is %s (wave %d); libmi355cube only carries gfx950 / wave64 code objects",
                    device_index, dp.gcnArchName, dp.warpSize);

    e = hipSetDevice(device_index);
    if (e != hipSuccess) return fail(nullptr, MI355_E_NO_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));

    mi355_ctx *ctx = new mi355_ctx();
    ctx->device = device_index;
    if ((e = hipStreamCreateWithFlags(&ctx->compute_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->fence_a, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->fence_b, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->fence_c, hipEventDisableTiming)) != hipSuccess) {
        int32_t rc = fail(nullptr, map_hip_error(e), "stream/event creation: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }

    mi355_device_props_t &p = ctx->props;
    memset(&p, 0, sizeof p);
    p.abi_version = MI355_ABI_VERSION;
    p.device_index = device_index;
    snprintf(p.name, sizeof p.name, "%s", dp.name);
    snprintf(p.gcn_arch_name, sizeof p.gcn_arch_name, "%s", dp.gcnArchName);
    snprintf(p.fingerprint, sizeof p.fingerprint, "mi355-aot_%s", dp.gcnArchName);
    p.load_width_bits = 128;
    p.plane_size_min = 64;
    p.plane_size_max = 64;
    p.max_bindings = 1024;
    // The reference copies sharedMemPerBlock (runtime.rs:103,170), which HIP reports as 64 KiB;
    // a gfx950 workgroup may in fact use the CU's whole 160 KiB when the kernel opts in.
    p.max_shared_memory_size = std::max<uint64_t>(dp.sharedMemPerBlock, dp.maxSharedMemoryPerMultiProcessor);
    p.max_cube_count[0] = (uint32_t)dp.maxGridSize[0];
    p.max_cube_count[1] = (uint32_t)dp.maxGridSize[1];
    p.max_cube_count[2] = (uint32_t)dp.maxGridSize[2];
    p.max_units_per_cube = (uint32_t)dp.maxThreadsPerBlock;
    p.max_cube_dim[0] = (uint32_t)dp.maxThreadsDim[0];
    p.max_cube_dim[1] = (uint32_t)dp.maxThreadsDim[1];
    p.max_cube_dim[2] = (uint32_t)dp.maxThreadsDim[2];
    p.num_streaming_multiprocessors = (uint32_t)dp.multiProcessorCount;
    p.num_tensor_cores = 4;
    p.min_tensor_cores_dim = 4;
    p.num_xcd = 8;
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
    p.total_memory = total_b ? total_b : dp.totalGlobalMem;
    p.max_page_size = p.total_memory / 4;
    p.mem_alignment = std::max<uint64_t>(32, std::max<uint64_t>(dp.textureAlignment, dp.surfaceAlignment));
    p.clock_khz = (uint32_t)dp.clockRate;
    p.memory_clock_khz = (uint32_t)dp.memoryClockRate;
    p.memory_bus_width_bits = (uint32_t)dp.memoryBusWidth;
    p.l2_cache_bytes = (uint32_t)dp.l2CacheSize;
    p.plane_ops = 1;
    p.plane_non_uniform = 1;
    p.timing_method_device = 1;
    p.server_comm_enabled = rccl_available() ? 1 : 0;
    // gfx950 MFMA shapes this backend executes + the 16x16x16 fragment the
    // reference's cmma known-answer tests ask for (run as one zero-padded 16x16x32 MFMA).
    add_mma(p, 32, 32, 16, MI355_DTYPE_BF16, MI355_DTYPE_F32);
    add_mma(p, 16, 16, 32, MI355_DTYPE_BF16, MI355_DTYPE_F32);
    add_mma(p, 32, 32, 16, MI355_DTYPE_F16, MI355_DTYPE_F32);
    add_mma(p, 16, 16, 32, MI355_DTYPE_F16, MI355_DTYPE_F32);
    add_mma(p, 16, 16, 16, MI355_DTYPE_F16, MI355_DTYPE_F32);
    add_mma(p, 16, 16, 16, MI355_DTYPE_BF16, MI355_DTYPE_F32);
    add_mma(p, 32, 32, 2, MI355_DTYPE_F32, MI355_DTYPE_F32);
    add_mma(p, 16, 16, 4, MI355_DTYPE_F32, MI355_DTYPE_F32);
    add_mma(p, 16, 16, 8, MI355_DTYPE_F32, MI355_DTYPE_F32);
    // OCP FP8 on v_mfma_f32_32x32x64_f8f6f4 (unscaled) and its block-scaled form (ue8m0 per 32 k: 2 scales per step)
    add_mma(p, 32, 32, 64, MI355_DTYPE_F8E4M3, MI355_DTYPE_F32);
    add_mma(p, 32, 32, 64, MI355_DTYPE_F8E5M2, MI355_DTYPE_F32);
    {
        const int32_t pairs[5][2] = {{MI355_DTYPE_F8E4M3, MI355_DTYPE_F8E4M3}, {MI355_DTYPE_F8E5M2, MI355_DTYPE_F8E5M2},
                                     {MI355_DTYPE_F8E4M3, MI355_DTYPE_F8E5M2}, {MI355_DTYPE_F8E5M2, MI355_DTYPE_F8E4M3},
                                     {MI355_DTYPE_F4E2M1X2, MI355_DTYPE_F4E2M1X2}};
        for (const auto &pr : pairs) {
            mi355_scaled_mma_config &c = p.scaled_mma_configs[p.num_scaled_mma_configs++];
            c.m = 32; c.n = 32; c.k = 64; c.a_type = pr[0]; c.b_type = pr[1]; c.cd_type = MI355_DTYPE_F32;
            c.scales_type = MI355_DTYPE_UE8M0; c.scales_factor = 2;
        }
    }

    // register_supported_types (crates/cubecl-cpp/src/shared/base.rs:322-375), same table, same usages: what generated
    // kernels may use on this device.  (The matrix types of features.matmul are the MMA lists above.)
    p.address_types = MI355_ADDRESS_TYPE_U32 | MI355_ADDRESS_TYPE_U64;
    const int32_t full[] = {MI355_DTYPE_INDEX, MI355_DTYPE_U8, MI355_DTYPE_U16, MI355_DTYPE_U32, MI355_DTYPE_U64, MI355_DTYPE_I8,
                            MI355_DTYPE_I16, MI355_DTYPE_I32, MI355_DTYPE_I64, MI355_DTYPE_BF16, MI355_DTYPE_F16, MI355_DTYPE_F32,
                            MI355_DTYPE_FLEX32, MI355_DTYPE_F64, MI355_DTYPE_BOOL};
    for (int32_t t : full) p.type_usage[p.num_type_usage++] = {t, MI355_TYPE_USAGE_ALL};
    for (int32_t t : {MI355_DTYPE_F8E4M3, MI355_DTYPE_F8E5M2})
        p.type_usage[p.num_type_usage++] = {t, MI355_TYPE_USAGE_CONVERSION | MI355_TYPE_USAGE_BUFFER};
    // atomics: every operation on 32-bit integers; add / load-store / exchange on i64, u64 and f32 (base.rs:364-372)
    for (int32_t t : {MI355_DTYPE_I32, MI355_DTYPE_I64, MI355_DTYPE_U32, MI355_DTYPE_U64, MI355_DTYPE_F32}) {
        const bool full32 = t == MI355_DTYPE_I32 || t == MI355_DTYPE_U32;
        p.atomic_usage[p.num_atomic_usage++] = {t, full32 ? (uint32_t)MI355_ATOMIC_ALL
                                                          : (uint32_t)(MI355_ATOMIC_ADD | MI355_ATOMIC_LOAD_STORE | MI355_ATOMIC_EXCHANGE)};
    }
    // TargetProperties.mma for MFMA (the reference's HIP values are RDNA WMMA with const_plane_size 32, runtime.rs:282-304)
    p.mma_properties = {32, 64, MI355_LAYOUT_ROW_MAJOR, MI355_LAYOUT_COL_MAJOR, MI355_LAYOUT_COL_MAJOR, 1, 1, 1, 128, 4};

    *out_ctx = ctx;
    return MI355_OK;
}

MI355_API int32_t mi355_ctx_destroy(mi355_ctx *ctx)
{
    if (!ctx) return MI355_OK;
    { MI355_LOCK_CTX(ctx); }               // let a call still running on another thread finish
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    for (void *p : ctx->pending_free) hipFree(p);
    pool_destroy(ctx);
    for (auto &s : ctx->profiles)
        if (s.live) { hipEventDestroy(s.start); hipEventDestroy(s.stop); }
    if (ctx->fence_a) hipEventDestroy(ctx->fence_a);
    if (ctx->fence_b) hipEventDestroy(ctx->fence_b);
    if (ctx->fence_c) hipEventDestroy(ctx->fence_c);
    if (ctx->ticket_buf) hipFree(ctx->ticket_buf);
    for (auto &kv : ctx->scratch) hipFree(kv.second.first);
    for (void *p : ctx->scratch_retired) hipFree(p);
    if (ctx->compute_stream) hipStreamDestroy(ctx->compute_stream);
    if (ctx->comm_stream) hipStreamDestroy(ctx->comm_stream);
    delete ctx;
    return MI355_OK;
}

MI355_API int32_t mi355_device_props(mi355_ctx *ctx, mi355_device_props_t *out_props)
{
    if (!ctx || !out_props) return MI355_E_INVALID_ARGUMENT;
    *out_props = ctx->props;
    return MI355_OK;
}

MI355_API int32_t mi355_error_count(mi355_ctx *ctx, int32_t *out_count)
{
    if (!ctx || !out_count) return MI355_E_INVALID_ARGUMENT;
    MI355_LOCK_CTX(ctx);
    *out_count = (int32_t)ctx->errors.size();
    return MI355_OK;
}

MI355_API int32_t mi355_error_pop(mi355_ctx *ctx, int32_t *out_code, uint64_t *out_requested,
                                  uint64_t *out_max, char *msg, size_t msg_capacity)
{
    if (!ctx) return MI355_E_INVALID_ARGUMENT;
    MI355_LOCK_CTX(ctx);
    if (ctx->errors.empty()) return MI355_E_NOT_FOUND;
    const mi355_queued_error &err = ctx->errors.front();
    if (out_code) *out_code = err.code;
    if (out_requested) *out_requested = err.requested;
    if (out_max) *out_max = err.max;
    if (msg && msg_capacity) snprintf(msg, msg_capacity, "%s", err.message.c_str());
    ctx->errors.pop_front();
    return MI355_OK;
}

/* =================================== Storage ============================================= */

MI355_API int32_t mi355_alloc(mi355_ctx *ctx, uint64_t bytes, void **out_dptr)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_dptr) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_alloc: out_dptr is NULL");
    *out_dptr = nullptr;
    if (bytes == 0) return MI355_OK;
    if (bytes > ctx->props.max_page_size)
        return fail(ctx, MI355_E_BUFFER_TOO_BIG, "allocation of %llu bytes exceeds max_page_size %llu",
                    (unsigned long long)bytes, (unsigned long long)ctx->props.max_page_size);
    hipError_t e = hipMalloc(out_dptr, bytes);
    if (e == hipErrorOutOfMemory) {
        // OOM => cleanup + one retry (crates/cubecl-hip/src/compute/command.rs:142-161)
        hipGetLastError();
        hipDeviceSynchronize();
        for (void *p : ctx->pending_free) hipFree(p);
        ctx->pending_free.clear();
        e = hipMalloc(out_dptr, bytes);
    }
    if (e != hipSuccess) {
        hipGetLastError();
        *out_dptr = nullptr;
        return fail(ctx, e == hipErrorOutOfMemory ? MI355_E_OUT_OF_MEMORY : map_hip_error(e),
                    "hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    return MI355_OK;
}

MI355_API int32_t mi355_free(mi355_ctx *ctx, void *dptr)
{
    if (!ctx) return MI355_E_INVALID_ARGUMENT;
    MI355_LOCK_CTX(ctx);
    if (dptr) ctx->pending_free.push_back(dptr);
    return MI355_OK;
}

MI355_API int32_t mi355_mem_info(mi355_ctx *ctx, uint64_t *out_free, uint64_t *out_total)
{
    MI355_REQUIRE_CTX(ctx);
    size_t f = 0, t = 0;
    MI355_HIP(ctx, hipMemGetInfo(&f, &t));
    if (out_free) *out_free = f;
    if (out_total) *out_total = t;
    return MI355_OK;
}

MI355_API int32_t mi355_pitched_row_bytes(mi355_ctx *ctx, uint64_t width_bytes, uint64_t *out_pitch)
{
    if (!ctx || !out_pitch) return MI355_E_INVALID_ARGUMENT;
    if (width_bytes == 0) { *out_pitch = 0; return MI355_OK; }
    // optimal_align (memory_pool/handle.rs:255-263): next_pow2(width) clamped to [16, alignment]
    uint64_t align = std::min<uint64_t>(std::max<uint64_t>(next_pow2(width_bytes), 16), ctx->props.mem_alignment);
    *out_pitch = (width_bytes + align - 1) / align * align;
    return MI355_OK;
}

MI355_API int32_t mi355_pinned_alloc(mi355_ctx *ctx, uint64_t bytes, void **out_hptr)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_hptr) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_pinned_alloc: out_hptr is NULL");
    *out_hptr = nullptr;
    if (bytes == 0) return MI355_OK;
    MI355_HIP(ctx, hipHostMalloc(out_hptr, bytes, hipHostMallocMapped));
    return MI355_OK;
}

MI355_API int32_t mi355_pinned_free(mi355_ctx *ctx, void *hptr)
{
    MI355_REQUIRE_CTX(ctx);
    if (hptr) MI355_HIP(ctx, hipHostFree(hptr));
    return MI355_OK;
}

/* =================================== Streams / events ==================================== */

MI355_API int32_t mi355_stream_create(mi355_ctx *ctx, mi355_stream *out_stream)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_stream) return fail(ctx, MI355_E_INVALID_ARGUMENT, "out_stream is NULL");
    hipStream_t s;
    MI355_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out_stream = s;
    return MI355_OK;
}

MI355_API int32_t mi355_stream_destroy(mi355_ctx *ctx, mi355_stream stream)
{
    MI355_REQUIRE_CTX(ctx);
    if (!stream) return MI355_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (s == ctx->compute_stream || s == ctx->comm_stream)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_stream_destroy: the context's own streams die with the context");
    // What the library keeps per stream goes with it: the scratch buffers (split-K slabs, re-laid-out operands, partial sums) and
    // the ticket slot.  A service that creates a stream per request would otherwise leak both -- and the driver hands a dead
    // stream's handle to the next hipStreamCreate, which would then inherit them.  The stream's work is finished first; scratch a
    // live graph replays against is retired (freed when the last such graph dies), not freed.
    MI355_HIP(ctx, hipStreamSynchronize(s));
    if (ctx->inline_stream == s) { ctx->inline_stream = nullptr; ctx->inline_dirty = false; }   // its inline collectives have completed
    for (auto it = ctx->scratch.begin(); it != ctx->scratch.end();) {
        if (it->first.first != s) { ++it; continue; }
        void *p = it->second.first;
        if (p) {
            const auto pin = ctx->scratch_refs.find(p);
            if (pin != ctx->scratch_refs.end() && pin->second > 0) ctx->scratch_retired.insert(p);
            else if (ctx->capture_scratch.count(p)) ctx->scratch_retired.insert(p);   // handed out inside the open window
            else hipFree(p);
        }
        it = ctx->scratch.erase(it);
    }
    const auto slot = ctx->ticket_slots.find(s);
    if (slot != ctx->ticket_slots.end()) {
        const auto pin = ctx->ticket_refs.find(slot->second);
        if ((pin != ctx->ticket_refs.end() && pin->second > 0) || ctx->capture_tickets.count(slot->second)) ctx->ticket_retired.insert(slot->second);
        else ctx->ticket_free.push_back(slot->second);
        ctx->ticket_slots.erase(slot);
    }
    MI355_HIP(ctx, hipStreamDestroy(s));
    return MI355_OK;
}

MI355_API int32_t mi355_default_stream(mi355_ctx *ctx, mi355_stream *out_stream)
{
    if (!ctx || !out_stream) return MI355_E_INVALID_ARGUMENT;
    *out_stream = ctx->compute_stream;
    return MI355_OK;
}

MI355_API int32_t mi355_comm_stream(mi355_ctx *ctx, mi355_stream *out_stream)
{
    if (!ctx || !out_stream) return MI355_E_INVALID_ARGUMENT;
    *out_stream = ctx->comm_stream;
    return MI355_OK;
}

MI355_API int32_t mi355_event_create(mi355_ctx *ctx, mi355_event *out_event)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_event) return fail(ctx, MI355_E_INVALID_ARGUMENT, "out_event is NULL");
    hipEvent_t ev;
    MI355_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDefault));
    *out_event = ev;
    return MI355_OK;
}

MI355_API int32_t mi355_event_destroy(mi355_ctx *ctx, mi355_event event)
{
    MI355_REQUIRE_CTX(ctx);
    ctx->captured_events.erase(event);
    if (event) MI355_HIP(ctx, hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return MI355_OK;
}

MI355_API int32_t mi355_event_record(mi355_ctx *ctx, mi355_event event, mi355_stream stream)
{
    MI355_REQUIRE_CTX(ctx);
    if (!event) return fail(ctx, MI355_E_INVALID_ARGUMENT, "event is NULL");
    MI355_HIP(ctx, hipEventRecord(reinterpret_cast<hipEvent_t>(event), stream_of(ctx, stream)));
    // recorded on the stream under capture: the event is a graph node now, not something the host can wait for
    if (ctx->capturing && stream_of(ctx, stream) == ctx->capture_stream) ctx->captured_events.insert(event);
    else ctx->captured_events.erase(event);
    return MI355_OK;
}

MI355_API int32_t mi355_stream_wait_event(mi355_ctx *ctx, mi355_stream stream, mi355_event event)
{
    MI355_REQUIRE_CTX(ctx);
    if (!event) return fail(ctx, MI355_E_INVALID_ARGUMENT, "event is NULL");
    MI355_HIP(ctx, hipStreamWaitEvent(stream_of(ctx, stream), reinterpret_cast<hipEvent_t>(event), 0));
    return MI355_OK;
}

MI355_API int32_t mi355_event_sync(mi355_ctx *ctx, mi355_event event)
{
    MI355_REQUIRE_CTX(ctx);
    if (!event) return fail(ctx, MI355_E_INVALID_ARGUMENT, "event is NULL");
    // only an event recorded INSIDE the open window cannot be waited for by the host; the other lanes run real work
    if (ctx->capturing && ctx->captured_events.count(event))
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_event_sync: not inside a graph capture window");
    MI355_HIP(ctx, hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return MI355_OK;
}

MI355_API int32_t mi355_event_elapsed_ms(mi355_ctx *ctx, mi355_event start, mi355_event stop, float *out_ms)
{
    MI355_REQUIRE_CTX(ctx);
    if (!start || !stop || !out_ms) return fail(ctx, MI355_E_INVALID_ARGUMENT, "NULL argument");
    MI355_HIP(ctx, hipEventElapsedTime(out_ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return MI355_OK;
}

/* =================================== IO =================================================== */

MI355_API int32_t mi355_write(mi355_ctx *ctx, mi355_stream stream, void *dst_dptr, const void *src_host,
                              uint64_t bytes)
{
    MI355_REQUIRE_CTX(ctx);
    if (bytes == 0) return MI355_OK;  // empty tensors skip copies (command.rs:363-369)
    if (!dst_dptr || !src_host) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_write: NULL pointer");
    MI355_HIP(ctx, hipMemcpyAsync(dst_dptr, src_host, bytes, hipMemcpyHostToDevice, stream_of(ctx, stream)));
    return MI355_OK;
}

MI355_API int32_t mi355_read_async(mi355_ctx *ctx, mi355_stream stream, void *dst_host, const void *src_dptr,
                                   uint64_t bytes)
{
    MI355_REQUIRE_CTX(ctx);
    if (bytes == 0) return MI355_OK;
    if (!dst_host || !src_dptr) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_read: NULL pointer");
    if (ctx->capturing && stream_of(ctx, stream) == ctx->capture_stream)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_read: a read-back inside a graph capture window would end the capture");
    MI355_HIP(ctx, hipMemcpyAsync(dst_host, src_dptr, bytes, hipMemcpyDeviceToHost, stream_of(ctx, stream)));
    return MI355_OK;
}

MI355_API int32_t mi355_read(mi355_ctx *ctx, mi355_stream stream, void *dst_host, const void *src_dptr,
                             uint64_t bytes)
{
    int32_t rc = mi355_read_async(ctx, stream, dst_host, src_dptr, bytes);
    if (rc != MI355_OK) return rc;
    return mi355_sync(ctx, stream);
}

MI355_API int32_t mi355_write_2d(mi355_ctx *ctx, mi355_stream stream, void *dst_dptr, uint64_t dst_pitch,
                                 const void *src_host, uint64_t src_pitch, uint64_t width_bytes, uint64_t rows)
{
    MI355_REQUIRE_CTX(ctx);
    if (width_bytes == 0 || rows == 0) return MI355_OK;
    if (!dst_dptr || !src_host) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_write_2d: NULL pointer");
    if (dst_pitch < width_bytes || src_pitch < width_bytes)
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "pitch smaller than row width");
    MI355_HIP(ctx, hipMemcpy2DAsync(dst_dptr, dst_pitch, src_host, src_pitch, width_bytes, rows,
                                    hipMemcpyHostToDevice, stream_of(ctx, stream)));
    return MI355_OK;
}

MI355_API int32_t mi355_read_2d(mi355_ctx *ctx, mi355_stream stream, void *dst_host, uint64_t dst_pitch,
                                const void *src_dptr, uint64_t src_pitch, uint64_t width_bytes, uint64_t rows)
{
    MI355_REQUIRE_CTX(ctx);
    if (width_bytes == 0 || rows == 0) return MI355_OK;
    if (!dst_host || !src_dptr) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_read_2d: NULL pointer");
    if (dst_pitch < width_bytes || src_pitch < width_bytes)
        return fail(ctx, MI355_E_UNSUPPORTED_STRIDES, "pitch smaller than row width");
    if (ctx->capturing && stream_of(ctx, stream) == ctx->capture_stream)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_read_2d: a read-back inside a graph capture window would end the capture");
    MI355_HIP(ctx, hipMemcpy2DAsync(dst_host, dst_pitch, src_dptr, src_pitch, width_bytes, rows,
                                    hipMemcpyDeviceToHost, stream_of(ctx, stream)));
    return mi355_sync(ctx, stream);
}

MI355_API int32_t mi355_copy_d2d(mi355_ctx *ctx, mi355_stream stream, void *dst_dptr, const void *src_dptr,
                                 uint64_t bytes)
{
    MI355_REQUIRE_CTX(ctx);
    if (bytes == 0) return MI355_OK;
    if (!dst_dptr || !src_dptr) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_d2d: NULL pointer");
    MI355_HIP(ctx, hipMemcpyAsync(dst_dptr, src_dptr, bytes, hipMemcpyDeviceToDevice, stream_of(ctx, stream)));
    return MI355_OK;
}

// ComputeClient::to_client (client.rs:733-751): one process holding several devices moves a buffer from one client's
// device to another's.  Here that is a peer copy over xGMI (hipMemcpyPeerAsync) instead of the reference's NCCL
// send / recv pair between two server threads (cuda server.rs:799-926): same process, so no rendezvous is needed.
// Ordering: the copy runs on the destination's stream AFTER everything already queued on the source stream (event),
// and the source stream then waits for the copy, so that whatever it does next to the source buffer -- including the
// memory pool handing it out again -- is ordered behind the read.
MI355_API int32_t mi355_copy_to_ctx(mi355_ctx *src_ctx, mi355_stream src_stream, const void *src_dptr, mi355_ctx *dst_ctx,
                                    mi355_stream dst_stream, void *dst_dptr, uint64_t bytes)
{
    if (!src_ctx || !dst_ctx) return MI355_E_INVALID_ARGUMENT;
    if (bytes == 0) return MI355_OK;
    // both servers take part: lock them in address order (two threads copying in opposite directions must not deadlock)
    mi355_ctx *lo = src_ctx < dst_ctx ? src_ctx : dst_ctx, *hi = src_ctx < dst_ctx ? dst_ctx : src_ctx;
    std::lock_guard<std::recursive_mutex> g_lo(lo->mu), g_hi(hi->mu);
    if (!src_dptr || !dst_dptr) return fail(dst_ctx, MI355_E_INVALID_ARGUMENT, "mi355_copy_to_ctx: NULL pointer");
    hipStream_t ss = stream_of(src_ctx, src_stream), ds = stream_of(dst_ctx, dst_stream);
    if (src_ctx->device != dst_ctx->device) {
        int can = 0;
        MI355_HIP(dst_ctx, hipDeviceCanAccessPeer(&can, dst_ctx->device, src_ctx->device));
        if (!can) return fail(dst_ctx, MI355_E_UNSUPPORTED, "mi355_copy_to_ctx: device %d cannot access device %d", dst_ctx->device, src_ctx->device);
    }
    MI355_HIP(src_ctx, hipSetDevice(src_ctx->device));
    if (!src_ctx->fence_a) MI355_HIP(src_ctx, hipEventCreateWithFlags(&src_ctx->fence_a, hipEventDisableTiming));
    MI355_HIP(src_ctx, hipEventRecord(src_ctx->fence_a, ss));
    MI355_HIP(dst_ctx, hipSetDevice(dst_ctx->device));
    MI355_HIP(dst_ctx, hipStreamWaitEvent(ds, src_ctx->fence_a, 0));
    MI355_HIP(dst_ctx, hipMemcpyPeerAsync(dst_dptr, dst_ctx->device, src_dptr, src_ctx->device, bytes, ds));
    if (!dst_ctx->fence_b) MI355_HIP(dst_ctx, hipEventCreateWithFlags(&dst_ctx->fence_b, hipEventDisableTiming));
    MI355_HIP(dst_ctx, hipEventRecord(dst_ctx->fence_b, ds));
    MI355_HIP(src_ctx, hipSetDevice(src_ctx->device));
    MI355_HIP(src_ctx, hipStreamWaitEvent(ss, dst_ctx->fence_b, 0));
    return MI355_OK;
}

MI355_API int32_t mi355_memset(mi355_ctx *ctx, mi355_stream stream, void *dptr, int32_t byte_value, uint64_t bytes)
{
    MI355_REQUIRE_CTX(ctx);
    if (bytes == 0) return MI355_OK;
    if (!dptr) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_memset: NULL pointer");
    MI355_HIP(ctx, hipMemsetAsync(dptr, byte_value, bytes, stream_of(ctx, stream)));
    return MI355_OK;
}

MI355_API int32_t mi355_sync(mi355_ctx *ctx, mi355_stream stream)
{
    MI355_REQUIRE_CTX(ctx);
    // a host synchronisation aborts an open capture (the reference defers or refuses them: crates/cubecl-hip/src/compute/
    // command.rs:404, :508): refuse it and keep the window alive
    // -- on the stream under capture; another lane is an ordinary stream (ThreadLocal capture mode)
    if (ctx->capturing && stream_of(ctx, stream) == ctx->capture_stream)
        return fail(ctx, MI355_E_UNSUPPORTED, "mi355_sync: not inside a graph capture window");
    hipError_t e = hipStreamSynchronize(stream_of(ctx, stream));
    if (e != hipSuccess) {
        hipGetLastError();
        queue_error(ctx, MI355_E_EXECUTION, 0, 0, "hipStreamSynchronize: %s", hipGetErrorString(e));
    }
    return report_queue(ctx);
}

MI355_API int32_t mi355_flush(mi355_ctx *ctx)
{
    MI355_REQUIRE_CTX(ctx);
    if (ctx->capturing) return MI355_OK;          // fenced frees are deferred to the first flush after the window (command.rs:404)
    if (!ctx->pending_free.empty()) {
        // frees wait behind every in-flight use (fence in the reference; here a device sync)
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            hipGetLastError();
            queue_error(ctx, MI355_E_EXECUTION, 0, 0, "hipDeviceSynchronize: %s", hipGetErrorString(e));
        }
        for (void *p : ctx->pending_free) {
            e = hipFree(p);
            if (e != hipSuccess) {
                hipGetLastError();
                queue_error(ctx, MI355_E_EXECUTION, 0, 0, "hipFree: %s", hipGetErrorString(e));
            }
        }
        ctx->pending_free.clear();
    }
    return report_queue(ctx);
}

/* =================================== Generic launch ====================================== */

MI355_API int32_t mi355_module_load(mi355_ctx *ctx, const void *image, size_t image_bytes, mi355_module *out_module)
{
    MI355_REQUIRE_CTX(ctx);
    if (!image || !image_bytes || !out_module) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_module_load: NULL argument");
    hipModule_t mod;
    hipError_t e = hipModuleLoadData(&mod, image);
    if (e != hipSuccess) {
        hipGetLastError();
        return fail(ctx, MI355_E_COMPILATION, "hipModuleLoadData: %s", hipGetErrorString(e));
    }
    *out_module = mod;
    return MI355_OK;
}

MI355_API int32_t mi355_module_unload(mi355_ctx *ctx, mi355_module module)
{
    MI355_REQUIRE_CTX(ctx);
    if (module) MI355_HIP(ctx, hipModuleUnload(reinterpret_cast<hipModule_t>(module)));
    return MI355_OK;
}

MI355_API int32_t mi355_module_get_function(mi355_ctx *ctx, mi355_module module, const char *name,
                                            mi355_function *out_function)
{
    MI355_REQUIRE_CTX(ctx);
    if (!module || !name || !out_function) return fail(ctx, MI355_E_INVALID_ARGUMENT, "NULL argument");
    hipFunction_t fn;
    hipError_t e = hipModuleGetFunction(&fn, reinterpret_cast<hipModule_t>(module), name);
    if (e != hipSuccess) {
        hipGetLastError();
        return fail(ctx, MI355_E_NOT_FOUND, "hipModuleGetFunction(%s): %s", name, hipGetErrorString(e));
    }
    *out_function = fn;
    return MI355_OK;
}

MI355_API int32_t mi355_launch(mi355_ctx *ctx, mi355_stream stream, mi355_function function,
                               const uint32_t grid[3], const uint32_t block[3], uint32_t shared_mem_bytes,
                               void *const *buffer_ptrs, uint32_t num_ptrs)
{
    MI355_REQUIRE_CTX(ctx);
    if (!function || !grid || !block) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_launch: NULL argument");
    if (num_ptrs && !buffer_ptrs) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_launch: NULL pointer table");
    // zero grid => no-op (client.rs:880-884; server.rs:796-799)
    if (grid[0] == 0 || grid[1] == 0 || grid[2] == 0) return MI355_OK;
    const mi355_device_props_t &p = ctx->props;
    // Resource validation; violations are queued and surface at flush/sync
    // (runtime_tests/launch.rs:226-348).
    if (shared_mem_bytes > p.max_shared_memory_size) {
        queue_error(ctx, MI355_E_SHARED_MEMORY, shared_mem_bytes, p.max_shared_memory_size,
                    "Too much shared memory requested. Requested %u bytes, maximum %llu bytes available.",
                    shared_mem_bytes, (unsigned long long)p.max_shared_memory_size);
        return MI355_OK;
    }
    if (block[0] > p.max_cube_dim[0] || block[1] > p.max_cube_dim[1] || block[2] > p.max_cube_dim[2]) {
        queue_error(ctx, MI355_E_CUBE_DIM, block[0], p.max_cube_dim[0],
                    "Cube dim exceeds maximum bounds. Requested (%u, %u, %u), max is (%u, %u, %u).", block[0],
                    block[1], block[2], p.max_cube_dim[0], p.max_cube_dim[1], p.max_cube_dim[2]);
        return MI355_OK;
    }
    const uint64_t units = (uint64_t)block[0] * block[1] * block[2];
    if (units == 0 || units > p.max_units_per_cube) {
        queue_error(ctx, MI355_E_UNITS, units, p.max_units_per_cube,
                    "Total unit count exceeds maximum. Requested %llu units, max units is %u.",
                    (unsigned long long)units, p.max_units_per_cube);
        return MI355_OK;
    }
    if (num_ptrs > p.max_bindings + 1) {
        queue_error(ctx, MI355_E_INVALID_ARGUMENT, num_ptrs, p.max_bindings, "too many bindings: %u", num_ptrs);
        return MI355_OK;
    }
    hipFunction_t fn = reinterpret_cast<hipFunction_t>(function);
    if (shared_mem_bytes > 64 * 1024) {
        // opt in to more than the default 64 KiB of dynamic LDS
        hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shared_mem_bytes);
        hipGetLastError();
    }
    // kernelParams[i] points at the i-th argument value, i.e. at the i-th device pointer
    // (context.rs:410-424).
    std::vector<void *> values(buffer_ptrs, buffer_ptrs + num_ptrs);
    std::vector<void *> params(num_ptrs);
    for (uint32_t i = 0; i < num_ptrs; ++i) params[i] = &values[i];
    hipError_t e = hipModuleLaunchKernel(fn, grid[0], grid[1], grid[2], block[0], block[1], block[2],
                                         shared_mem_bytes, stream_of(ctx, stream),
                                         num_ptrs ? params.data() : nullptr, nullptr);
    if (e != hipSuccess) {
        hipGetLastError();
        if (shared_mem_bytes > 64 * 1024 && (e == hipErrorInvalidValue || e == hipErrorLaunchOutOfResources || e == hipErrorOutOfMemory))
            // (only the codes a refused LDS request produces: a bad grid or a stale function stays a launch failure)
            // HIP has no driver-side attribute call for a hipFunction_t (hipFuncSetAttribute resolves host stubs only): when
            // the driver refuses the opt-in for a module kernel, the limit that really applies to it is the default one
            queue_error(ctx, MI355_E_SHARED_MEMORY, shared_mem_bytes, 64 * 1024,
                        "Too much shared memory requested. Requested %u bytes, maximum %u bytes available to a module kernel "
                        "(hipModuleLaunchKernel: %s).", shared_mem_bytes, 64 * 1024, hipGetErrorString(e));
        else
            queue_error(ctx, MI355_E_LAUNCH, 0, 0, "hipModuleLaunchKernel: %s", hipGetErrorString(e));
    }
    return MI355_OK;
}

/* =================================== Profiling =========================================== */

MI355_API int32_t mi355_profile_start(mi355_ctx *ctx, mi355_stream stream, uint64_t *out_token)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_token) return fail(ctx, MI355_E_INVALID_ARGUMENT, "out_token is NULL");
    size_t slot = ctx->profiles.size();
    for (size_t i = 0; i < ctx->profiles.size(); ++i)
        if (!ctx->profiles[i].live) { slot = i; break; }
    mi355_profile_slot s{};
    MI355_HIP(ctx, hipEventCreate(&s.start));
    MI355_HIP(ctx, hipEventCreate(&s.stop));
    s.live = true;
    MI355_HIP(ctx, hipEventRecord(s.start, stream_of(ctx, stream)));
    if (slot == ctx->profiles.size()) ctx->profiles.push_back(s);
    else ctx->profiles[slot] = s;
    *out_token = slot + 1;
    return MI355_OK;
}

MI355_API int32_t mi355_profile_stop(mi355_ctx *ctx, mi355_stream stream, uint64_t token, uint64_t *out_nanos)
{
    MI355_REQUIRE_CTX(ctx);
    if (token == 0 || token > ctx->profiles.size() || !ctx->profiles[token - 1].live)
        return fail(ctx, MI355_E_PROFILE, "unknown profile token %llu", (unsigned long long)token);
    mi355_profile_slot &s = ctx->profiles[token - 1];
    float ms = 0.f;
    hipError_t e = hipEventRecord(s.stop, stream_of(ctx, stream));
    if (e == hipSuccess) e = hipEventSynchronize(s.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, s.start, s.stop);
    hipEventDestroy(s.start);
    hipEventDestroy(s.stop);
    s.live = false;
    if (e != hipSuccess) {
        hipGetLastError();
        return fail(ctx, MI355_E_PROFILE, "profile stop: %s", hipGetErrorString(e));
    }
    if (out_nanos) *out_nanos = (uint64_t)((double)ms * 1.0e6);
    // a profiled region that hit an execution error poisons the measurement (server.rs:728-735)
    return report_queue(ctx);
}

/* =================================== Library scratch ===================================== */
namespace mi355 {

// Library-owned device scratch, one buffer per (stream, kind): calls on one stream are stream-ordered, so a
// buffer is never shared by two operations in flight.  Grows on demand (a synchronising hipMalloc, rare).
int32_t scratch_get(mi355_ctx *ctx, hipStream_t s, int kind, size_t bytes, void **out)
{
    auto &slot = ctx->scratch[{s, kind}];
    if (slot.second < bytes) {
        // growing means hipMalloc (+ a stream sync and hipFree): none of that is legal inside a capture window; the caller
        // falls back to a path without scratch, or the launch fails -- warm the sequence up once before capturing
        if (ctx->capturing) return MI355_E_UNSUPPORTED;
        if (slot.first) {
            auto pin = ctx->scratch_refs.find(slot.first);
            if (pin != ctx->scratch_refs.end() && pin->second > 0) {
                // a live graph replays kernels that carry this address: the buffer is retired, not freed -- the last
                // graph that pins it releases it (scratch_release)
                ctx->scratch_retired.insert(slot.first);
            } else {
                if (hipStreamSynchronize(s) != hipSuccess) return MI355_E_EXECUTION;
                hipFree(slot.first);
            }
            slot = {nullptr, 0};
        }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return MI355_E_OUT_OF_MEMORY;
        }
        slot = {p, bytes};
    }
    if (ctx->capturing) ctx->capture_scratch.insert(slot.first);   // becomes a pin when the capture ends
    *out = slot.first;
    return MI355_OK;
}

// One arrival-ticket word per stream in library-owned device scratch (zeroed when created, left zero by
// every completed call).  Calls on one stream are stream-ordered, so a word is never shared by two
// launches in flight.  A stream's slot is 16 KiB: word 0 is the reductions' top ticket, words 16 ... 511 are the per-strip
// tickets of gemm_nnrows.hip (mi355::strip_tickets_for_stream), and from byte 2048 on 32 group tickets of the array-wide
// reductions sit 256 bytes apart (reduce.hip arrive_is_last: one L2 serves the adds on one address at 11 ns each).
constexpr uint32_t TICKET_SLOT_BYTES = 16384;
int32_t ticket_for_stream(mi355_ctx *ctx, hipStream_t s, unsigned int **out)
{
    constexpr uint32_t SLOTS = 1024;
    if (ctx->capturing && (!ctx->ticket_buf || ctx->tickets_dirty))      // hipMalloc / hipMemset are not capturable
        return fail(ctx, MI355_E_UNSUPPORTED, "reduction inside a graph capture before its scratch exists: run it once before capturing");
    if (!ctx->ticket_buf) {
        MI355_HIP(ctx, hipMalloc(&ctx->ticket_buf, (size_t)SLOTS * TICKET_SLOT_BYTES));
        MI355_HIP(ctx, hipMemset(ctx->ticket_buf, 0, (size_t)SLOTS * TICKET_SLOT_BYTES));
    } else if (ctx->tickets_dirty) {
        MI355_HIP(ctx, hipDeviceSynchronize());
        MI355_HIP(ctx, hipMemset(ctx->ticket_buf, 0, (size_t)SLOTS * TICKET_SLOT_BYTES));
    }
    ctx->tickets_dirty = false;
    auto it = ctx->ticket_slots.find(s);
    if (it == ctx->ticket_slots.end()) {
        uint32_t slot;
        if (!ctx->ticket_free.empty()) { slot = ctx->ticket_free.back(); ctx->ticket_free.pop_back(); }
        else if (ctx->ticket_next < SLOTS) slot = ctx->ticket_next++;
        else return fail(ctx, MI355_E_UNSUPPORTED, "reductions were issued on more than %u live streams of one context", SLOTS);
        it = ctx->ticket_slots.emplace(s, slot).first;
    }
    if (ctx->capturing && s == ctx->capture_stream) ctx->capture_tickets.insert(it->second);   // becomes a pin when the capture ends (the other
                                                                                              // lanes run real work: nothing of theirs is a graph node)
    *out = reinterpret_cast<unsigned int *>(static_cast<char *>(ctx->ticket_buf) + (size_t)it->second * TICKET_SLOT_BYTES);
    return MI355_OK;
}

// gemm_nnrows.hip: the per-strip arrival words of the stream's slot (zero between calls, as the reductions' word)
int32_t strip_tickets_for_stream(mi355_ctx *ctx, hipStream_t s, unsigned int **out)
{
    unsigned int *t = nullptr;
    const int32_t rc = ticket_for_stream(ctx, s, &t);
    if (rc != MI355_OK) return rc;
    *out = t + 16;
    return MI355_OK;
}

void strip_tickets_mark_dirty(mi355_ctx *ctx) { ctx->tickets_dirty = true; }

void scratch_release(mi355_ctx *ctx, void *ptr)
{
    auto pin = ctx->scratch_refs.find(ptr);
    if (pin == ctx->scratch_refs.end()) return;
    if (--pin->second > 0) return;
    ctx->scratch_refs.erase(pin);
    if (ctx->scratch_retired.erase(ptr)) hipFree(ptr);             // nobody replays against it any more
}

}  // namespace mi355

// =================================== Graph capture ==========================================
// ComputeServer::{begin_capture, end_capture, replay, graph_destroy}
// (crates/cubecl-runtime/src/server/base.rs:472-532; the HIP backend's implementation over
// hipStreamBeginCapture / hipGraphInstantiate / hipGraphLaunch: crates/cubecl-hip/src/compute/server.rs:288-521).
// Everything this library launches is capturable once its lazily created scratch exists (run the sequence once
// before capturing -- the reference asks for the same warm-up, base.rs:453-470): kernels take their state from
// arguments, the reductions' arrival tickets are reset by the kernels themselves, nothing synchronises the host.

struct mi355_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    uint64_t id;                          // the capture window it came from: pins pool blocks under this id
    std::set<void *> scratch;             // library scratch its nodes carry
    std::set<uint32_t> tickets;           // arrival-ticket slots its nodes carry
    std::set<hipStream_t> streams;        // streams it was replayed on (waited for before the executable dies)
};

// A capture that fails pins nothing: slots handed out in its window go back to being ordinary -- and the slot of a stream that was
// destroyed INSIDE the window (mi355_stream_destroy retired it because the window might have become a graph) returns to the free list
// unless a live graph still carries it.  (Advisor, round 5: one of the 1 024 slots leaked per failed capture.)
static void drop_capture_tickets(mi355_ctx *ctx)
{
    for (uint32_t t : ctx->capture_tickets) {
        const auto pin = ctx->ticket_refs.find(t);
        if (ctx->ticket_retired.count(t) && (pin == ctx->ticket_refs.end() || pin->second <= 0)) {
            ctx->ticket_retired.erase(t);
            ctx->ticket_free.push_back(t);
        }
    }
    ctx->capture_tickets.clear();
}

MI355_API int32_t mi355_graph_begin_capture(mi355_ctx *ctx, mi355_stream stream)
{
    MI355_REQUIRE_CTX(ctx);
    if (ctx->capturing) return fail(ctx, MI355_E_INVALID_ARGUMENT, "begin_capture: a capture is already open on this context");
    MI355_HIP(ctx, hipStreamBeginCapture(stream_of(ctx, stream), hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    ctx->capture_stream = stream_of(ctx, stream);
    ctx->capture_id = ctx->next_capture_id++;
    ctx->capture_scratch.clear();
    ctx->capture_tickets.clear();
    return MI355_OK;
}

MI355_API int32_t mi355_graph_end_capture(mi355_ctx *ctx, mi355_stream stream, mi355_graph **out_graph)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_graph) return fail(ctx, MI355_E_INVALID_ARGUMENT, "end_capture: out_graph is NULL");
    *out_graph = nullptr;
    if (!ctx->capturing) return fail(ctx, MI355_E_INVALID_ARGUMENT, "end_capture without begin_capture");
    ctx->capturing = false;
    ctx->capture_stream = nullptr;
    ctx->captured_events.clear();
    const uint64_t id = ctx->capture_id;
    ctx->capture_id = 0;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(stream_of(ctx, stream), &g);
    hipGraphExec_t x = nullptr;
    if (e != hipSuccess || !g) {
        (void)hipGetLastError();
        pool_release_graph(ctx, id);          // nothing will ever replay: what the window pinned is free memory again
        ctx->capture_scratch.clear();
        drop_capture_tickets(ctx);
        return fail(ctx, MI355_E_EXECUTION, "hipStreamEndCapture: %s (an operation in the window was not capturable: "
                    "run the sequence once before capturing so that library scratch exists)", hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        hipGraphDestroy(g);
        pool_release_graph(ctx, id);
        ctx->capture_scratch.clear();
        drop_capture_tickets(ctx);
        return fail(ctx, MI355_E_EXECUTION, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    // From here on the graph owns a pin on every pool block allocated or freed in its window (pool.cpp keeps them out of
    // the free lists under `id`) and on the library scratch its kernels were captured with.
    mi355_graph *out = new mi355_graph{g, x, id, {}, {}, {}};
    out->scratch.swap(ctx->capture_scratch);
    for (void *p : out->scratch) ++ctx->scratch_refs[p];
    out->tickets.swap(ctx->capture_tickets);
    for (uint32_t t : out->tickets) ++ctx->ticket_refs[t];
    ctx->live_graphs.insert(id);
    *out_graph = out;
    return MI355_OK;
}

MI355_API int32_t mi355_graph_replay(mi355_ctx *ctx, mi355_stream stream, mi355_graph *graph)
{
    MI355_REQUIRE_CTX(ctx);
    if (!graph) return fail(ctx, MI355_E_NOT_FOUND, "replay: unknown graph");
    graph->streams.insert(stream_of(ctx, stream));
    hipError_t e = hipGraphLaunch(graph->exec, stream_of(ctx, stream));
    if (e != hipSuccess) {                       // fire-and-forget like launch: queued, reported by flush / sync
        (void)hipGetLastError();
        queue_error(ctx, MI355_E_LAUNCH, 0, 0, "hipGraphLaunch: %s", hipGetErrorString(e));
    }
    return MI355_OK;
}

MI355_API int32_t mi355_graph_destroy(mi355_ctx *ctx, mi355_graph *graph)
{
    MI355_REQUIRE_CTX(ctx);
    if (!graph) return MI355_OK;
    // replay returns at enqueue time: wait for the replays still running against this executable before it goes
    // (crates/cubecl-hip/src/compute/server.rs:497-521); a failed wait means the stream has faulted -- nothing runs any
    // more, destroying is safe -- and is surfaced through the error queue
    for (hipStream_t s : graph->streams) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            queue_error(ctx, MI355_E_EXECUTION, 0, 0, "graph_destroy: hipStreamSynchronize: %s", hipGetErrorString(e));
        }
    }
    hipGraphExecDestroy(graph->exec);
    hipGraphDestroy(graph->graph);
    ctx->live_graphs.erase(graph->id);
    pool_release_graph(ctx, graph->id);          // its pinned blocks are ordinary free memory again
    for (void *p : graph->scratch) scratch_release(ctx, p);
    for (uint32_t t : graph->tickets) {
        auto pin = ctx->ticket_refs.find(t);
        if (pin == ctx->ticket_refs.end() || --pin->second > 0) continue;
        ctx->ticket_refs.erase(pin);
        if (ctx->ticket_retired.erase(t)) ctx->ticket_free.push_back(t);   // its stream died meanwhile: the slot is reusable now
    }
    delete graph;
    return MI355_OK;
}
