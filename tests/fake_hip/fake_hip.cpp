// Fake HIP runtime + the few runtime.cpp helpers pool.cpp links against + test controls (see hip/hip_runtime.h here).
// Device memory is a range of made-up addresses (the pool never dereferences what it hands out); a stream is a pair of
// counters (work submitted, work completed) that the test advances by hand, which makes "has this event completed"
// a deterministic question.
#include "../../cubecl_amd/csrc/internal.hpp"

#include <set>

struct fake_hip_stream { uint64_t submitted = 0, completed = 0; };
struct fake_hip_event { fake_hip_stream *stream = nullptr; uint64_t seq = 0; };

namespace {
struct fake_state {
    std::map<uintptr_t, size_t> live;         // device allocations
    uintptr_t next = 0x7f0000000000ull;
    uint64_t capacity = ~0ull, in_use = 0;
    uint64_t mallocs = 0, frees = 0, bad_frees = 0, event_creates = 0, event_destroys = 0, event_queries = 0, device_syncs = 0;
    std::set<fake_hip_stream *> streams;
    std::set<fake_hip_event *> events;
} g;
}  // namespace

extern "C" {

hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipErrorOutOfMemory ? "out of memory" : e == hipSuccess ? "no error" : "fake error"; }

hipError_t hipMalloc(void **ptr, size_t bytes)
{
    if (g.in_use + bytes > g.capacity) { *ptr = nullptr; return hipErrorOutOfMemory; }
    const uintptr_t p = g.next;
    g.next += (bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20) + (2u << 20);    // 2 MiB aligned, with a guard gap
    g.live[p] = bytes;
    g.in_use += bytes;
    ++g.mallocs;
    *ptr = reinterpret_cast<void *>(p);
    return hipSuccess;
}

hipError_t hipFree(void *ptr)
{
    auto it = g.live.find(reinterpret_cast<uintptr_t>(ptr));
    if (it == g.live.end()) { ++g.bad_frees; return hipErrorInvalidValue; }
    g.in_use -= it->second;
    g.live.erase(it);
    ++g.frees;
    return hipSuccess;
}

hipError_t hipDeviceSynchronize(void)
{
    for (fake_hip_stream *s : g.streams) s->completed = s->submitted;
    ++g.device_syncs;
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t *event, unsigned)
{
    *event = new fake_hip_event();
    g.events.insert(*event);
    ++g.event_creates;
    return hipSuccess;
}

hipError_t hipEventDestroy(hipEvent_t event)
{
    if (!g.events.erase(event)) return hipErrorInvalidValue;
    delete event;
    ++g.event_destroys;
    return hipSuccess;
}

hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream)
{
    if (!event || !stream) return hipErrorInvalidValue;
    event->stream = stream;
    event->seq = ++stream->submitted;          // the record itself is a piece of work on the stream
    return hipSuccess;
}

hipError_t hipEventQuery(hipEvent_t event)
{
    ++g.event_queries;
    if (!event || !event->stream) return hipErrorInvalidValue;
    return event->stream->completed >= event->seq ? hipSuccess : hipErrorNotReady;
}

}  // extern "C"

// ---- what pool.cpp needs from runtime.cpp -------------------------------------------------------------------------
namespace mi355 {
int32_t fail(mi355_ctx *ctx, int32_t code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return code;
}
void queue_error(mi355_ctx *ctx, int32_t code, uint64_t requested, uint64_t max, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx->errors.push_back({code, requested, max, std::string(buf)});
}
int32_t map_hip_error(hipError_t e) { return e == hipSuccess ? MI355_OK : e == hipErrorOutOfMemory ? MI355_E_OUT_OF_MEMORY : MI355_E_EXECUTION; }
}  // namespace mi355

// ---- test controls --------------------------------------------------------------------------------------------------
#define TEST_API extern "C" __attribute__((visibility("default")))

TEST_API mi355_ctx *pooltest_ctx_create(uint64_t max_page_size)
{
    mi355_ctx *ctx = new mi355_ctx();
    ctx->props.max_page_size = max_page_size;
    ctx->compute_stream = new fake_hip_stream();
    g.streams.insert(ctx->compute_stream);
    return ctx;
}
TEST_API void pooltest_ctx_destroy(mi355_ctx *ctx)
{
    mi355::pool_destroy(ctx);
    g.streams.erase(ctx->compute_stream);
    delete ctx->compute_stream;
    delete ctx;
}
TEST_API void *pooltest_stream_create(void)
{
    fake_hip_stream *s = new fake_hip_stream();
    g.streams.insert(s);
    return s;
}
TEST_API void pooltest_stream_destroy(void *s)
{
    g.streams.erase(static_cast<fake_hip_stream *>(s));
    delete static_cast<fake_hip_stream *>(s);
}
// everything submitted to the stream so far (NULL: the context's compute stream) has now run
TEST_API void pooltest_stream_complete(mi355_ctx *ctx, void *s)
{
    fake_hip_stream *st = s ? static_cast<fake_hip_stream *>(s) : ctx->compute_stream;
    st->completed = st->submitted;
}
TEST_API void pooltest_set_capturing(mi355_ctx *ctx, int32_t on) { ctx->capturing = on != 0; }
TEST_API void pooltest_set_capacity(uint64_t bytes) { g.capacity = bytes; }
TEST_API const char *pooltest_last_error(mi355_ctx *ctx) { return ctx->last_error.c_str(); }
// {mallocs, frees, bad frees, live device allocations, bytes in use on the device, events alive, event queries, device syncs}
TEST_API void pooltest_counters(uint64_t out[8])
{
    out[0] = g.mallocs; out[1] = g.frees; out[2] = g.bad_frees; out[3] = g.live.size(); out[4] = g.in_use;
    out[5] = g.events.size(); out[6] = g.event_queries; out[7] = g.device_syncs;
}
// is [ptr, ptr + bytes) inside one live device allocation?
TEST_API int32_t pooltest_inside_allocation(void *ptr, uint64_t bytes)
{
    const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
    auto it = g.live.upper_bound(p);
    if (it == g.live.begin()) return 0;
    --it;
    return p >= it->first && p + bytes <= it->first + it->second;
}
