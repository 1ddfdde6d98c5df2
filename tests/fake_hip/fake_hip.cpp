// Fake HIP runtime + the few runtime.cpp helpers pool.cpp links against + test controls (see hip/hip_runtime.h here).
// Device memory is a range of made-up addresses (the pool never dereferences what it hands out); a stream is a pair of
// counters (work submitted, work completed) that the test advances by hand, which makes "has this event completed"
// a deterministic question.
#include "../../cubecl_amd/csrc/internal.hpp"

#include <mutex>
#include <set>
#include <sys/mman.h>

struct fake_hip_stream { uint64_t submitted = 0, completed = 0; };
struct fake_hip_event { fake_hip_stream *stream = nullptr; uint64_t seq = 0; };

namespace {
struct fake_state {
    std::map<uintptr_t, size_t> live;         // device allocations
    uint64_t capacity = ~0ull, in_use = 0;
    uint64_t mallocs = 0, frees = 0, bad_frees = 0, event_creates = 0, event_destroys = 0, event_queries = 0, device_syncs = 0;
    std::set<fake_hip_stream *> streams;
    std::set<fake_hip_event *> events;
} g;
// One lock around the whole fake: the real runtime is thread-safe and the library is driven from one thread per device
// (tests/test_comm_cpu.py THREADED: four contexts created, used and destroyed concurrently).
std::recursive_mutex g_mu;
#define LOCKED std::lock_guard<std::recursive_mutex> lock_(g_mu)
}  // namespace

extern "C" {

hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipErrorOutOfMemory ? "out of memory" : e == hipSuccess ? "no error" : "fake error"; }

hipError_t hipMalloc(void **ptr, size_t bytes)
{
    LOCKED;
    if (g.in_use + bytes > g.capacity) { *ptr = nullptr; return hipErrorOutOfMemory; }
    // address space only (MAP_NORESERVE): pages exist once a copy touches them, so 256 MiB slab pages cost nothing
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { *ptr = nullptr; return hipErrorOutOfMemory; }
    g.live[reinterpret_cast<uintptr_t>(m)] = bytes;
    g.in_use += bytes;
    ++g.mallocs;
    *ptr = m;
    return hipSuccess;
}

hipError_t hipFree(void *ptr)
{
    LOCKED;
    auto it = g.live.find(reinterpret_cast<uintptr_t>(ptr));
    if (it == g.live.end()) { ++g.bad_frees; return hipErrorInvalidValue; }
    munmap(ptr, it->second);
    g.in_use -= it->second;
    g.live.erase(it);
    ++g.frees;
    return hipSuccess;
}

hipError_t hipDeviceSynchronize(void)
{
    LOCKED;
    for (fake_hip_stream *s : g.streams) s->completed = s->submitted;
    ++g.device_syncs;
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t *event, unsigned)
{
    LOCKED;
    *event = new fake_hip_event();
    g.events.insert(*event);
    ++g.event_creates;
    return hipSuccess;
}

hipError_t hipEventDestroy(hipEvent_t event)
{
    LOCKED;
    if (!g.events.erase(event)) return hipErrorInvalidValue;
    delete event;
    ++g.event_destroys;
    return hipSuccess;
}

hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream)
{
    LOCKED;
    if (!event || !stream) return hipErrorInvalidValue;
    event->stream = stream;
    event->seq = ++stream->submitted;          // the record itself is a piece of work on the stream
    return hipSuccess;
}

hipError_t hipEventQuery(hipEvent_t event)
{
    LOCKED;
    ++g.event_queries;
    if (!event || !event->stream) return hipErrorInvalidValue;
    return event->stream->completed >= event->seq ? hipSuccess : hipErrorNotReady;
}

}  // extern "C"

#ifndef FAKE_WITH_RUNTIME
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

#endif

// ---- test controls --------------------------------------------------------------------------------------------------
#define TEST_API extern "C" __attribute__((visibility("default")))

#ifndef FAKE_WITH_RUNTIME
TEST_API mi355_ctx *pooltest_ctx_create(uint64_t max_page_size)
{
    LOCKED;
    mi355_ctx *ctx = new mi355_ctx();
    ctx->props.max_page_size = max_page_size;
    ctx->compute_stream = new fake_hip_stream();
    g.streams.insert(ctx->compute_stream);
    return ctx;
}
TEST_API void pooltest_ctx_destroy(mi355_ctx *ctx)
{
    LOCKED;
    mi355::pool_destroy(ctx);
    g.streams.erase(ctx->compute_stream);
    delete ctx->compute_stream;
    delete ctx;
}
#endif
TEST_API void *pooltest_stream_create(void)
{
    LOCKED;
    fake_hip_stream *s = new fake_hip_stream();
    g.streams.insert(s);
    return s;
}
TEST_API void pooltest_stream_destroy(void *s)
{
    LOCKED;
    g.streams.erase(static_cast<fake_hip_stream *>(s));
    delete static_cast<fake_hip_stream *>(s);
}
// everything submitted to the stream so far (NULL: the context's compute stream) has now run
TEST_API void pooltest_stream_complete(mi355_ctx *ctx, void *s)
{
    LOCKED;
    fake_hip_stream *st = s ? static_cast<fake_hip_stream *>(s) : ctx->compute_stream;
    st->completed = st->submitted;
}
TEST_API void pooltest_set_capturing(mi355_ctx *ctx, int32_t on) { ctx->capturing = on != 0; }
TEST_API void pooltest_set_capacity(uint64_t bytes) { LOCKED; g.capacity = bytes; }
TEST_API const char *pooltest_last_error(mi355_ctx *ctx) { return ctx->last_error.c_str(); }
// {mallocs, frees, bad frees, live device allocations, bytes in use on the device, events alive, event queries, device syncs}
TEST_API void pooltest_counters(uint64_t out[8])
{
    LOCKED;
    out[0] = g.mallocs; out[1] = g.frees; out[2] = g.bad_frees; out[3] = g.live.size(); out[4] = g.in_use;
    out[5] = g.events.size(); out[6] = g.event_queries; out[7] = g.device_syncs;
}
// is [ptr, ptr + bytes) inside one live device allocation?
TEST_API int32_t pooltest_inside_allocation(void *ptr, uint64_t bytes)
{
    LOCKED;
    const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
    auto it = g.live.upper_bound(p);
    if (it == g.live.begin()) return 0;
    --it;
    return p >= it->first && p + bytes <= it->first + it->second;
}

// ---- the rest of the HIP surface runtime.cpp / comm.cpp use: everything executes at once, in order -------------------
struct fake_hip_module { int tag; };
struct fake_hip_function { char name[64]; };
struct fake_hip_graph { int nodes; };
struct fake_hip_graph_exec { int nodes; };

namespace {
struct runtime_state {
    char arch[64] = "gfx950:sramecc+:xnack-";
    int warp = 64, devices = 1;
    hipError_t fail_next_sync = hipSuccess, fail_next_launch = hipSuccess;
    uint64_t launches = 0, graph_launches = 0, captured = 0;
    unsigned last_launch[8] = {0};          // grid xyz, block xyz, lds bytes, number of parameters seen
    uintptr_t last_params[8] = {0};
    bool capturing = false;
    int max_dynamic_lds = 0;
    hipError_t fail_next_end_capture = hipSuccess;
    uint64_t stream_syncs = 0, stream_waits = 0;
    hipStream_t last_wait_stream = nullptr, last_sync_stream = nullptr;
} r;
void work(hipStream_t s) { LOCKED; if (s) { ++s->submitted; s->completed = s->submitted; } }
}  // namespace

extern "C" {
hipError_t hipGetDeviceCount(int *count)
{
    static const int from_env = [] { const char *e = getenv("FAKE_HIP_DEVICES"); return e ? atoi(e) : 0; }();   // (one rank per process: tests/test_bench_cpu.py)
    if (from_env > 0 && r.devices == 1) r.devices = from_env;
    *count = r.devices;
    return r.devices > 0 ? hipSuccess : hipErrorNoDevice;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "Fake Instinct");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "%s", r.arch);
    p->totalGlobalMem = 288ull << 30; p->sharedMemPerBlock = 64 << 10; p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->textureAlignment = 256; p->surfaceAlignment = 256; p->warpSize = r.warp; p->maxThreadsPerBlock = 1024;
    p->maxThreadsDim[0] = p->maxThreadsDim[1] = p->maxThreadsDim[2] = 1024;
    p->maxGridSize[0] = 2147483647; p->maxGridSize[1] = p->maxGridSize[2] = 65535;
    p->multiProcessorCount = 256; p->clockRate = 2400000; p->memoryClockRate = 2000000; p->memoryBusWidth = 8192; p->l2CacheSize = 4 << 20;
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *f, size_t *t) { LOCKED; *t = 288ull << 30; *f = *t - g.in_use; return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { LOCKED; *s = new fake_hip_stream(); g.streams.insert(*s); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { LOCKED; g.streams.erase(s); delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s)
{
    LOCKED;
    ++r.stream_syncs;
    r.last_sync_stream = s;
    if (s) s->completed = s->submitted;
    const hipError_t e = r.fail_next_sync;
    r.fail_next_sync = hipSuccess;
    return e;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t, unsigned) { ++r.stream_waits; r.last_wait_stream = s; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventSynchronize(hipEvent_t e) { LOCKED; if (e && e->stream) e->stream->completed = e->stream->submitted; return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 1.5f; return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { *p = malloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st) { memcpy(d, s, n); work(st); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st)
{
    for (size_t i = 0; i < h; ++i) memcpy(static_cast<char *>(d) + i * dp, static_cast<const char *>(s) + i * sp, w);
    work(st);
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t st) { memcpy(d, s, n); work(st); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st) { memset(d, v, n); work(st); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipModuleLoadData(hipModule_t *m, const void *image)
{
    if (memcmp(image, "FAKEHSACO", 9) != 0) return hipErrorInvalidValue;      // what a bad code object gets from the driver
    *m = new fake_hip_module{1};
    return hipSuccess;
}
hipError_t hipModuleUnload(hipModule_t m) { delete m; return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t *f, hipModule_t, const char *name)
{
    if (strncmp(name, "missing", 7) == 0) return hipErrorNotFound;
    *f = new fake_hip_function();
    snprintf((*f)->name, sizeof (*f)->name, "%s", name);
    return hipSuccess;
}
hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                 unsigned lds, hipStream_t st, void **params, void **)
{
    if (r.fail_next_launch != hipSuccess) { const hipError_t e = r.fail_next_launch; r.fail_next_launch = hipSuccess; return e; }
    const unsigned v[7] = {gx, gy, gz, bx, by, bz, lds};
    memcpy(r.last_launch, v, sizeof v);
    for (int i = 0; i < 8; ++i) r.last_params[i] = 0;
    if (params) for (int i = 0; i < 8 && i < (int)r.last_launch[7]; ++i) r.last_params[i] = *reinterpret_cast<uintptr_t *>(params[i]);
    if (r.capturing) ++r.captured; else ++r.launches;
    work(st);
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int value) { r.max_dynamic_lds = value; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { r.capturing = true; r.captured = 0; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *graph)
{
    r.capturing = false;
    if (r.fail_next_end_capture != hipSuccess) { const hipError_t e = r.fail_next_end_capture; r.fail_next_end_capture = hipSuccess; *graph = nullptr; return e; }
    *graph = new fake_hip_graph{(int)r.captured};
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t *x, hipGraph_t graph, void *, char *, size_t) { *x = new fake_hip_graph_exec{graph->nodes}; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t st) { r.graph_launches += 1; r.launches += (uint64_t)x->nodes; work(st); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t x) { delete x; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t graph) { delete graph; return hipSuccess; }
}  // extern "C"

TEST_API void faketest_set_device(const char *arch, int32_t warp, int32_t devices) { snprintf(r.arch, sizeof r.arch, "%s", arch); r.warp = warp; r.devices = devices; }
TEST_API void faketest_fail_next(int32_t sync_error, int32_t launch_error) { r.fail_next_sync = sync_error; r.fail_next_launch = launch_error; }
TEST_API void faketest_expect_params(uint32_t n) { r.last_launch[7] = n; }
// {launches executed, graph replays, nodes in the last capture, grid xyz, block xyz, lds, max dynamic LDS attribute, params...}
TEST_API void faketest_launch_log(uint64_t out[20])
{
    out[0] = r.launches; out[1] = r.graph_launches; out[2] = r.captured;
    for (int i = 0; i < 7; ++i) out[3 + i] = r.last_launch[i];
    out[10] = (uint64_t)r.max_dynamic_lds;
    for (int i = 0; i < 8; ++i) out[11 + i] = r.last_params[i];
}

// {hipStreamSynchronize calls, hipStreamWaitEvent calls, stream of the last wait, stream of the last synchronize}
TEST_API void faketest_stream_log(uint64_t out[4])
{
    out[0] = r.stream_syncs; out[1] = r.stream_waits;
    out[2] = reinterpret_cast<uintptr_t>(r.last_wait_stream); out[3] = reinterpret_cast<uintptr_t>(r.last_sync_stream);
}
TEST_API void faketest_fail_end_capture(int32_t error) { r.fail_next_end_capture = (hipError_t)error; }
#ifdef FAKE_WITH_RUNTIME
// library scratch has no entry point of its own (the GEMM launchers call it): reach it directly
TEST_API int32_t faketest_scratch_get(mi355_ctx *ctx, void *stream, int32_t kind, uint64_t bytes, void **out)
{
    return mi355::scratch_get(ctx, stream ? static_cast<hipStream_t>(stream) : ctx->compute_stream, kind, bytes, out);
}
TEST_API void faketest_set_comm_dirty(mi355_ctx *ctx, int32_t on) { ctx->comm_dirty = on != 0; }
// ... and so do the arrival-ticket slots (the reductions and gemm_nnrows.hip call it): returns the slot index of the stream
TEST_API int32_t faketest_ticket_slot(mi355_ctx *ctx, void *stream, int64_t *out_slot)
{
    unsigned int *t = nullptr;
    const int32_t rc = mi355::ticket_for_stream(ctx, stream ? static_cast<hipStream_t>(stream) : ctx->compute_stream, &t);
    if (rc == MI355_OK) *out_slot = (reinterpret_cast<char *>(t) - static_cast<char *>(ctx->ticket_buf)) / 16384;
    return rc;
}
#endif
