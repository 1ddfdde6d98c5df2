// internal.hpp -- shared state of libmi355cube.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/mi355cube.h"

#define MI355_API extern "C" __attribute__((visibility("default")))

struct mi355_queued_error {
    int32_t code;
    uint64_t requested;
    uint64_t max;
    std::string message;
};

struct mi355_profile_slot {
    hipEvent_t start;
    hipEvent_t stop;
    bool live;
};

namespace mi355 { struct memory_pool; }   // pool.cpp

// One per DeviceId; the reference's HipServer + HipContext
// (crates/cubecl-hip/src/compute/server.rs:151-161).
struct mi355_ctx {
    int device = 0;
    hipStream_t compute_stream = nullptr;  // hipStreamNonBlocking (stream.rs:91-99)
    hipStream_t comm_stream = nullptr;     // dedicated collective stream (cuda server.rs:749)
    mi355_device_props_t props{};
    std::string last_error;
    std::deque<mi355_queued_error> errors;  // per-stream error queue collapsed to per-server
    std::vector<void *> pending_free;       // PendingDropQueue (stream.rs:134-150)
    std::vector<mi355_profile_slot> profiles;
    hipEvent_t fence_a = nullptr;  // reusable fences for the comm <-> compute hand-offs
    hipEvent_t fence_b = nullptr;
    bool comm_dirty = false;
    // collectives of a few bytes run in the CALLER's stream order (comm.cpp collective_stream): the stream that carries the last such
    // collective not yet fenced towards another stream, and the event mi355_sync_collective(other stream) waits on
    hipStream_t inline_stream = nullptr;
    bool inline_dirty = false;
    hipEvent_t fence_c = nullptr;
    void *ticket_buf = nullptr;            // library-owned device scratch: arrival tickets of the reductions
    std::unordered_map<hipStream_t, uint32_t> ticket_slots;
    std::vector<uint32_t> ticket_free;     // slots of destroyed streams (mi355_stream_destroy), reused first
    uint32_t ticket_next = 0;              // first slot never handed out
    bool tickets_dirty = false;
    bool capturing = false;                // a hipStream capture window is open (graph API)
    hipStream_t capture_stream = nullptr;  // ... on this stream (ThreadLocal mode): the other lanes keep running real work
    std::set<void *> captured_events;      // events recorded on capture_stream inside the open window (graph nodes)
    // Memory a graph replays against must outlive the graph (the reference routes the allocations of a capture window into
    // a persistent pool and pins them for the graph's lifetime: crates/cubecl-hip/src/compute/server.rs:288-521):
    //   capture_id      id of the open window / of the graph it becomes (0: none)
    //   live_graphs     ids of graphs not yet destroyed
    //                   (pool.cpp keeps the blocks freed while pinned -- allocated or freed inside a window -- out of
    //                   its free lists under that id until mi355_graph_destroy)
    //   capture_scratch library scratch buffers handed out inside the open window
    //   scratch_refs    scratch buffer -> number of live graphs whose nodes carry its address; a pinned buffer is never
    //                   freed or regrown in place (scratch_get retires it and the last graph releases it)
    uint64_t capture_id = 0, next_capture_id = 1;
    std::set<uint64_t> live_graphs;
    std::set<void *> capture_scratch;
    std::map<void *, int> scratch_refs;
    std::set<void *> scratch_retired;      // pinned scratch that scratch_get has replaced by a bigger buffer
    // ... and so must the arrival-ticket slot a captured reduction / strip kernel counts on: a destroyed stream's slot goes back
    // to ticket_free only when no live graph carries its address (a replay sharing the words with a new stream's launches would
    // miscount arrivals): capture_tickets = slots handed out inside the open window, ticket_refs = slot -> live graphs carrying
    // it, ticket_retired = slots of destroyed streams still pinned
    std::set<uint32_t> capture_tickets, ticket_retired;
    std::map<uint32_t, int> ticket_refs;
    // library-owned device scratch per (stream, kind): split-K slabs, re-laid-out GEMM operands
    std::map<std::pair<hipStream_t, int>, std::pair<void *, size_t>> scratch;
    mi355::memory_pool *pool = nullptr;    // caching allocator behind mi355_pool_* (pool.cpp)
    std::set<const void *> lds_opted;  // kernels whose dynamic-LDS attribute is already raised on this device (lds_opt_in)
    // One context = one server: the reference funnels every call through one runner thread per device
    // (crates/cubecl-common/src/device/handle/channel.rs:75-110).  Bindings without that discipline (Python: a Handle
    // dropped by the garbage collector on another thread while ctypes has released the GIL) are serialised here.
    std::recursive_mutex mu;
};

namespace mi355 {

int32_t fail(mi355_ctx *ctx, int32_t code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
void queue_error(mi355_ctx *ctx, int32_t code, uint64_t requested, uint64_t max, const char *fmt, ...)
    __attribute__((format(printf, 5, 6)));
int32_t map_hip_error(hipError_t e);
// Checks the sticky/last launch error after a kernel launch and queues it (fire-and-forget
// contract: crates/cubecl-hip/src/compute/server.rs:263-269).
void check_launch(mi355_ctx *ctx, const char *what);
bool rccl_available();  // comm.cpp
int comm_world_size(const mi355_comm *comm);   // comm.cpp (the struct is private to it)
// pool.cpp
int32_t pool_alloc(mi355_ctx *ctx, hipStream_t stream, uint64_t bytes, void **out);
int32_t pool_free(mi355_ctx *ctx, hipStream_t stream, void *ptr);
void pool_release_graph(mi355_ctx *ctx, uint64_t graph_id);   // a graph died: its pinned blocks go back to the free lists
void scratch_release(mi355_ctx *ctx, void *ptr);               // a graph died: drop one pin of a library scratch buffer
// library-owned per-(stream, kind) device scratch (runtime.cpp): split-K slabs, re-laid-out GEMM operands, MX scales
int32_t scratch_get(mi355_ctx *ctx, hipStream_t s, int kind, size_t bytes, void **out);
// arrival-ticket words in library-owned device memory, one 16 KiB slot per live stream (runtime.cpp): word 0 the reductions' top ticket (their 32 group tickets from byte 2048 on),
// words 16 ... 511 gemm_nnrows.hip's per-strip tickets
int32_t ticket_for_stream(mi355_ctx *ctx, hipStream_t s, unsigned int **out);
int32_t strip_tickets_for_stream(mi355_ctx *ctx, hipStream_t s, unsigned int **out);
void strip_tickets_mark_dirty(mi355_ctx *ctx);
int32_t pool_cleanup(mi355_ctx *ctx, int32_t explicit_);
void pool_destroy(mi355_ctx *ctx);
// More than 64 KiB of dynamic LDS needs a per-kernel opt-in; once per (context = device, kernel), keyed by the kernel's
// host stub so that no two instantiations can ever share a bookkeeping slot.
inline void lds_opt_in(mi355_ctx *ctx, const void *kernel, int lds_bytes)
{
    if (ctx->lds_opted.insert(kernel).second)
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}
inline hipStream_t stream_of(mi355_ctx *ctx, mi355_stream s)
{
    return s ? reinterpret_cast<hipStream_t>(s) : ctx->compute_stream;
}
inline bool is_fp8(int32_t dtype) { return dtype == MI355_DTYPE_F8E4M3 || dtype == MI355_DTYPE_F8E5M2; }
inline size_t dtype_size(int32_t dtype)
{
    switch (dtype) {
    case MI355_DTYPE_F32: case MI355_DTYPE_I32: case MI355_DTYPE_U32: return 4;
    case MI355_DTYPE_BF16: case MI355_DTYPE_F16: return 2;
    case MI355_DTYPE_F64: case MI355_DTYPE_I64: case MI355_DTYPE_U64: return 8;
    case MI355_DTYPE_U8: case MI355_DTYPE_I8: case MI355_DTYPE_F8E4M3: case MI355_DTYPE_F8E5M2: return 1;
    case MI355_DTYPE_F4E2M1X2: case MI355_DTYPE_UE8M0: return 1;   // (a packed fp4 PAIR is one byte)
    default: return 0;
    }
}

}  // namespace mi355

// Entry of every call that touches a context: argument check, the context's lock (held to the end of the calling
// function; recursive, entry points call each other), and the device selection the reference does once per runner thread.
#define MI355_REQUIRE_CTX(ctx)                                          \
    if (!(ctx)) return MI355_E_INVALID_ARGUMENT;                        \
    std::lock_guard<std::recursive_mutex> _mi355_ctx_guard((ctx)->mu);  \
    {                                                                   \
        hipError_t _e = hipSetDevice((ctx)->device);                    \
        if (_e != hipSuccess)                                           \
            return mi355::fail((ctx), MI355_E_NO_DEVICE, "hipSetDevice(%d): %s", (ctx)->device, \
                               hipGetErrorString(_e));                  \
    }
#define MI355_LOCK_CTX(ctx) std::lock_guard<std::recursive_mutex> _mi355_ctx_guard((ctx)->mu)

#define MI355_HIP(ctx, expr)                                                               \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return mi355::fail((ctx), mi355::map_hip_error(_e), "%s: %s", #expr,           \
                               hipGetErrorString(_e));                                     \
    } while (0)
