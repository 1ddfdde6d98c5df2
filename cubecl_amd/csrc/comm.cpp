// comm.cpp -- ServerCommunication over RCCL (xGMI inside a node).
//
// The reference implements collectives only for CUDA/NCCL
// (crates/cubecl-cuda/src/compute/server.rs:666-926; HIP has SERVER_COMM_ENABLED = false,
// crates/cubecl-hip/src/compute/server.rs:657-659).  This is the MI355X counterpart: one
// communicator per (device set, rank), a dedicated communication stream, and the two event
// fences of the reference (compute -> comm before the collective, comm -> compute in
// sync_collective).  librccl is resolved with dlopen at first use so that the library itself
// loads (and its non-collective paths work) on hosts without RCCL.
//
// Message sizes on this path are tiny (one f32 partial, or 8 x 16-byte argmax records): the
// collectives are latency-bound, not xGMI-bandwidth-bound.
#include "internal.hpp"

#include <cstdlib>
#include <dlfcn.h>
#include <unistd.h>

using namespace mi355;

namespace {

typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;

struct rccl_api {
    void *handle = nullptr;
    bool tried = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

rccl_api g_rccl;
std::mutex g_rccl_mutex;

bool load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.tried) return g_rccl.handle != nullptr;
    g_rccl.tried = true;
    // By soname first: a process that already has a librccl mapped (PyTorch bundles its own) gets THAT copy, so one RCCL serves
    // everybody.  MI355_RCCL_LIBRARY names another file outright (a site build of RCCL; the test suite's stand-in).
    const char *names[] = {getenv("MI355_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return false;
#define LOAD(field, sym)                                                       \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));   \
    if (!g_rccl.field) { dlclose(h); return false; }
    LOAD(GetUniqueId, "ncclGetUniqueId")
    LOAD(CommInitRank, "ncclCommInitRank")
    LOAD(CommDestroy, "ncclCommDestroy")
    LOAD(AllReduce, "ncclAllReduce")
    LOAD(AllGather, "ncclAllGather")
    LOAD(Send, "ncclSend")
    LOAD(Recv, "ncclRecv")
    LOAD(GetErrorString, "ncclGetErrorString")
#undef LOAD
    g_rccl.handle = h;
    return true;
}

// get_nccl_dtype_count (crates/cubecl-cuda/src/compute/communication.rs:34-108)
bool to_nccl_dtype(int32_t dtype, ncclDataType_t *out)
{
    switch (dtype) {
    case MI355_DTYPE_I8: *out = 0; return true;
    case MI355_DTYPE_U8: *out = 1; return true;
    case MI355_DTYPE_I32: *out = 2; return true;
    case MI355_DTYPE_U32: *out = 3; return true;
    case MI355_DTYPE_I64: *out = 4; return true;
    case MI355_DTYPE_U64: *out = 5; return true;
    case MI355_DTYPE_F16: *out = 6; return true;
    case MI355_DTYPE_F32: *out = 7; return true;
    case MI355_DTYPE_F64: *out = 8; return true;
    case MI355_DTYPE_BF16: *out = 9; return true;
    default: return false;
    }
}

// to_nccl_op (communication.rs:27-32) + Max / Min / Prod (ncclSum 0, ncclProd 1, ncclMax 2, ncclMin 3, ncclAvg 4); the index
// reductions MI355_REDUCE_ARGMAX / ARGMIN have no collective form (the sharded argmax all-gathers records: sharded.py)
bool to_nccl_op(int32_t op, ncclRedOp_t *out)
{
    switch (op) {
    case MI355_REDUCE_SUM: *out = 0; return true;
    case MI355_REDUCE_MEAN: *out = 4; return true;
    case MI355_REDUCE_MAX: *out = 2; return true;
    case MI355_REDUCE_MIN: *out = 3; return true;
    case MI355_REDUCE_PROD: *out = 1; return true;
    default: return false;
    }
}

}  // namespace

struct mi355_comm {
    ncclComm_t comm;
    int rank;
    int world;
};

namespace mi355 {
// Cheap presence probe for the property block (dlopen of the 570 MB library is deferred to the
// first collective).
bool rccl_available()
{
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        if (g_rccl.tried) return g_rccl.handle != nullptr;
    }
    if (void *h = dlopen("librccl.so.1", RTLD_LAZY | RTLD_NOLOAD)) { dlclose(h); return true; }
    const char *paths[] = {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "/usr/lib/librccl.so.1"};
    for (const char *p : paths)
        if (access(p, R_OK) == 0) return true;
    return false;
}
}  // namespace mi355

#define MI355_NCCL(ctx, expr)                                                                   \
    do {                                                                                        \
        ncclResult_t _r = (expr);                                                               \
        if (_r != 0) return fail((ctx), MI355_E_COMM, "%s: %s", #expr, g_rccl.GetErrorString(_r)); \
    } while (0)

// compute -> comm fence: the collective must see everything already queued on the data's stream
// (crates/cubecl-cuda/src/compute/server.rs:749, :764-772)
static int32_t fence_compute_to_comm(mi355_ctx *ctx, hipStream_t compute)
{
    MI355_HIP(ctx, hipEventRecord(ctx->fence_a, compute));
    MI355_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->fence_a, 0));
    ctx->comm_dirty = true;
    return MI355_OK;
}

// Orders stream `waiter` behind the last inline collective when that ran on another stream (no-op otherwise).
static int32_t fence_inline_to(mi355_ctx *ctx, hipStream_t waiter)
{
    if (!ctx->inline_dirty || ctx->inline_stream == waiter) return MI355_OK;
    MI355_HIP(ctx, hipEventRecord(ctx->fence_c, ctx->inline_stream));
    MI355_HIP(ctx, hipStreamWaitEvent(waiter, ctx->fence_c, 0));
    return MI355_OK;
}

// The communication stream takes the next operation: behind everything queued on `compute` (the reference's compute -> comm fence) and
// behind an inline collective still running on ANOTHER compute stream; one on `compute` itself is covered by the first fence.  From
// here on the communication stream carries that order (mi355_sync_collective fences it towards whoever asks).
static int32_t onto_comm_stream(mi355_ctx *ctx, hipStream_t compute)
{
    if (ctx->inline_dirty) {
        if (ctx->inline_stream != compute) {
            const int32_t rc = fence_inline_to(ctx, ctx->comm_stream);
            if (rc != MI355_OK) return rc;
        }
        ctx->inline_dirty = false;
    }
    return fence_compute_to_comm(ctx, compute);
}

// Which stream a collective of `bytes` (the larger of what it reads and writes on this rank) runs on.  The reference's shape
// -- a communication stream between two event fences -- lets compute run beside a transfer; a collective of a few bytes has
// no transfer to hide, and the fences are most of its time (one all-gather of a 16-byte record + the combine kernel behind
// it, 1 rank on real RCCL: 31.8 us through the fences, see profiles/r05_c4_shard.md for the inline figure).  So a message of at
// most MI355_COMM_INLINE_BYTES (default 4096) is queued IN the compute stream's order: no fence before it, and for THAT stream
// nothing for mi355_sync_collective to do after it.  The guarantee of the reference's sync_collective (crates/cubecl-cuda/src/
// compute/server.rs:782-797) -- any stream named there is ordered behind every collective issued before -- is kept for OTHER
// streams through `inline_stream` / `fence_c`: the library remembers which stream carries the last inline collective;
// mi355_sync_collective(another stream), a later collective on another stream (inline or on the communication stream) and
// mi355_comm_destroy order themselves behind it.  A communicator therefore never has operations in flight on two streams at
// once, and RCCL sees them in call order.
static hipStream_t collective_stream(mi355_ctx *ctx, hipStream_t compute, uint64_t bytes, int32_t *rc)
{
    static const uint64_t inline_bytes = [] { const char *e = getenv("MI355_COMM_INLINE_BYTES"); return e ? (uint64_t)strtoull(e, nullptr, 10) : 4096ull; }();
    *rc = MI355_OK;
    if (bytes <= inline_bytes && !ctx->comm_dirty) {
        if ((*rc = fence_inline_to(ctx, compute)) != MI355_OK) return compute;    // an inline collective still running on ANOTHER compute stream
        ctx->inline_stream = compute;
        ctx->inline_dirty = true;
        return compute;
    }
    *rc = onto_comm_stream(ctx, compute);
    return ctx->comm_stream;
}

MI355_API int32_t mi355_comm_unique_id(uint8_t id[MI355_UNIQUE_ID_BYTES])
{
    if (!id) return MI355_E_INVALID_ARGUMENT;
    if (!load_rccl()) return fail(nullptr, MI355_E_COMM, "librccl.so not found");
    ncclUniqueId uid;
    ncclResult_t r = g_rccl.GetUniqueId(&uid);
    if (r != 0) return fail(nullptr, MI355_E_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    memcpy(id, uid.internal, MI355_UNIQUE_ID_BYTES);
    return MI355_OK;
}

MI355_API int32_t mi355_comm_init(mi355_ctx *ctx, const uint8_t id[MI355_UNIQUE_ID_BYTES], int32_t rank,
                                  int32_t world_size, mi355_comm **out_comm)
{
    MI355_REQUIRE_CTX(ctx);
    if (!id || !out_comm) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_comm_init: NULL argument");
    *out_comm = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_comm_init: rank %d of %d", rank, world_size);
    if (!load_rccl()) return fail(ctx, MI355_E_COMM, "librccl.so not found");
    ncclUniqueId uid;
    memcpy(uid.internal, id, MI355_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    MI355_NCCL(ctx, g_rccl.CommInitRank(&comm, world_size, uid, rank));
    mi355_comm *c = new mi355_comm{comm, rank, world_size};
    *out_comm = c;
    return MI355_OK;
}

MI355_API int32_t mi355_comm_destroy(mi355_ctx *ctx, mi355_comm *comm)
{
    MI355_REQUIRE_CTX(ctx);
    if (!comm) return MI355_OK;
    hipStreamSynchronize(ctx->comm_stream);
    if (ctx->inline_dirty && ctx->inline_stream) {          // collectives queued in a compute stream's order (collective_stream)
        hipStreamSynchronize(ctx->inline_stream);
        ctx->inline_dirty = false;
    }
    if (comm->comm) g_rccl.CommDestroy(comm->comm);
    delete comm;
    return MI355_OK;
}

MI355_API int32_t mi355_all_reduce(mi355_ctx *ctx, mi355_comm *comm, mi355_stream compute_stream, const void *src,
                                   void *dst, uint64_t count, int32_t dtype, int32_t op)
{
    MI355_REQUIRE_CTX(ctx);
    if (!comm) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_all_reduce: communicator is NULL (call mi355_comm_init)");
    if (count == 0) return MI355_OK;
    if (!src || !dst) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_all_reduce: NULL buffer");
    ncclDataType_t dt;
    ncclRedOp_t rop;
    if (!to_nccl_dtype(dtype, &dt)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_all_reduce: dtype %d not supported by RCCL", dtype);
    if (!to_nccl_op(op, &rop)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_all_reduce: unknown op %d", op);
    int32_t rc;
    hipStream_t on = collective_stream(ctx, stream_of(ctx, compute_stream), count * dtype_size(dtype), &rc);
    if (rc != MI355_OK) return rc;
    MI355_NCCL(ctx, g_rccl.AllReduce(src, dst, count, dt, rop, comm->comm, on));
    return MI355_OK;
}

MI355_API int32_t mi355_all_gather(mi355_ctx *ctx, mi355_comm *comm, mi355_stream compute_stream, const void *src,
                                   void *dst, uint64_t count, int32_t dtype)
{
    MI355_REQUIRE_CTX(ctx);
    if (!comm) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_all_gather: communicator is NULL");
    if (count == 0) return MI355_OK;
    if (!src || !dst) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_all_gather: NULL buffer");
    ncclDataType_t dt;
    if (!to_nccl_dtype(dtype, &dt)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_all_gather: dtype %d not supported by RCCL", dtype);
    int32_t rc;
    hipStream_t on = collective_stream(ctx, stream_of(ctx, compute_stream), count * dtype_size(dtype) * (uint64_t)comm->world, &rc);
    if (rc != MI355_OK) return rc;
    MI355_NCCL(ctx, g_rccl.AllGather(src, dst, count, dt, comm->comm, on));
    return MI355_OK;
}

MI355_API int32_t mi355_send(mi355_ctx *ctx, mi355_comm *comm, mi355_stream compute_stream, const void *src,
                             uint64_t count, int32_t dtype, int32_t peer)
{
    MI355_REQUIRE_CTX(ctx);
    if (!comm) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_send: communicator is NULL");
    if (count == 0) return MI355_OK;
    if (!src) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_send: NULL buffer");
    if (peer < 0 || peer >= comm->world || peer == comm->rank)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_send: bad peer %d", peer);
    ncclDataType_t dt;
    if (!to_nccl_dtype(dtype, &dt)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_send: dtype %d not supported by RCCL", dtype);
    int32_t rc = onto_comm_stream(ctx, stream_of(ctx, compute_stream));
    if (rc != MI355_OK) return rc;
    MI355_NCCL(ctx, g_rccl.Send(src, count, dt, peer, comm->comm, ctx->comm_stream));
    return MI355_OK;
}

MI355_API int32_t mi355_recv(mi355_ctx *ctx, mi355_comm *comm, mi355_stream compute_stream, void *dst, uint64_t count,
                             int32_t dtype, int32_t peer)
{
    MI355_REQUIRE_CTX(ctx);
    if (!comm) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_recv: communicator is NULL");
    if (count == 0) return MI355_OK;
    if (!dst) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_recv: NULL buffer");
    if (peer < 0 || peer >= comm->world || peer == comm->rank)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_recv: bad peer %d", peer);
    ncclDataType_t dt;
    if (!to_nccl_dtype(dtype, &dt)) return fail(ctx, MI355_E_UNSUPPORTED, "mi355_recv: dtype %d not supported by RCCL", dtype);
    int32_t rc = onto_comm_stream(ctx, stream_of(ctx, compute_stream));
    if (rc != MI355_OK) return rc;
    MI355_NCCL(ctx, g_rccl.Recv(dst, count, dt, peer, comm->comm, ctx->comm_stream));
    return MI355_OK;
}

// comm -> compute fence (server.rs:782-797)
MI355_API int32_t mi355_sync_collective(mi355_ctx *ctx, mi355_stream compute_stream)
{
    MI355_REQUIRE_CTX(ctx);
    hipStream_t s = stream_of(ctx, compute_stream);
    // a collective that ran inline on ANOTHER compute stream: this stream waits for it as it would for the communication stream
    // (the flag stays up: a third stream may still ask; the carrying stream itself never needs a fence)
    int32_t rc = fence_inline_to(ctx, s);
    if (rc != MI355_OK) return rc;
    if (!ctx->comm_dirty) return MI355_OK;
    MI355_HIP(ctx, hipEventRecord(ctx->fence_b, ctx->comm_stream));
    MI355_HIP(ctx, hipStreamWaitEvent(s, ctx->fence_b, 0));
    ctx->comm_dirty = false;
    return MI355_OK;
}

namespace mi355 {
int comm_world_size(const mi355_comm *comm) { return comm ? comm->world : 0; }
}  // namespace mi355
