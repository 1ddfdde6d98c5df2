// pool.cpp -- the memory pool behind ComputeClient::empty / create: stream-ordered caching allocator for one device.
//
// Role in the reference: MemoryManagement (crates/cubecl-runtime/src/memory_management/memory_manage.rs): `reserve`
// (:1084) hands out slices of pooled pages, `cleanup` (:938) returns unused pages to the driver (periodically, every
// reservation drives it; explicitly on request; never during a graph capture), `memory_usage` (:1238) reports
// {number_allocs, bytes_in_use, bytes_padding, bytes_reserved} (base.rs:8-28), `mode` (:900) switches to the
// persistent strategy for data that is never freed.  The default layout there is a ladder of sliced pages plus
// exact-size exclusive pages with a dealloc period of 5000 x (1 + size / 1 GiB) reservations (:585-655).
//
// Built for this device rather than translated:
//   * hipMalloc / hipFree are synchronising driver calls (~100 us and a device-wide fence for hipFree); a launch-bound
//     sequence must never see them.  Every request is served from a cache when one fits.
//   * Requests up to 32 MiB are SLICES of a size class (quarter-octave steps: padding < 25 %) carved by bump pointer
//     out of slab pages (2 MiB .. 256 MiB, 64 slices each); bigger requests are EXCLUSIVE pages rounded to 2 MiB (the
//     driver's large-fragment size) and cached by size when freed, reused when the cached page is at most 12.5 %
//     larger than the request.  With 288 GB per device fragmentation is cheap and driver calls are not, so nothing is
//     ever split or coalesced.
//   * Reuse is stream-ordered: a freed block remembers the stream it was freed on and an event recorded there.  The
//     same stream reuses it at once (its later work is ordered after everything that touched the block); another
//     stream only once the event has completed.  No device synchronisation anywhere on the reuse path.
//   * Persistent mode: exact-size exclusive pages that periodic cleanup never releases (weights, KV-like state).
#include <algorithm>
#include <map>
#include <unordered_map>
#include <vector>

#include "internal.hpp"

using namespace mi355;

namespace mi355 {

namespace {

constexpr size_t KiB = 1024, MiB = 1024 * KiB, GiB = 1024 * MiB;
constexpr size_t MIN_SLICE = 512;                 // first size class
constexpr int STEPS = 4;                          // size classes per octave
constexpr size_t MAX_SLICE = 32 * MiB;            // above this: exclusive pages
constexpr size_t EXCL_GRAIN = 2 * MiB;
constexpr uint64_t BASE_DEALLOC_PERIOD = 5000;    // memory_manage.rs:242

struct pool_page {
    void *base = nullptr;
    size_t bytes = 0, bump = 0;
    uint32_t live = 0;                            // slices handed out and not yet freed
    int bin = -1;
};

struct pool_block {
    void *ptr = nullptr;
    size_t size = 0;                              // rounded (what the block really spans)
    size_t requested = 0;
    int bin = -1;                                 // -1: exclusive page
    pool_page *page = nullptr;                    // slab page of a slice
    hipStream_t stream = nullptr;                 // stream it was freed on
    hipEvent_t event = nullptr;                   // recorded at free time
    uint64_t freed_tick = 0;
    bool persistent = false;
    uint64_t graph_id = 0;                        // allocated inside this capture window: pinned while that graph lives
    uint64_t free_graph_id = 0;                   // freed inside this capture window: a second, independent pin
    bool idle = false;                            // released by a destroyed graph (its replays were waited for): any stream may take it
};

size_t bin_size(int b)
{
    const size_t base = MIN_SLICE << (b / STEPS);
    return base + (base / STEPS) * (size_t)(b % STEPS);
}
int bin_of(size_t bytes)                           // smallest class >= bytes
{
    int b = 0;
    while (bin_size(b) < bytes) ++b;
    return b;
}
size_t page_bytes_for(int b) { return std::min<size_t>(std::max<size_t>(64 * bin_size(b), 2 * MiB), 256 * MiB); }

}  // namespace

struct memory_pool {
    std::unordered_map<void *, pool_block> live;
    std::vector<std::vector<pool_block>> bins;
    std::multimap<size_t, pool_block> big_free;    // exclusive pages waiting for reuse, by size
    std::vector<pool_page *> pages;
    std::vector<hipEvent_t> spare_events;
    uint64_t tick = 0;                             // reservations so far (drives the dealloc periods)
    int mode = MI355_ALLOC_MODE_AUTO;
    uint64_t n_allocs = 0, bytes_in_use = 0, bytes_padding = 0, bytes_reserved = 0;
    uint64_t driver_allocs = 0, driver_frees = 0, cache_hits = 0;
    std::vector<pool_block> held;                  // blocks freed while pinned by one or two graphs (out of the free lists)
};

namespace {

memory_pool *pool_of(mi355_ctx *ctx)
{
    if (!ctx->pool) {
        ctx->pool = new memory_pool();
        ctx->pool->bins.resize(bin_of(MAX_SLICE) + 1);
    }
    return ctx->pool;
}

hipEvent_t take_event(memory_pool *p)
{
    if (!p->spare_events.empty()) {
        hipEvent_t e = p->spare_events.back();
        p->spare_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return e;
}

// May `stream` start using a block that was freed on `b.stream`?
bool reusable(const mi355_ctx *ctx, const pool_block &b, hipStream_t stream)
{
    if (b.idle || b.stream == stream) return true;
    if (!b.event || ctx->capturing) return false;  // (no event queries while this thread has a capture open)
    const hipError_t e = hipEventQuery(b.event);
    if (e == hipSuccess) return true;
    (void)hipGetLastError();                       // hipErrorNotReady is not an error here
    return false;
}

// Is the graph (or open window) with this id still able to replay?
bool pin_alive(const mi355_ctx *ctx, uint64_t id)
{
    return id != 0 && (id == ctx->capture_id || ctx->live_graphs.count(id) != 0);
}

void retire_event(memory_pool *p, pool_block &b)
{
    if (b.event) p->spare_events.push_back(b.event);
    b.event = nullptr;
}

int32_t driver_alloc(mi355_ctx *ctx, memory_pool *p, size_t bytes, void **out)
{
    if (ctx->capturing)
        return fail(ctx, MI355_E_UNSUPPORTED, "memory pool: a fresh device allocation of %llu bytes is needed inside a graph "
                    "capture window; warm the sequence up once before capturing", (unsigned long long)bytes);
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {                // release every cached page, then one retry (command.rs:142-161)
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        pool_cleanup(ctx, 1);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        if (e == hipErrorOutOfMemory)
            return fail(ctx, MI355_E_OUT_OF_MEMORY, "memory pool: out of device memory allocating %llu bytes", (unsigned long long)bytes);
        return fail(ctx, MI355_E_EXECUTION, "memory pool: hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    ++p->driver_allocs;
    p->bytes_reserved += bytes;
    return MI355_OK;
}

void account_alloc(memory_pool *p, const pool_block &b)
{
    ++p->n_allocs;
    p->bytes_in_use += b.requested;
    p->bytes_padding += b.size - b.requested;
}

}  // namespace

int32_t pool_alloc(mi355_ctx *ctx, hipStream_t stream, uint64_t bytes, void **out)
{
    memory_pool *p = pool_of(ctx);
    *out = nullptr;
    if (bytes == 0) return MI355_OK;
    if (bytes > ctx->props.max_page_size)
        return fail(ctx, MI355_E_BUFFER_TOO_BIG, "allocation of %llu bytes exceeds max_page_size %llu", (unsigned long long)bytes,
                    (unsigned long long)ctx->props.max_page_size);
    ++p->tick;
    if ((p->tick & 1023u) == 0) pool_cleanup(ctx, 0);           // drive the periodic release (memory_manage.rs:1089-1094)

    pool_block blk;
    const bool persistent = p->mode == MI355_ALLOC_MODE_PERSISTENT;
    if (!persistent && bytes <= MAX_SLICE) {
        const int b = bin_of(bytes);
        auto &fl = p->bins[b];
        for (size_t i = fl.size(); i-- > 0;) {                   // most recently freed first (still warm in L2 / MALL)
            if (!reusable(ctx, fl[i], stream)) continue;
            blk = fl[i];
            fl.erase(fl.begin() + (long)i);
            retire_event(p, blk);
            ++p->cache_hits;
            goto have_slice;
        }
        {
            pool_page *pg = nullptr;
            for (pool_page *q : p->pages)
                if (q->bin == b && q->bump + bin_size(b) <= q->bytes) { pg = q; break; }
            if (!pg) {
                void *base = nullptr;
                const size_t pb = page_bytes_for(b);
                const int32_t rc = driver_alloc(ctx, p, pb, &base);
                if (rc != MI355_OK) return rc;
                pg = new pool_page();
                pg->base = base; pg->bytes = pb; pg->bin = b;
                p->pages.push_back(pg);
            }
            blk = pool_block();
            blk.ptr = static_cast<char *>(pg->base) + pg->bump;
            blk.size = bin_size(b);
            blk.bin = b;
            blk.page = pg;
            pg->bump += blk.size;
        }
    have_slice:
        blk.requested = bytes;
        ++blk.page->live;
    } else {
        // exclusive page: exact size in persistent mode (256-byte granules), 2 MiB granules otherwise
        const size_t grain = persistent ? 256 : EXCL_GRAIN;
        const size_t want = (bytes + grain - 1) / grain * grain;
        const size_t slack = persistent ? 0 : want / 8;
        bool found = false;
        for (auto it = p->big_free.lower_bound(want); it != p->big_free.end() && it->first <= want + slack; ++it) {
            if (it->second.persistent != persistent || !reusable(ctx, it->second, stream)) continue;
            blk = it->second;
            p->big_free.erase(it);
            retire_event(p, blk);
            ++p->cache_hits;
            found = true;
            break;
        }
        if (!found) {
            void *base = nullptr;
            const int32_t rc = driver_alloc(ctx, p, want, &base);
            if (rc != MI355_OK) return rc;
            blk = pool_block();
            blk.ptr = base;
            blk.size = want;
            blk.persistent = persistent;
        }
        blk.requested = bytes;
    }
    blk.idle = false;
    // An allocation made inside a capture window is baked into the graph's nodes: it stays pinned (never handed to anyone
    // else) for as long as that graph lives, whenever its owner drops it.
    blk.graph_id = (ctx->capturing && stream == ctx->capture_stream) ? ctx->capture_id : 0;
    blk.free_graph_id = 0;
    account_alloc(p, blk);
    p->live[blk.ptr] = blk;
    *out = blk.ptr;
    return MI355_OK;
}

int32_t pool_free(mi355_ctx *ctx, hipStream_t stream, void *ptr)
{
    if (!ptr) return MI355_OK;
    memory_pool *p = pool_of(ctx);
    auto it = p->live.find(ptr);
    if (it == p->live.end())
        return fail(ctx, MI355_E_NOT_FOUND, "memory pool: %p is not a live allocation of this context", ptr);
    pool_block blk = it->second;
    p->live.erase(it);
    --p->n_allocs;
    p->bytes_in_use -= blk.requested;
    p->bytes_padding -= blk.size - blk.requested;
    // A collective still in flight on the communication stream may read or write this block (a temporary handed to
    // all_reduce / send and dropped before sync_collective): order the freeing stream behind the communication stream
    // first, so that same-stream reuse and the event recorded below both cover it.
    // "Inside the window" means: on the stream that is being captured.  Another lane keeps running real work while the
    // window is open, and a block it frees is ordered by an event like any other.
    const bool in_window = ctx->capturing && stream == ctx->capture_stream;
    if (ctx->comm_dirty && !in_window && ctx->fence_b && ctx->comm_stream) {
        if (hipEventRecord(ctx->fence_b, ctx->comm_stream) != hipSuccess || hipStreamWaitEvent(stream, ctx->fence_b, 0) != hipSuccess)
            (void)hipGetLastError();
    }
    // ... and so may a small collective queued in ANOTHER compute stream's order (comm.cpp collective_stream)
    if (ctx->inline_dirty && !in_window && ctx->inline_stream && ctx->inline_stream != stream && ctx->fence_c) {
        if (hipEventRecord(ctx->fence_c, ctx->inline_stream) != hipSuccess || hipStreamWaitEvent(stream, ctx->fence_c, 0) != hipSuccess)
            (void)hipGetLastError();
    }
    blk.stream = stream;
    blk.freed_tick = p->tick;
    blk.event = nullptr;
    if (!in_window) {                                             // (an event recorded inside a capture is a graph node)
        blk.event = take_event(p);
        if (blk.event && hipEventRecord(blk.event, stream) != hipSuccess) {
            (void)hipGetLastError();
            p->spare_events.push_back(blk.event);
            blk.event = nullptr;
        }
    }
    // Freed inside a capture window (the captured kernels before this point use it), and / or allocated inside the window
    // of a graph that is still alive: the address is part of one or two replayable graphs, so the block stays out of the
    // free lists until the LAST of them is gone (mi355_graph_destroy).  Both pins are kept: a block allocated in window A
    // and freed in a later window B must outlive A as well as B.  (A slab slice keeps its page's live count: the page must
    // not be released either.)
    blk.free_graph_id = in_window ? ctx->capture_id : 0;
    if (pin_alive(ctx, blk.graph_id) || pin_alive(ctx, blk.free_graph_id)) {
        p->held.push_back(blk);
        return MI355_OK;
    }
    blk.graph_id = blk.free_graph_id = 0;
    if (blk.bin >= 0) {
        --blk.page->live;
        p->bins[blk.bin].push_back(blk);
    } else {
        p->big_free.emplace(blk.size, blk);
    }
    return MI355_OK;
}

// The graph with this id is gone (destroyed after its replays were waited for, or its capture failed): what ONLY it
// pinned is ordinary free memory again; a block another live graph (or the open window) still pins stays held.  Blocks
// freed at capture time carry no event -- nothing but the graphs ever used them after that point -- so they are marked idle.
void pool_release_graph(mi355_ctx *ctx, uint64_t graph_id)
{
    memory_pool *p = ctx->pool;
    if (!p) return;
    std::vector<pool_block> keep;
    for (pool_block &blk : p->held) {
        if (blk.graph_id == graph_id) blk.graph_id = 0;
        if (blk.free_graph_id == graph_id) blk.free_graph_id = 0;
        if (pin_alive(ctx, blk.graph_id) || pin_alive(ctx, blk.free_graph_id)) {
            keep.push_back(blk);
            continue;
        }
        blk.graph_id = blk.free_graph_id = 0;
        if (!blk.event) blk.idle = true;
        if (blk.bin >= 0) {
            --blk.page->live;
            p->bins[blk.bin].push_back(blk);
        } else {
            p->big_free.emplace(blk.size, blk);
        }
    }
    p->held.swap(keep);
}

// explicit == 0: release exclusive pages that sat unused for their dealloc period; explicit != 0: release everything
// that is not in use (cached exclusive pages and slab pages without a live slice).  A block whose free-time event has
// not completed yet is kept for the next round: hipFree of memory a kernel still uses is never issued.
int32_t pool_cleanup(mi355_ctx *ctx, int32_t explicit_)
{
    memory_pool *p = ctx->pool;
    if (!p || ctx->capturing) return MI355_OK;                    // nothing may be freed during a capture (:948-952)
    if (explicit_) (void)hipDeviceSynchronize();                  // "release the memory now": everything freed is idle after this
    auto done = [explicit_](const pool_block &b) {
        if (!b.event) return explicit_ != 0 || b.idle;        // idle: released by a dead graph whose replays were waited for
        if (hipEventQuery(b.event) == hipSuccess) return true;
        (void)hipGetLastError();
        return false;
    };
    for (auto it = p->big_free.begin(); it != p->big_free.end();) {
        pool_block &b = it->second;
        const uint64_t period = BASE_DEALLOC_PERIOD * (1 + (uint64_t)((double)b.size / (double)GiB + 0.5));
        const bool due = explicit_ ? true : (!b.persistent && p->tick - b.freed_tick >= period);
        if (due && done(b)) {
            (void)hipFree(b.ptr);
            ++p->driver_frees;
            p->bytes_reserved -= b.size;
            retire_event(p, b);
            it = p->big_free.erase(it);
        } else {
            ++it;
        }
    }
    if (explicit_) {
        for (size_t i = 0; i < p->pages.size();) {
            pool_page *pg = p->pages[i];
            if (pg->live != 0) { ++i; continue; }
            auto &fl = p->bins[pg->bin];
            bool all_done = true;
            for (const pool_block &b : fl)
                if (b.page == pg && !done(b)) { all_done = false; break; }
            if (!all_done) { ++i; continue; }
            for (size_t k = fl.size(); k-- > 0;)
                if (fl[k].page == pg) {
                    retire_event(p, fl[k]);
                    fl.erase(fl.begin() + (long)k);
                }
            (void)hipFree(pg->base);
            ++p->driver_frees;
            p->bytes_reserved -= pg->bytes;
            delete pg;
            p->pages.erase(p->pages.begin() + (long)i);
        }
    }
    return MI355_OK;
}

void pool_destroy(mi355_ctx *ctx)
{
    memory_pool *p = ctx->pool;
    if (!p) return;
    for (auto &kv : p->live)
        if (kv.second.bin < 0) (void)hipFree(kv.second.ptr);     // leaked exclusive pages; slices die with their page
    for (auto &kv : p->big_free) {
        (void)hipFree(kv.second.ptr);
        if (kv.second.event) (void)hipEventDestroy(kv.second.event);
    }
    for (auto &b : p->held) {
        if (b.bin < 0) (void)hipFree(b.ptr);
        if (b.event) (void)hipEventDestroy(b.event);
    }
    for (auto &fl : p->bins)
        for (auto &b : fl)
            if (b.event) (void)hipEventDestroy(b.event);
    for (pool_page *pg : p->pages) {
        (void)hipFree(pg->base);
        delete pg;
    }
    for (hipEvent_t e : p->spare_events) (void)hipEventDestroy(e);
    delete p;
    ctx->pool = nullptr;
}

}  // namespace mi355

MI355_API int32_t mi355_pool_alloc(mi355_ctx *ctx, mi355_stream stream, uint64_t bytes, void **out_dptr)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out_dptr) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_pool_alloc: out_dptr is NULL");
    return pool_alloc(ctx, stream_of(ctx, stream), bytes, out_dptr);
}

MI355_API int32_t mi355_pool_free(mi355_ctx *ctx, mi355_stream stream, void *dptr)
{
    MI355_REQUIRE_CTX(ctx);               // (events are created / recorded on this context's device)
    return pool_free(ctx, stream_of(ctx, stream), dptr);
}

MI355_API int32_t mi355_pool_cleanup(mi355_ctx *ctx, int32_t explicit_cleanup)
{
    MI355_REQUIRE_CTX(ctx);
    return pool_cleanup(ctx, explicit_cleanup);
}

MI355_API int32_t mi355_pool_mode(mi355_ctx *ctx, int32_t mode)
{
    MI355_REQUIRE_CTX(ctx);
    if (mode != MI355_ALLOC_MODE_AUTO && mode != MI355_ALLOC_MODE_PERSISTENT)
        return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_pool_mode: unknown mode %d", mode);
    pool_of(ctx)->mode = mode;
    return MI355_OK;
}

MI355_API int32_t mi355_pool_usage(mi355_ctx *ctx, mi355_memory_usage *out)
{
    MI355_REQUIRE_CTX(ctx);
    if (!out) return fail(ctx, MI355_E_INVALID_ARGUMENT, "mi355_pool_usage: out is NULL");
    memory_pool *p = pool_of(ctx);
    out->number_allocs = p->n_allocs;
    out->bytes_in_use = p->bytes_in_use;
    out->bytes_padding = p->bytes_padding;
    out->bytes_reserved = p->bytes_reserved;
    out->driver_allocs = p->driver_allocs;
    out->driver_frees = p->driver_frees;
    out->cache_hits = p->cache_hits;
    out->reserved = 0;
    return MI355_OK;
}
