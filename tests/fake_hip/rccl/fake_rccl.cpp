// A single-process stand-in for librccl.so.1 (TEST INFRASTRUCTURE ONLY; tests/test_comm_cpu.py puts its directory on
// LD_LIBRARY_PATH of a child process so that comm.cpp's dlopen("librccl.so.1") finds it).  Ranks are contexts of one
// process, the reference's own model (crates/cubecl-core/src/runtime_tests/all_reduce.rs: one client per device, one
// thread); a collective executes when its last rank has called, on the host memory the fake HIP runtime hands out.
// Every entry point takes one lock, so ranks may also be driven from one thread each (the model comm_init assumes:
// crates/cubecl-cuda/src/compute/server.rs:669-703 runs on each device's own runner thread).  With FAKE_RCCL_BLOCKING_INIT=1
// ncclCommInitRank behaves like the real one: it returns only when every rank of the communicator has called it.
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

typedef int ncclResult_t;                      // 0 = success, 4 = invalid argument, 5 = invalid usage
struct hipStream_opaque;
typedef hipStream_opaque *hipStream_t;
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
struct call { const void *send = nullptr; void *recv = nullptr; size_t count = 0; int dtype = 0, op = 0; bool set = false; };
struct p2p { const void *send = nullptr; void *recv = nullptr; size_t count = 0; int dtype = 0; };
struct group {
    int nranks = 0, joined = 0;
    std::vector<call> reduce, gather;
    unsigned long long gen_reduce = 0, gen_gather = 0;   // completed collectives (FAKE_RCCL_BLOCKING: a rank waits for its own to complete)
    std::map<std::pair<int, int>, p2p> wires;   // (from, to)
};
std::map<std::string, group> groups;
int next_id = 1;
std::mutex mu;
std::condition_variable cv;
#define LOCKED std::unique_lock<std::mutex> lock(mu)
// FAKE_RCCL_BLOCKING=1 (one THREAD per rank, each issuing collectives at its own pace -- bench.py --threads): a call returns when the
// collective it belongs to has executed, and a rank that is a collective ahead of its peers waits for them instead of being refused.
// Without it a call returns at once ("enqueued") and the last rank to arrive executes -- what a single thread driving every rank needs.
bool blocking() { static const bool b = [] { const char *e = getenv("FAKE_RCCL_BLOCKING"); return e && e[0] == '1'; }(); return b; }
size_t size_of(int dt) { const size_t s[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2}; return dt >= 0 && dt < 10 ? s[dt] : 0; }

template <typename T> void reduce_typed(group &g, size_t n, int op)
{
    std::vector<T> out(n);
    for (size_t i = 0; i < n; ++i) {
        T acc = static_cast<const T *>(g.reduce[0].send)[i];
        for (int r = 1; r < g.nranks; ++r) {
            const T v = static_cast<const T *>(g.reduce[r].send)[i];
            acc = op == 2 ? (v > acc ? v : acc) : op == 3 ? (v < acc ? v : acc) : op == 1 ? (T)(acc * v) : (T)(acc + v);     // max, min, prod, sum / avg
        }
        out[i] = op == 4 ? (T)(acc / (T)g.nranks) : acc;
    }
    for (int r = 0; r < g.nranks; ++r) memcpy(g.reduce[r].recv, out.data(), n * sizeof(T));
}
}  // namespace

struct ncclComm { group *g; int rank; };
typedef ncclComm *ncclComm_t;

extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    LOCKED;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "fake-rccl-%d", next_id++);
    return 0;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    LOCKED;
    group &g = groups[std::string(id.internal, sizeof id.internal)];
    if (g.nranks == 0) { g.nranks = nranks; g.reduce.resize(nranks); g.gather.resize(nranks); }
    if (g.nranks != nranks || rank < 0 || rank >= nranks) return 4;
    *comm = new ncclComm{&g, rank};
    ++g.joined;
    cv.notify_all();
    const char *blocking = getenv("FAKE_RCCL_BLOCKING_INIT");
    if (blocking && blocking[0] == '1' &&
        !cv.wait_for(lock, std::chrono::seconds(30), [&g] { return g.joined >= g.nranks; }))
        return 5;                                                // a rank never showed up
    return 0;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return 0; }
__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : r == 4 ? "invalid argument" : "invalid usage"; }

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t comm, hipStream_t)
{
    LOCKED;
    group &g = *comm->g;
    if (g.reduce[comm->rank].set) {                              // the same rank twice before the others arrived
        if (!blocking() || !cv.wait_for(lock, std::chrono::seconds(60), [&] { return !g.reduce[comm->rank].set; })) return 5;
    }
    g.reduce[comm->rank] = {send, recv, count, dtype, op, true};
    const unsigned long long mine = g.gen_reduce;
    for (const call &c : g.reduce)
        if (!c.set) {                                            // not everyone is here yet: enqueued
            if (blocking() && !cv.wait_for(lock, std::chrono::seconds(60), [&] { return g.gen_reduce != mine; })) return 5;
            return 0;
        }
    for (const call &c : g.reduce) if (c.count != count || c.dtype != dtype || c.op != op) return 4;
    switch (dtype) {
    case 2: reduce_typed<int32_t>(g, count, op); break;
    case 3: reduce_typed<uint32_t>(g, count, op); break;
    case 4: reduce_typed<int64_t>(g, count, op); break;
    case 5: reduce_typed<uint64_t>(g, count, op); break;
    case 7: reduce_typed<float>(g, count, op); break;
    case 8: reduce_typed<double>(g, count, op); break;
    default: return 4;
    }
    for (call &c : g.reduce) c.set = false;
    ++g.gen_reduce;
    cv.notify_all();
    return 0;
}
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t comm, hipStream_t)
{
    LOCKED;
    group &g = *comm->g;
    if (g.gather[comm->rank].set) {
        if (!blocking() || !cv.wait_for(lock, std::chrono::seconds(60), [&] { return !g.gather[comm->rank].set; })) return 5;
    }
    g.gather[comm->rank] = {send, recv, count, dtype, 0, true};
    const unsigned long long mine = g.gen_gather;
    for (const call &c : g.gather)
        if (!c.set) {
            if (blocking() && !cv.wait_for(lock, std::chrono::seconds(60), [&] { return g.gen_gather != mine; })) return 5;
            return 0;
        }
    const size_t bytes = count * size_of(dtype);
    std::vector<char> all(bytes * g.nranks);
    for (int r = 0; r < g.nranks; ++r) {
        if (g.gather[r].count != count || g.gather[r].dtype != dtype) return 4;
        memcpy(all.data() + r * bytes, g.gather[r].send, bytes);
    }
    for (int r = 0; r < g.nranks; ++r) memcpy(g.gather[r].recv, all.data(), all.size());
    for (call &c : g.gather) c.set = false;
    ++g.gen_gather;
    cv.notify_all();
    return 0;
}
static ncclResult_t wire(group &g, int from, int to)
{
    p2p &w = g.wires[{from, to}];
    if (!w.send || !w.recv) return 0;
    memcpy(w.recv, w.send, w.count * size_of(w.dtype));
    g.wires.erase({from, to});
    return 0;
}
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void *send, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t)
{
    LOCKED;
    p2p &w = comm->g->wires[{comm->rank, peer}];
    if (w.recv && (w.count != count || w.dtype != dtype)) return 4;
    w.send = send; w.count = count; w.dtype = dtype;
    return wire(*comm->g, comm->rank, peer);
}
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void *recv, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t)
{
    LOCKED;
    p2p &w = comm->g->wires[{peer, comm->rank}];
    if (w.send && (w.count != count || w.dtype != dtype)) return 4;
    w.recv = recv; w.count = count; w.dtype = dtype;
    return wire(*comm->g, peer, comm->rank);
}
}  // extern "C"
