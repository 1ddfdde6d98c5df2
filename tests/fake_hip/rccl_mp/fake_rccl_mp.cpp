// A MULTI-PROCESS stand-in for librccl.so.1 (TEST INFRASTRUCTURE ONLY): one rank per process, as bench.py --gpus N and the
// driver's launcher run them.  The ranks of a communicator meet in a file-backed shared mapping under /tmp named by the unique id
// (process-shared mutex + condition variable); a collective blocks until every rank has called it and works on the HOST memory
// the fake HIP runtime hands out (payloads up to 256 bytes per rank: the job-level barrier / max-over-ranks and the C4
// exchange move 4 to 128).  tests/test_bench_cpu.py puts this directory on LD_LIBRARY_PATH so that comm.cpp's own
// dlopen("librccl.so.1") finds it -- the product's loading code, dtype / op mapping and fences run unchanged.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

typedef int ncclResult_t;                      // 0 = success, 4 = invalid argument, 5 = invalid usage
struct hipStream_opaque;
typedef hipStream_opaque *hipStream_t;
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
constexpr int MAX_RANKS = 64, SLOT = 256;
struct shared {
    volatile int ready;                        // the creator has initialised mutex and condition variable
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int nranks, joined, arrived, departed, phase;   // phase 0: gathering calls, 1: handing the result out
    unsigned long long gen;
    int kind, dtype, op; size_t count;         // of the collective being gathered (every rank must agree)
    int bad;
    char slot[MAX_RANKS][SLOT];
    char result[MAX_RANKS * SLOT];
};
size_t size_of(int dt) { const size_t s[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2}; return dt >= 0 && dt < 10 ? s[dt] : 0; }
bool wait_until(shared *sh, bool (*pred)(shared *, unsigned long long), unsigned long long arg, int seconds)
{
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += seconds;
    while (!pred(sh, arg))
        if (pthread_cond_timedwait(&sh->cv, &sh->mu, &ts) == ETIMEDOUT) return pred(sh, arg);
    return true;
}
template <typename T> void reduce_typed(shared *sh, size_t n, int op)
{
    T *out = reinterpret_cast<T *>(sh->result);
    for (size_t i = 0; i < n; ++i) {
        T acc = reinterpret_cast<const T *>(sh->slot[0])[i];
        for (int r = 1; r < sh->nranks; ++r) {
            const T v = reinterpret_cast<const T *>(sh->slot[r])[i];
            acc = op == 2 ? (v > acc ? v : acc) : op == 3 ? (v < acc ? v : acc) : op == 1 ? (T)(acc * v) : (T)(acc + v);
        }
        out[i] = op == 4 ? (T)(acc / (T)sh->nranks) : acc;
    }
}
int counter = 0;
}  // namespace

struct ncclComm { shared *sh; int rank; };
typedef ncclComm *ncclComm_t;

// kind 0 = all-reduce, 1 = all-gather
static ncclResult_t collective(ncclComm_t comm, int kind, const void *send, void *recv, size_t count, int dtype, int op)
{
    shared *sh = comm->sh;
    const size_t bytes = count * size_of(dtype);
    if (bytes == 0 || bytes > SLOT) return 4;
    pthread_mutex_lock(&sh->mu);
    if (!wait_until(sh, [](shared *s, unsigned long long) { return s->phase == 0; }, 0, 60)) { pthread_mutex_unlock(&sh->mu); return 5; }
    if (sh->arrived == 0) { sh->kind = kind; sh->dtype = dtype; sh->op = op; sh->count = count; sh->bad = 0; }
    else if (sh->kind != kind || sh->dtype != dtype || sh->op != op || sh->count != count) sh->bad = 1;
    memcpy(sh->slot[comm->rank], send, bytes);
    const unsigned long long mine = sh->gen;
    if (++sh->arrived == sh->nranks) {
        if (kind == 0) {
            switch (dtype) {
            case 2: reduce_typed<int32_t>(sh, count, op); break;
            case 3: reduce_typed<uint32_t>(sh, count, op); break;
            case 4: reduce_typed<int64_t>(sh, count, op); break;
            case 5: reduce_typed<uint64_t>(sh, count, op); break;
            case 7: reduce_typed<float>(sh, count, op); break;
            case 8: reduce_typed<double>(sh, count, op); break;
            default: sh->bad = 1;
            }
        } else {
            for (int r = 0; r < sh->nranks; ++r) memcpy(sh->result + r * bytes, sh->slot[r], bytes);
        }
        sh->phase = 1; sh->departed = 0; ++sh->gen;
        pthread_cond_broadcast(&sh->cv);
    } else if (!wait_until(sh, [](shared *s, unsigned long long g) { return s->gen != g; }, mine, 60)) {
        pthread_mutex_unlock(&sh->mu);
        return 5;                                                // a rank never showed up
    }
    const int bad = sh->bad;
    if (!bad) memcpy(recv, sh->result, kind == 0 ? bytes : bytes * sh->nranks);
    if (++sh->departed == sh->nranks) { sh->phase = 0; sh->arrived = 0; pthread_cond_broadcast(&sh->cv); }
    pthread_mutex_unlock(&sh->mu);
    return bad ? 4 : 0;
}

extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "fake-rccl-mp-%d-%d-%ld", (int)getpid(), counter++, (long)time(nullptr));
    return 0;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 4;
    char path[192];
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(path, sizeof path, "/tmp/%s.shm", id.internal);
    int fd = open(path, O_RDWR | O_CREAT | O_EXCL, 0600);
    const bool creator = fd >= 0;
    if (!creator) fd = open(path, O_RDWR);
    if (fd < 0) return 5;
    if (creator && ftruncate(fd, sizeof(shared)) != 0) { close(fd); return 5; }
    for (int spin = 0; !creator && spin < 6000; ++spin) {        // wait for the creator's ftruncate
        off_t end = lseek(fd, 0, SEEK_END);
        if (end >= (off_t)sizeof(shared)) break;
        usleep(10000);
    }
    void *m = mmap(nullptr, sizeof(shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 5;
    shared *sh = static_cast<shared *>(m);
    if (creator) {
        pthread_mutexattr_t ma; pthread_mutexattr_init(&ma); pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
        pthread_condattr_t ca; pthread_condattr_init(&ca); pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
        pthread_mutex_init(&sh->mu, &ma);
        pthread_cond_init(&sh->cv, &ca);
        sh->nranks = nranks;
        __sync_synchronize();
        sh->ready = 1;
    } else {
        for (int spin = 0; !sh->ready && spin < 6000; ++spin) usleep(10000);
        if (!sh->ready) return 5;
    }
    pthread_mutex_lock(&sh->mu);
    ncclResult_t rc = 0;
    if (sh->nranks != nranks) rc = 4;
    else {
        ++sh->joined;
        pthread_cond_broadcast(&sh->cv);
        // like the real ncclCommInitRank: returns when every rank of the communicator has called it
        if (!wait_until(sh, [](shared *s, unsigned long long) { return s->joined >= s->nranks; }, 0, 60)) rc = 5;
    }
    pthread_mutex_unlock(&sh->mu);
    if (rc == 0) *comm = new ncclComm{sh, rank};
    if (creator) unlink(path);                                   // everybody has it mapped (or the job is lost anyway)
    return rc;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) { if (comm) { munmap(comm->sh, sizeof(shared)); delete comm; } return 0; }
__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : r == 4 ? "invalid argument" : "invalid usage"; }
__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t comm, hipStream_t)
{ return collective(comm, 0, send, recv, count, dtype, op); }
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t comm, hipStream_t)
{ return collective(comm, 1, send, recv, count, dtype, 0); }
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void *, size_t, int, int, ncclComm_t, hipStream_t) { return 5; }
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void *, size_t, int, int, ncclComm_t, hipStream_t) { return 5; }
}  // extern "C"
