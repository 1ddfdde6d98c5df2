"""ServerCommunication over the C ABI (cubecl_amd/csrc/comm.cpp) with two ranks, without GPUs: the product sources on the
fake HIP runtime (tests/fake_hip/) and a single-process stand-in for librccl.so.1 (tests/fake_hip/rccl/) found through
LD_LIBRARY_PATH by comm.cpp's own dlopen -- so the child process below runs the real loading code, the dtype / op mapping
(crates/cubecl-cuda/src/compute/communication.rs:27-108), the two event fences and the argument checks.  Ranks are two
contexts of one process, as in the reference's own test (crates/cubecl-core/src/runtime_tests/all_reduce.rs:5-62).
The real multi-GPU run is the driver's (bench.py --gpus N); test infrastructure only."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
FAKE = ROOT / "tests" / "fake_hip"

CHILD = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from cubecl_amd import _native as N
lib = C.CDLL(sys.argv[2])
for name, (restype, argtypes) in N.PROTOTYPES.items():
    if hasattr(lib, name):
        getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
lib.faketest_set_device.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
lib.faketest_set_device(b"gfx950:sramecc+:xnack-", 64, 2)
ctxs = []
for i in range(2):
    c = C.c_void_p(); assert lib.mi355_ctx_create(i, C.byref(c)) == N.OK; ctxs.append(c)
p = N.DeviceProps(); assert lib.mi355_device_props(ctxs[0], C.byref(p)) == N.OK and p.server_comm_enabled == 1

def dev(ctx, arr):
    d = C.c_void_p(); assert lib.mi355_alloc(ctx, max(arr.nbytes, 16), C.byref(d)) == N.OK
    assert lib.mi355_write(ctx, None, d, arr.ctypes.data, arr.nbytes) == N.OK
    return d
def host(ctx, d, like):
    out = np.zeros_like(like); assert lib.mi355_read(ctx, None, out.ctypes.data, d, out.nbytes) == N.OK
    return out

uid = (C.c_uint8 * N.UNIQUE_ID_BYTES)()
assert lib.mi355_comm_unique_id(uid) == N.OK and bytes(uid).startswith(b"fake-rccl-")
comms = []
for r in range(2):
    cm = C.c_void_p(); assert lib.mi355_comm_init(ctxs[r], uid, r, 2, C.byref(cm)) == N.OK; comms.append(cm)
bad = C.c_void_p()
assert lib.mi355_comm_init(ctxs[0], uid, 2, 2, C.byref(bad)) == N.E_INVALID_ARGUMENT        # rank outside the world

# all_reduce, sum (runtime_tests/all_reduce.rs: every device contributes, every device ends with the sum), in place
x = [np.array([1, 2, 3, 4], dtype=np.float32), np.array([10, 20, 30, 40], dtype=np.float32)]
d = [dev(ctxs[r], x[r]) for r in range(2)]
for r in range(2):
    assert lib.mi355_all_reduce(ctxs[r], comms[r], None, d[r], d[r], 4, N.DTYPE_F32, N.REDUCE_SUM) == N.OK
    assert lib.mi355_sync_collective(ctxs[r], None) == N.OK
for r in range(2):
    assert host(ctxs[r], d[r], x[0]).tolist() == [11, 22, 33, 44]
# mean / max / min, out of place, 64-bit integers
for op, want in ((N.REDUCE_MEAN, [5.5, 11, 16.5, 22]), (N.REDUCE_MAX, [10, 20, 30, 40]), (N.REDUCE_MIN, [1, 2, 3, 4])):
    src = [dev(ctxs[r], x[r].astype(np.float64)) for r in range(2)]
    dst = [dev(ctxs[r], np.zeros(4)) for r in range(2)]
    for r in range(2):
        assert lib.mi355_all_reduce(ctxs[r], comms[r], None, src[r], dst[r], 4, N.DTYPE_F64, op) == N.OK
    assert all(host(ctxs[r], dst[r], np.zeros(4)).tolist() == want for r in range(2))
# all_gather of the (value, index) records the sharded argmax exchanges: 2 x u64 per rank, rank order
rec = [np.array([7 + r, 1000 * (r + 1)], dtype=np.uint64) for r in range(2)]
src = [dev(ctxs[r], rec[r]) for r in range(2)]
dst = [dev(ctxs[r], np.zeros(4, dtype=np.uint64)) for r in range(2)]
for r in range(2):
    assert lib.mi355_all_gather(ctxs[r], comms[r], None, src[r], dst[r], 2, N.DTYPE_U64) == N.OK
assert all(host(ctxs[r], dst[r], np.zeros(4, dtype=np.uint64)).tolist() == [7, 1000, 8, 2000] for r in range(2))
# which stream a collective runs on (comm.cpp collective_stream): a message of at most 4 KiB is queued in the compute stream's
# order -- no compute -> comm fence before it, nothing for sync_collective to do behind it -- a larger one goes through the
# communication stream between the two event fences (crates/cubecl-cuda/src/compute/server.rs:749, :764-797), and so does a
# small one issued while a larger one is still un-fenced (one communicator never has work in flight on two streams)
log = (C.c_uint64 * 4)()
def waits():
    lib.faketest_stream_log(log); return int(log[1])
w0 = waits()
for r in range(2):
    assert lib.mi355_all_gather(ctxs[r], comms[r], None, src[r], dst[r], 2, N.DTYPE_U64) == N.OK
    assert lib.mi355_sync_collective(ctxs[r], None) == N.OK
assert waits() == w0, "a 32-byte all-gather must not fence"
# ... but sync_collective keeps the reference's guarantee for ANY stream (crates/cubecl-cuda/src/compute/server.rs:782-797): a second
# compute stream that asks is ordered behind the inline collective by an event; the stream that carried it needs nothing; a later
# inline collective issued from the other stream waits for the first (a communicator never has work in flight on two streams)
s2 = C.c_void_p(); assert lib.mi355_stream_create(ctxs[0], C.byref(s2)) == N.OK
for r in range(2):
    assert lib.mi355_all_gather(ctxs[r], comms[r], None, src[r], dst[r], 2, N.DTYPE_U64) == N.OK
assert lib.mi355_sync_collective(ctxs[0], s2) == N.OK
lib.faketest_stream_log(log)
assert int(log[1]) == w0 + 1 and int(log[2]) == s2.value, "another stream is ordered behind an inline collective"
assert lib.mi355_sync_collective(ctxs[0], None) == N.OK and waits() == w0 + 1, "the carrying stream needs no fence"
assert lib.mi355_all_gather(ctxs[0], comms[0], s2, src[0], dst[0], 2, N.DTYPE_U64) == N.OK       # rank 0 from its second stream
assert lib.mi355_all_gather(ctxs[1], comms[1], None, src[1], dst[1], 2, N.DTYPE_U64) == N.OK
lib.faketest_stream_log(log)
assert int(log[1]) == w0 + 2 and int(log[2]) == s2.value, "an inline collective on another stream waits for the previous one"
assert lib.mi355_sync_collective(ctxs[0], s2) == N.OK and waits() == w0 + 2
assert lib.mi355_stream_destroy(ctxs[0], s2) == N.OK                                            # (synchronizes: nothing left to fence)
assert lib.mi355_sync_collective(ctxs[0], None) == N.OK and waits() == w0 + 2
w0 = waits()
big = [dev(ctxs[r], np.full(2048, r + 1.0, dtype=np.float32)) for r in range(2)]                 # 8 KiB
for r in range(2):
    assert lib.mi355_all_reduce(ctxs[r], comms[r], None, big[r], big[r], 2048, N.DTYPE_F32, N.REDUCE_SUM) == N.OK
assert waits() == w0 + 2, "compute -> comm fence of the 8 KiB all-reduce, once per rank"
for r in range(2):                                                                               # small, behind the un-fenced large one
    assert lib.mi355_all_gather(ctxs[r], comms[r], None, src[r], dst[r], 2, N.DTYPE_U64) == N.OK
assert waits() == w0 + 4, "a small collective behind an un-fenced large one follows it onto the communication stream"
for r in range(2):
    assert lib.mi355_sync_collective(ctxs[r], None) == N.OK
assert waits() == w0 + 6 and host(ctxs[0], big[0], np.zeros(2048, dtype=np.float32)).tolist() == [3.0] * 2048
# send / recv in either call order (runtime_tests/to_client.rs moves 0..6 as f32)
payload = np.arange(6, dtype=np.float32)
s0, r1 = dev(ctxs[0], payload), dev(ctxs[1], np.zeros(6, dtype=np.float32))
assert lib.mi355_send(ctxs[0], comms[0], None, s0, 6, N.DTYPE_F32, 1) == N.OK
assert lib.mi355_recv(ctxs[1], comms[1], None, r1, 6, N.DTYPE_F32, 0) == N.OK
assert host(ctxs[1], r1, payload).tolist() == payload.tolist()
back = dev(ctxs[0], np.zeros(6, dtype=np.float32))
assert lib.mi355_recv(ctxs[0], comms[0], None, back, 6, N.DTYPE_F32, 1) == N.OK
assert lib.mi355_send(ctxs[1], comms[1], None, r1, 6, N.DTYPE_F32, 0) == N.OK
assert host(ctxs[0], back, payload).tolist() == payload.tolist()
# argument checks of the wrappers themselves
assert lib.mi355_all_reduce(ctxs[0], comms[0], None, d[0], d[0], 4, N.DTYPE_F8E4M3, N.REDUCE_SUM) == N.E_UNSUPPORTED
assert lib.mi355_all_reduce(ctxs[0], comms[0], None, d[0], d[0], 4, N.DTYPE_F32, 77) == N.E_UNSUPPORTED
assert lib.mi355_all_reduce(ctxs[0], comms[0], None, d[0], d[0], 0, N.DTYPE_F32, N.REDUCE_SUM) == N.OK       # empty: no-op
assert lib.mi355_all_reduce(ctxs[0], comms[0], None, None, d[0], 4, N.DTYPE_F32, N.REDUCE_SUM) == N.E_INVALID_ARGUMENT
assert lib.mi355_send(ctxs[0], comms[0], None, s0, 6, N.DTYPE_F32, 0) == N.E_INVALID_ARGUMENT               # to itself
assert lib.mi355_recv(ctxs[0], comms[0], None, back, 6, N.DTYPE_F32, 5) == N.E_INVALID_ARGUMENT
# a collective the library refuses comes back as MI355_E_COMM with RCCL's message (same rank twice before its peer)
assert lib.mi355_all_reduce(ctxs[0], comms[0], None, d[0], d[0], 4, N.DTYPE_F32, N.REDUCE_SUM) == N.OK
assert lib.mi355_all_reduce(ctxs[0], comms[0], None, d[0], d[0], 4, N.DTYPE_F32, N.REDUCE_SUM) == N.E_COMM
assert b"invalid usage" in lib.mi355_last_error(ctxs[0])
for r in range(2):
    assert lib.mi355_comm_destroy(ctxs[r], comms[r]) == N.OK and lib.mi355_ctx_destroy(ctxs[r]) == N.OK
print("comm ok")
'''


def _run_child(code, extra_env=None):
    rccl = FAKE / "rccl" / "librccl.so.1"
    src = FAKE / "rccl" / "fake_rccl.cpp"
    if not rccl.exists() or rccl.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-shared", "-fPIC", "-fvisibility=hidden", "-pthread", "-o", str(rccl),
                        str(src)], check=True)
    from test_runtime_cpu import build_runtime_lib           # the product sources on the fake HIP runtime
    runtime = build_runtime_lib()
    env = dict(os.environ, LD_LIBRARY_PATH=str(FAKE / "rccl") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), **(extra_env or {}))
    return subprocess.run([sys.executable, "-c", code, str(ROOT), str(runtime)], env=env, capture_output=True, text=True, timeout=120)


def test_collectives_between_two_ranks_on_the_fake_device():
    out = _run_child(CHILD)
    assert out.returncode == 0 and "comm ok" in out.stdout, out.stderr[-2000:]


# One process, one context per device, one THREAD per device -- the model the reference's comm_init assumes (it runs on each
# device's own runner thread and blocks until every rank has joined: crates/cubecl-cuda/src/compute/server.rs:669-703) -- with
# the reference's test body and closed form (crates/cubecl-core/src/runtime_tests/all_reduce.rs:5-62): device i contributes
# i + j to handle j, every device must end with sum(i) + j * device_count.  The stand-in RCCL blocks in ncclCommInitRank like
# the real one here; tests/test_gpu_runtime.py runs the same body over real RCCL when the box has two GPUs or more.
THREADED = r'''
import ctypes as C, sys, threading
import numpy as np
sys.path.insert(0, sys.argv[1])
from cubecl_amd import _native as N
lib = C.CDLL(sys.argv[2])
for name, (restype, argtypes) in N.PROTOTYPES.items():
    if hasattr(lib, name):
        getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
lib.faketest_set_device.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
NDEV, SIZE, NUM_HANDLES = 4, 100, 8
lib.faketest_set_device(b"gfx950:sramecc+:xnack-", 64, NDEV)
uid = (C.c_uint8 * N.UNIQUE_ID_BYTES)()
assert lib.mi355_comm_unique_id(uid) == N.OK
step = threading.Barrier(NDEV)            # the stand-in executes a collective inside its LAST caller: line the ranks up per call
results, errors = {}, []

def device_thread(i):
    try:
        ctx, comm = C.c_void_p(), C.c_void_p()
        assert lib.mi355_ctx_create(i, C.byref(ctx)) == N.OK
        assert lib.mi355_comm_init(ctx, uid, i, NDEV, C.byref(comm)) == N.OK            # returns once all four have joined
        handles = []
        for j in range(NUM_HANDLES):
            d = C.c_void_p()
            src = np.full(SIZE, i + j, dtype=np.float32)
            assert lib.mi355_pool_alloc(ctx, None, src.nbytes, C.byref(d)) == N.OK
            assert lib.mi355_write(ctx, None, d, src.ctypes.data, src.nbytes) == N.OK
            handles.append(d)
        for d in handles:
            assert lib.mi355_all_reduce(ctx, comm, None, d, d, SIZE, N.DTYPE_F32, N.REDUCE_SUM) == N.OK
            step.wait()
        assert lib.mi355_sync_collective(ctx, None) == N.OK                             # AFTER all the all_reduce calls, as the reference
        got = []
        for d in handles:
            out = np.zeros(SIZE, dtype=np.float32)
            assert lib.mi355_read(ctx, None, out.ctypes.data, d, out.nbytes) == N.OK
            got.append(out)
            assert lib.mi355_pool_free(ctx, None, d) == N.OK
        # the argmax exchange on the same communicator: all-gather of {value, local index} records + the device-side rule's
        # host twin is checked in tests/test_sharded_gloo.py; here the gather itself, every rank, rank order
        rec, allrec = C.c_void_p(), C.c_void_p()
        mine = np.array([100 + i, 7 * i], dtype=np.uint64)
        assert lib.mi355_pool_alloc(ctx, None, 16, C.byref(rec)) == N.OK and lib.mi355_pool_alloc(ctx, None, 16 * NDEV, C.byref(allrec)) == N.OK
        assert lib.mi355_write(ctx, None, rec, mine.ctypes.data, 16) == N.OK
        assert lib.mi355_all_gather(ctx, comm, None, rec, allrec, 2, N.DTYPE_U64) == N.OK
        step.wait()
        assert lib.mi355_sync_collective(ctx, None) == N.OK
        gathered = np.zeros(2 * NDEV, dtype=np.uint64)
        assert lib.mi355_read(ctx, None, gathered.ctypes.data, allrec, gathered.nbytes) == N.OK
        results[i] = (got, gathered.tolist())
        step.wait()
        assert lib.mi355_comm_destroy(ctx, comm) == N.OK and lib.mi355_ctx_destroy(ctx) == N.OK
    except BaseException as exc:          # a failed rank must not leave the others waiting forever
        errors.append(f"device {i}: {type(exc).__name__}: {exc}")
        step.abort()

threads = [threading.Thread(target=device_thread, args=(i,)) for i in range(NDEV)]
for t in threads: t.start()
for t in threads: t.join(60)
assert not errors and not any(t.is_alive() for t in threads), errors
value_base = float(sum(range(NDEV)))
for i in range(NDEV):
    got, gathered = results[i]
    for j, out in enumerate(got):
        assert np.array_equal(out, np.full(SIZE, value_base + j * NDEV, dtype=np.float32)), (i, j, out[:4])
    assert gathered == [v for r in range(NDEV) for v in (100 + r, 7 * r)]
print("threaded comm ok")
'''


def test_all_reduce_from_one_thread_per_device_matches_the_references_closed_form():
    out = _run_child(THREADED, {"FAKE_RCCL_BLOCKING_INIT": "1"})
    assert out.returncode == 0 and "threaded comm ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
