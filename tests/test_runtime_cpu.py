"""The host runtime behind the C ABI (cubecl_amd/csrc/runtime.cpp + pool.cpp + comm.cpp) on a fake device: the product
sources compiled with g++ against tests/fake_hip/ (device memory = anonymous mappings, every stream operation executes at
once, failures injected on request).  What the reference checks on its DummyServer (crates/cubecl-runtime/tests/
integration_test.rs) and in runtime_tests/launch.rs -- context creation, storage, IO, the fire-and-forget error queue,
resource limits, profiling tokens, graph capture -- checked here without a GPU; the same cases run on the MI355X in
tests/test_gpu_runtime.py.  Test infrastructure only."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from cubecl_amd import _native as N

ROOT = Path(__file__).resolve().parents[1]
FAKE = ROOT / "tests" / "fake_hip"
CSRC = ROOT / "cubecl_amd" / "csrc"
HIP_ERROR_LAUNCH_FAILURE, HIP_ERROR_INVALID_VALUE = 719, 1


def build_runtime_lib() -> Path:
    so = FAKE / "libruntimetest.so"
    srcs = [CSRC / "runtime.cpp", CSRC / "pool.cpp", CSRC / "comm.cpp", FAKE / "fake_hip.cpp"]
    deps = srcs + [CSRC / "internal.hpp", FAKE / "hip" / "hip_runtime.h", ROOT / "include" / "mi355cube.h", Path(__file__)]
    if not so.exists() or so.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-Wno-format-truncation", "-DFAKE_WITH_RUNTIME", "-shared",
                        "-fPIC", "-Wl,-Bsymbolic", "-I", str(FAKE), "-o", str(so)] + [str(s) for s in srcs] + ["-ldl"], check=True)
    return so


@pytest.fixture(scope="module")
def lib():
    lib = C.CDLL(str(build_runtime_lib()))
    for name, (restype, argtypes) in N.PROTOTYPES.items():          # the product's own prototype table
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
    lib.faketest_set_device.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
    lib.faketest_fail_next.argtypes = [C.c_int32, C.c_int32]
    lib.faketest_expect_params.argtypes = [C.c_uint32]
    lib.faketest_launch_log.argtypes = [C.POINTER(C.c_uint64)]
    lib.pooltest_set_capacity.argtypes = [C.c_uint64]
    lib.pooltest_counters.argtypes = [C.POINTER(C.c_uint64)]
    lib.faketest_stream_log.argtypes = [C.POINTER(C.c_uint64)]
    lib.faketest_fail_end_capture.argtypes = [C.c_int32]
    lib.faketest_scratch_get.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.faketest_scratch_get.restype = C.c_int32
    lib.faketest_set_comm_dirty.argtypes = [C.c_void_p, C.c_int32]
    lib.pooltest_inside_allocation.argtypes = [C.c_void_p, C.c_uint64]
    lib.pooltest_inside_allocation.restype = C.c_int32
    return lib


@pytest.fixture()
def ctx(lib):
    lib.faketest_set_device(b"gfx950:sramecc+:xnack-", 64, 1)
    lib.pooltest_set_capacity(288 << 30)
    c = C.c_void_p()
    assert lib.mi355_ctx_create(0, C.byref(c)) == N.OK
    yield c
    assert lib.mi355_ctx_destroy(c) == N.OK


def _device(lib):
    out = (C.c_uint64 * 8)()
    lib.pooltest_counters(out)
    return dict(zip(("mallocs", "frees", "bad_frees", "live", "in_use", "events", "event_queries", "device_syncs"), out))


def _launches(lib):
    out = (C.c_uint64 * 20)()
    lib.faketest_launch_log(out)
    return list(out)


def _streams(lib):
    out = (C.c_uint64 * 4)()
    lib.faketest_stream_log(out)
    return dict(zip(("syncs", "waits", "last_wait_stream", "last_sync_stream"), out))


def _pop(lib, ctx):
    code, req, mx, buf = C.c_int32(), C.c_uint64(), C.c_uint64(), C.create_string_buffer(512)
    rc = lib.mi355_error_pop(ctx, C.byref(code), C.byref(req), C.byref(mx), buf, 512)
    return rc, code.value, req.value, mx.value, buf.value.decode()


def test_context_refuses_everything_but_wave64_gfx950_and_describes_the_device(lib, ctx):
    p = N.DeviceProps()
    assert lib.mi355_device_props(ctx, C.byref(p)) == N.OK
    assert (p.plane_size_min, p.plane_size_max, p.load_width_bits, p.num_streaming_multiprocessors, p.num_xcd) == (64, 64, 128, 256, 8)
    assert p.max_shared_memory_size == 160 << 10 and p.max_units_per_cube == 1024 and tuple(p.max_cube_count) == (2147483647, 65535, 65535)
    assert p.gcn_arch_name.startswith(b"gfx950") and p.fingerprint.startswith(b"mi355-aot_gfx950") and p.mem_alignment == 256
    assert p.total_memory == 288 << 30 and p.max_page_size == (288 << 30) // 4 and p.plane_ops == 1 and p.abi_version == N.ABI_VERSION
    cfgs = {(c.a_type, c.cd_type, c.m, c.n, c.k) for c in p.mma_configs[: p.num_mma_configs]}
    assert {(N.DTYPE_BF16, N.DTYPE_F32, 32, 32, 16), (N.DTYPE_F32, N.DTYPE_F32, 32, 32, 2), (N.DTYPE_F16, N.DTYPE_F32, 16, 16, 16),
            (N.DTYPE_F8E4M3, N.DTYPE_F32, 32, 32, 64)} <= cfgs and p.num_scaled_mma_configs == 5
    other = C.c_void_p()
    for arch, warp, count, index in ((b"gfx942:sramecc+", 64, 1, 0), (b"gfx1100", 32, 1, 0), (b"gfx950", 64, 0, 0), (b"gfx950", 64, 1, 3)):
        lib.faketest_set_device(arch, warp, count)
        assert lib.mi355_ctx_create(index, C.byref(other)) == N.E_NO_DEVICE and not other.value
        assert lib.mi355_last_global_error()                               # says why: a loud error, not a fallback
    n = C.c_int32(7)
    assert lib.mi355_device_count(C.byref(n)) == N.OK and n.value == 1
    assert lib.mi355_ctx_create(0, None) == N.E_INVALID_ARGUMENT and lib.mi355_sync(None, None) == N.E_INVALID_ARGUMENT


def test_storage_limits_deferred_frees_and_io_round_trips(lib, ctx):
    p = C.c_void_p()
    assert lib.mi355_alloc(ctx, (72 << 30) + 1, C.byref(p)) == N.E_BUFFER_TOO_BIG and b"max_page_size" in lib.mi355_last_error(ctx)
    assert lib.mi355_alloc(ctx, 0, C.byref(p)) == N.OK and not p.value                      # empty allocation
    a, b = C.c_void_p(), C.c_void_p()
    assert lib.mi355_alloc(ctx, 1 << 20, C.byref(a)) == N.OK and lib.mi355_alloc(ctx, 1 << 20, C.byref(b)) == N.OK
    live = _device(lib)["live"]
    assert lib.mi355_free(ctx, a) == N.OK and _device(lib)["live"] == live                  # deferred: nothing reaches the driver ...
    assert lib.mi355_flush(ctx) == N.OK and _device(lib)["live"] == live - 1                # ... before flush, behind a device sync
    lib.pooltest_set_capacity(_device(lib)["in_use"] + (1 << 20))
    assert lib.mi355_free(ctx, b) == N.OK
    assert lib.mi355_alloc(ctx, 2 << 20, C.byref(a)) == N.OK                                # OOM -> pending frees released -> retry fits
    assert lib.mi355_alloc(ctx, 2 << 20, C.byref(b)) == N.E_OUT_OF_MEMORY                   # a driver OOM is OutOfMemory, not BufferTooBig
    lib.pooltest_set_capacity(288 << 30)
    x = np.arange(1000, dtype=np.float32)
    back = np.zeros_like(x)
    assert lib.mi355_write(ctx, None, a, x.ctypes.data, x.nbytes) == N.OK
    assert lib.mi355_read(ctx, None, back.ctypes.data, a, x.nbytes) == N.OK and np.array_equal(back, x)
    assert lib.mi355_write(ctx, None, None, None, 0) == N.OK and lib.mi355_write(ctx, None, None, x.ctypes.data, 4) == N.E_INVALID_ARGUMENT
    # pitched rows (PitchedMemoryLayoutPolicy): 5 rows of 100 bytes on a 128-byte pitch, padding left alone
    pitch = C.c_uint64()
    for width, want in ((100, 128), (5, 16), (16, 16), (300, 512), (4096, 4096), (4100, 4352), (0, 0)):
        assert lib.mi355_pitched_row_bytes(ctx, width, C.byref(pitch)) == N.OK and pitch.value == want, width
    assert lib.mi355_memset(ctx, None, a, 0xEE, 5 * 128) == N.OK
    rows = np.arange(500, dtype=np.uint8).reshape(5, 100)
    assert lib.mi355_write_2d(ctx, None, a, 128, rows.ctypes.data, 100, 100, 5) == N.OK
    raw = np.zeros(5 * 128, dtype=np.uint8)
    assert lib.mi355_read(ctx, None, raw.ctypes.data, a, raw.nbytes) == N.OK
    assert np.array_equal(raw.reshape(5, 128)[:, :100], rows) and np.all(raw.reshape(5, 128)[:, 100:] == 0xEE)
    out = np.zeros((5, 100), dtype=np.uint8)
    assert lib.mi355_read_2d(ctx, None, out.ctypes.data, 100, a, 128, 100, 5) == N.OK and np.array_equal(out, rows)
    assert lib.mi355_write_2d(ctx, None, a, 64, rows.ctypes.data, 100, 100, 5) == N.E_UNSUPPORTED_STRIDES
    c = C.c_void_p()
    assert lib.mi355_alloc(ctx, 4096, C.byref(c)) == N.OK and lib.mi355_copy_d2d(ctx, None, c, a, 640) == N.OK
    assert lib.mi355_read(ctx, None, raw.ctypes.data, c, 640) == N.OK and np.array_equal(raw.reshape(5, 128)[:, :100], rows)
    for ptr in (a, c):
        lib.mi355_free(ctx, ptr)


def test_launch_is_fire_and_forget_and_failures_surface_at_flush(lib, ctx):
    mod, fn = C.c_void_p(), C.c_void_p()
    assert lib.mi355_module_load(ctx, b"not a code object", 17, C.byref(mod)) == N.E_COMPILATION
    assert lib.mi355_module_load(ctx, b"FAKEHSACO....", 13, C.byref(mod)) == N.OK
    assert lib.mi355_module_get_function(ctx, mod, b"missing_kernel", C.byref(fn)) == N.E_NOT_FOUND
    assert lib.mi355_module_get_function(ctx, mod, b"abi_axpb", C.byref(fn)) == N.OK
    grid, block = (C.c_uint32 * 3), (C.c_uint32 * 3)
    ptrs = (C.c_void_p * 3)(0x1000, 0x2000, 0x3000)
    n0 = _launches(lib)[0]
    for zero in ((0, 1, 1), (1, 0, 1), (1, 1, 0)):                                          # client.rs:880-884: a no-op, not an error
        assert lib.mi355_launch(ctx, None, fn, grid(*zero), block(64, 1, 1), 0, ptrs, 3) == N.OK
    assert _launches(lib)[0] == n0 and lib.mi355_flush(ctx) == N.OK
    lib.faketest_expect_params(3)
    assert lib.mi355_launch(ctx, None, fn, grid(4, 2, 1), block(256, 1, 1), 96 << 10, ptrs, 3) == N.OK
    log = _launches(lib)
    assert log[0] == n0 + 1 and log[3:10] == [4, 2, 1, 256, 1, 1, 96 << 10] and log[10] == 96 << 10     # > 64 KiB of LDS: opted in
    assert log[11:14] == [0x1000, 0x2000, 0x3000]                                           # kernelParams[i] -> the i-th pointer VALUE
    # resource limits (runtime_tests/launch.rs:226-348): accepted, queued, reported by the next flush with their numbers
    assert lib.mi355_launch(ctx, None, fn, grid(1, 1, 1), block(64, 1, 1), (160 << 10) + 1, ptrs, 3) == N.OK
    assert lib.mi355_launch(ctx, None, fn, grid(1, 1, 1), block(2048, 1, 1), 0, ptrs, 3) == N.OK
    assert lib.mi355_launch(ctx, None, fn, grid(1, 1, 1), block(64, 32, 1), 0, ptrs, 3) == N.OK
    lib.faketest_fail_next(0, HIP_ERROR_LAUNCH_FAILURE)
    assert lib.mi355_launch(ctx, None, fn, grid(1, 1, 1), block(64, 1, 1), 0, ptrs, 3) == N.OK
    count = C.c_int32()
    assert _launches(lib)[0] == n0 + 1 and lib.mi355_error_count(ctx, C.byref(count)) == N.OK and count.value == 4
    assert lib.mi355_flush(ctx) == N.E_SERVER_UNHEALTHY and b"4 queued error" in lib.mi355_last_error(ctx)
    assert _pop(lib, ctx)[:4] == (N.OK, N.E_SHARED_MEMORY, (160 << 10) + 1, 160 << 10)
    rc, code, req, mx, msg = _pop(lib, ctx)
    assert (code, req, mx) == (N.E_CUBE_DIM, 2048, 1024) and "(2048, 1, 1), max is (1024, 1024, 1024)" in msg
    assert _pop(lib, ctx)[1:4] == (N.E_UNITS, 2048, 1024) and _pop(lib, ctx)[1] == N.E_LAUNCH
    assert _pop(lib, ctx)[0] == N.E_NOT_FOUND and lib.mi355_flush(ctx) == N.OK              # drained: healthy again
    # an execution failure discovered at a synchronisation point is reported the same way
    lib.faketest_fail_next(HIP_ERROR_INVALID_VALUE, 0)
    assert lib.mi355_sync(ctx, None) == N.E_SERVER_UNHEALTHY and _pop(lib, ctx)[1] == N.E_EXECUTION and lib.mi355_sync(ctx, None) == N.OK
    assert lib.mi355_launch(ctx, None, None, grid(1, 1, 1), block(1, 1, 1), 0, ptrs, 3) == N.E_INVALID_ARGUMENT
    assert lib.mi355_module_unload(ctx, mod) == N.OK


def test_profile_tokens_events_and_graph_capture(lib, ctx):
    token, nanos = C.c_uint64(), C.c_uint64()
    assert lib.mi355_profile_start(ctx, None, C.byref(token)) == N.OK and token.value == 1
    second = C.c_uint64()
    assert lib.mi355_profile_start(ctx, None, C.byref(second)) == N.OK and second.value == 2              # nested regions
    assert lib.mi355_profile_stop(ctx, None, token, C.byref(nanos)) == N.OK and nanos.value == 1_500_000
    assert lib.mi355_profile_stop(ctx, None, token, C.byref(nanos)) == N.E_PROFILE                         # a token ends once
    assert lib.mi355_profile_stop(ctx, None, second, C.byref(nanos)) == N.OK and lib.mi355_profile_stop(ctx, None, 99, C.byref(nanos)) == N.E_PROFILE
    assert lib.mi355_profile_start(ctx, None, C.byref(token)) == N.OK and token.value == 1                 # slots are reused
    assert lib.mi355_profile_stop(ctx, None, token, None) == N.OK
    ea, eb, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    assert lib.mi355_event_create(ctx, C.byref(ea)) == N.OK and lib.mi355_event_create(ctx, C.byref(eb)) == N.OK
    assert lib.mi355_event_record(ctx, ea, None) == N.OK and lib.mi355_event_record(ctx, eb, None) == N.OK and lib.mi355_event_sync(ctx, eb) == N.OK
    assert lib.mi355_event_elapsed_ms(ctx, ea, eb, C.byref(ms)) == N.OK and ms.value == 1.5 and lib.mi355_event_record(ctx, None, None) == N.E_INVALID_ARGUMENT
    assert lib.mi355_event_destroy(ctx, ea) == N.OK and lib.mi355_event_destroy(ctx, eb) == N.OK
    # graph capture (server/base.rs:472-532): three launches captured, none executed, every replay runs the three
    mod, fn, graph = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.mi355_module_load(ctx, b"FAKEHSACO", 9, C.byref(mod)) == N.OK and lib.mi355_module_get_function(ctx, mod, b"k", C.byref(fn)) == N.OK
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(graph)) == N.E_INVALID_ARGUMENT                   # no window open
    assert lib.mi355_graph_begin_capture(ctx, None) == N.OK and lib.mi355_graph_begin_capture(ctx, None) == N.E_INVALID_ARGUMENT
    one, before = (C.c_uint32 * 3)(1, 1, 1), _launches(lib)[0]
    p = C.c_void_p()
    assert lib.mi355_pool_alloc(ctx, None, 1 << 20, C.byref(p)) == N.E_UNSUPPORTED                          # no driver allocation inside the window
    for _ in range(3):
        assert lib.mi355_launch(ctx, None, fn, one, one, 0, None, 0) == N.OK
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(graph)) == N.OK and graph.value and _launches(lib)[0] == before
    for _ in range(2):
        assert lib.mi355_graph_replay(ctx, None, graph) == N.OK
    log = _launches(lib)
    assert log[0] == before + 6 and log[2] == 3 and lib.mi355_graph_replay(ctx, None, None) == N.E_NOT_FOUND
    assert lib.mi355_graph_destroy(ctx, graph) == N.OK and lib.mi355_pool_alloc(ctx, None, 1 << 20, C.byref(p)) == N.OK
    assert lib.mi355_pool_free(ctx, None, p) == N.OK and lib.mi355_module_unload(ctx, mod) == N.OK


def test_two_contexts_move_data_and_collectives_need_a_communicator(lib, ctx):
    other = C.c_void_p()
    assert lib.mi355_ctx_create(0, C.byref(other)) == N.OK
    a, b = C.c_void_p(), C.c_void_p()
    assert lib.mi355_alloc(ctx, 4096, C.byref(a)) == N.OK and lib.mi355_alloc(other, 4096, C.byref(b)) == N.OK
    x, back = np.arange(1024, dtype=np.int32), np.zeros(1024, dtype=np.int32)
    assert lib.mi355_write(ctx, None, a, x.ctypes.data, 4096) == N.OK
    assert lib.mi355_copy_to_ctx(ctx, None, a, other, None, b, 4096) == N.OK                                # ComputeClient::to_client
    assert lib.mi355_read(other, None, back.ctypes.data, b, 4096) == N.OK and np.array_equal(back, x)
    assert lib.mi355_copy_to_ctx(ctx, None, a, other, None, b, 0) == N.OK and lib.mi355_copy_to_ctx(ctx, None, None, other, None, b, 8) == N.E_INVALID_ARGUMENT
    assert lib.mi355_all_reduce(ctx, None, None, a, a, 1, N.DTYPE_F32, N.REDUCE_SUM) == N.E_INVALID_ARGUMENT     # before comm_init
    assert b"mi355_comm_init" in lib.mi355_last_error(ctx) and lib.mi355_sync_collective(ctx, None) == N.OK
    s = C.c_void_p()
    assert lib.mi355_stream_create(ctx, C.byref(s)) == N.OK and lib.mi355_sync(ctx, s) == N.OK and lib.mi355_stream_destroy(ctx, s) == N.OK
    for c_, p_ in ((ctx, a), (other, b)):
        lib.mi355_free(c_, p_)
    assert lib.mi355_ctx_destroy(other) == N.OK


def test_remaining_runtime_entry_points(lib, ctx):
    """Pinned staging, asynchronous reads behind an event, stream getters, memory info, the pool behind client.empty."""
    host, dev = C.c_void_p(), C.c_void_p()
    assert lib.mi355_pinned_alloc(ctx, 4096, C.byref(host)) == N.OK and host.value and lib.mi355_pinned_alloc(ctx, 0, C.byref(dev)) == N.OK
    assert lib.mi355_pool_alloc(ctx, None, 4096, C.byref(dev)) == N.OK
    src = np.arange(1024, dtype=np.uint32)
    C.memmove(host.value, src.ctypes.data, 4096)
    s, ev = C.c_void_p(), C.c_void_p()
    assert lib.mi355_stream_create(ctx, C.byref(s)) == N.OK and lib.mi355_event_create(ctx, C.byref(ev)) == N.OK
    assert lib.mi355_write(ctx, s, dev, host, 4096) == N.OK and lib.mi355_event_record(ctx, ev, s) == N.OK
    assert lib.mi355_stream_wait_event(ctx, None, ev) == N.OK                      # the compute stream continues behind the upload
    back = np.zeros(1024, dtype=np.uint32)
    assert lib.mi355_read_async(ctx, None, back.ctypes.data, dev, 4096) == N.OK and lib.mi355_event_record(ctx, ev, None) == N.OK
    assert lib.mi355_event_sync(ctx, ev) == N.OK and np.array_equal(back, src)
    assert lib.mi355_read_async(ctx, None, None, dev, 0) == N.OK and lib.mi355_read_async(ctx, None, None, dev, 4) == N.E_INVALID_ARGUMENT
    d, c_ = C.c_void_p(), C.c_void_p()
    assert lib.mi355_default_stream(ctx, C.byref(d)) == N.OK and lib.mi355_comm_stream(ctx, C.byref(c_)) == N.OK and d.value != c_.value
    free_b, total_b = C.c_uint64(), C.c_uint64()
    assert lib.mi355_mem_info(ctx, C.byref(free_b), C.byref(total_b)) == N.OK and 0 < free_b.value < total_b.value == 288 << 30
    u = N.MemoryUsage()
    assert lib.mi355_pool_usage(ctx, C.byref(u)) == N.OK and (u.number_allocs, u.bytes_in_use) == (1, 4096)
    assert lib.mi355_pool_free(ctx, s, dev) == N.OK and lib.mi355_pool_cleanup(ctx, 1) == N.OK
    assert lib.mi355_pool_usage(ctx, C.byref(u)) == N.OK and (u.number_allocs, u.bytes_reserved) == (0, 0)
    assert lib.mi355_pool_mode(ctx, N.ALLOC_MODE_PERSISTENT) == N.OK and lib.mi355_pool_mode(ctx, N.ALLOC_MODE_AUTO) == N.OK
    assert lib.mi355_event_destroy(ctx, ev) == N.OK and lib.mi355_stream_destroy(ctx, s) == N.OK and lib.mi355_pinned_free(ctx, host) == N.OK
    assert lib.mi355_abi_version() == N.ABI_VERSION


# ---- memory a graph replays against stays pinned; destroying a graph waits for its replays ----------------------------------
def _alloc(lib, ctx, nbytes, stream=None):
    p = C.c_void_p()
    assert lib.mi355_pool_alloc(ctx, stream, nbytes, C.byref(p)) == N.OK and p.value
    return p.value


def test_graph_pins_the_memory_it_replays_against(lib, ctx):
    """The reference pins what a capture window allocates for the graph's lifetime (crates/cubecl-hip/src/compute/server.rs:
    288-521).  Here: a block allocated or freed inside the window never reaches the free lists while the graph lives, so no
    later client.empty() can receive an address the next replay writes to; mi355_graph_destroy hands everything back."""
    size = 1 << 20
    warm = [_alloc(lib, ctx, size) for _ in range(4)]               # warm-up run: the slices exist before the window opens
    pre = _alloc(lib, ctx, size)                                     # alive before the window, dropped inside it
    for w in warm:
        assert lib.mi355_pool_free(ctx, None, w) == N.OK
    graph = C.c_void_p()
    assert lib.mi355_graph_begin_capture(ctx, None) == N.OK
    tmp = _alloc(lib, ctx, size)                                     # a temporary of the captured sequence (cache hit: no hipMalloc)
    keep = _alloc(lib, ctx, size)                                    # an output the caller keeps beyond the window
    assert tmp in warm and keep in warm
    assert lib.mi355_pool_free(ctx, None, tmp) == N.OK and lib.mi355_pool_free(ctx, None, pre) == N.OK
    again = _alloc(lib, ctx, size)                                   # inside the window too: the freed temporary is NOT handed out twice
    assert again not in (tmp, pre)
    # host synchronisation would abort the capture: refused, the window stays open
    host = (C.c_uint8 * 16)()
    assert lib.mi355_sync(ctx, None) == N.E_UNSUPPORTED and lib.mi355_read(ctx, None, host, C.c_void_p(keep), 16) == N.E_UNSUPPORTED
    assert lib.mi355_flush(ctx) == N.OK
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(graph)) == N.OK and graph.value
    assert lib.mi355_pool_free(ctx, None, again) == N.OK
    taken = [_alloc(lib, ctx, size) for _ in range(8)]               # drain every cached slice of the class and then some
    assert tmp not in taken and pre not in taken and again not in taken and keep not in taken
    assert lib.mi355_pool_free(ctx, None, keep) == N.OK              # the caller drops the output while the graph is alive
    assert _alloc(lib, ctx, size) != keep                            # ... still pinned: a replay would overwrite whoever got it
    assert lib.mi355_graph_replay(ctx, None, graph) == N.OK
    assert lib.mi355_graph_destroy(ctx, graph) == N.OK
    back = {_alloc(lib, ctx, size) for _ in range(4)}                # no driver call needed: the four pinned blocks are free memory again
    assert back == {tmp, pre, again, keep}


def test_block_allocated_in_one_window_and_freed_in_another_stays_pinned_by_both(lib, ctx):
    """Review finding (round 2): a block allocated inside graph A's window and dropped inside a later window B used to be held
    under B only -- destroying B handed it out again while A could still replay into it.  Both pins count."""
    size = 1 << 20
    warm = [_alloc(lib, ctx, size) for _ in range(3)]
    for w in warm:
        assert lib.mi355_pool_free(ctx, None, w) == N.OK
    ga, gb = C.c_void_p(), C.c_void_p()
    assert lib.mi355_graph_begin_capture(ctx, None) == N.OK
    x = _alloc(lib, ctx, size)                                       # baked into A's nodes
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(ga)) == N.OK
    assert lib.mi355_graph_begin_capture(ctx, None) == N.OK
    assert lib.mi355_pool_free(ctx, None, x) == N.OK                 # dropped inside B's window
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(gb)) == N.OK
    assert lib.mi355_graph_destroy(ctx, gb) == N.OK                  # B is gone, A is not
    taken = [_alloc(lib, ctx, size) for _ in range(6)]
    assert x not in taken                                            # A's replays still write there
    assert lib.mi355_graph_replay(ctx, None, ga) == N.OK
    assert lib.mi355_graph_destroy(ctx, ga) == N.OK
    assert _alloc(lib, ctx, size) == x                               # the last pin is gone: free memory again
    # an exclusive page released by a dead graph carries no event: the periodic cleanup must still see it as idle
    big = _alloc(lib, ctx, 64 << 20)
    gc_ = C.c_void_p()
    assert lib.mi355_graph_begin_capture(ctx, None) == N.OK
    assert lib.mi355_pool_free(ctx, None, big) == N.OK
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(gc_)) == N.OK
    assert lib.mi355_graph_destroy(ctx, gc_) == N.OK
    before = _device(lib)["frees"]
    for _ in range(12 * 1024):                                       # > 5000 x (1 + 64 MiB / 1 GiB) reservations without reuse
        q = _alloc(lib, ctx, 1024)
        assert lib.mi355_pool_free(ctx, None, q) == N.OK
    assert _device(lib)["frees"] == before + 1                       # the idle page went back to the driver


def test_capture_window_is_scoped_to_the_stream_under_capture(lib, ctx):
    """Review finding (round 2): the window used to be context-wide -- host syncs and reads on OTHER lanes were refused, and a
    block freed from another lane inside the window got no event and a graph pin.  ThreadLocal capture only concerns the
    captured stream: the other lanes are ordinary streams."""
    size = 1 << 20
    lane_a, lane_b, ev, graph = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.mi355_stream_create(ctx, C.byref(lane_a)) == N.OK and lib.mi355_stream_create(ctx, C.byref(lane_b)) == N.OK
    assert lib.mi355_event_create(ctx, C.byref(ev)) == N.OK
    blk = _alloc(lib, ctx, size, lane_b)
    host = (C.c_uint8 * 16)()
    assert lib.mi355_graph_begin_capture(ctx, lane_a) == N.OK
    # lane B is not captured: syncs, reads and events work
    assert lib.mi355_sync(ctx, lane_b) == N.OK
    assert lib.mi355_read(ctx, lane_b, host, C.c_void_p(blk), 16) == N.OK
    assert lib.mi355_event_record(ctx, ev, lane_b) == N.OK and lib.mi355_event_sync(ctx, ev) == N.OK
    # lane A is: refused, the window survives
    assert lib.mi355_sync(ctx, lane_a) == N.E_UNSUPPORTED
    assert lib.mi355_read(ctx, lane_a, host, C.c_void_p(blk), 16) == N.E_UNSUPPORTED
    assert lib.mi355_event_record(ctx, ev, lane_a) == N.OK and lib.mi355_event_sync(ctx, ev) == N.E_UNSUPPORTED
    events_before = _device(lib)["events"]
    assert lib.mi355_pool_free(ctx, lane_b, blk) == N.OK             # freed from lane B: ordered by an event, not pinned by the graph
    assert lib.mi355_graph_end_capture(ctx, lane_a, C.byref(graph)) == N.OK
    assert _alloc(lib, ctx, size, lane_b) == blk                     # lane B gets it back at once although the graph is alive
    assert _device(lib)["events"] >= events_before
    assert lib.mi355_graph_destroy(ctx, graph) == N.OK
    assert lib.mi355_event_destroy(ctx, ev) == N.OK
    assert lib.mi355_stream_destroy(ctx, lane_a) == N.OK and lib.mi355_stream_destroy(ctx, lane_b) == N.OK


def test_failed_capture_releases_what_the_window_pinned(lib, ctx):
    size = 1 << 20
    a = _alloc(lib, ctx, size)
    graph = C.c_void_p()
    assert lib.mi355_graph_begin_capture(ctx, None) == N.OK
    assert lib.mi355_pool_free(ctx, None, a) == N.OK
    lib.faketest_fail_end_capture(HIP_ERROR_INVALID_VALUE)
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(graph)) == N.E_EXECUTION and not graph.value
    assert _alloc(lib, ctx, size) == a                               # nothing will ever replay: reusable at once


def test_library_scratch_baked_into_a_graph_is_never_freed_or_regrown_under_it(lib, ctx):
    """Split-K slabs / re-laid-out operands live in library scratch whose address is a kernel argument of the captured
    launches: a later, larger eager call must not hipFree that buffer while the graph can still be replayed."""
    p1, p2, p3, graph = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.faketest_scratch_get(ctx, None, 0, 1 << 20, C.byref(p1)) == N.OK                # warm-up
    assert lib.mi355_graph_begin_capture(ctx, None) == N.OK
    assert lib.faketest_scratch_get(ctx, None, 0, 1 << 19, C.byref(p2)) == N.OK and p2.value == p1.value
    assert lib.faketest_scratch_get(ctx, None, 0, 1 << 22, C.byref(p3)) == N.E_UNSUPPORTED     # growing inside the window: refused
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(graph)) == N.OK
    frees = _device(lib)["frees"]
    assert lib.faketest_scratch_get(ctx, None, 0, 1 << 22, C.byref(p3)) == N.OK and p3.value != p1.value
    assert _device(lib)["frees"] == frees and lib.pooltest_inside_allocation(p1, 1 << 20) == 1  # retired, still mapped
    assert lib.mi355_graph_replay(ctx, None, graph) == N.OK
    assert lib.mi355_graph_destroy(ctx, graph) == N.OK
    assert _device(lib)["frees"] == frees + 1 and lib.pooltest_inside_allocation(p1, 1 << 20) == 0
    assert lib.pooltest_inside_allocation(p3, 1 << 22) == 1                                      # the current buffer is untouched


def test_a_destroyed_stream_takes_its_library_scratch_and_its_ticket_slot_with_it(lib, ctx):
    """The library keeps scratch buffers and one ticket slot per stream.  A service that creates a stream per request must not
    leak them (1024 ticket slots per context), and the driver may hand a dead stream's handle to the next hipStreamCreate, which
    must not inherit buffers; scratch a live graph replays against outlives the stream until that graph dies."""
    lib.faketest_ticket_slot.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.faketest_ticket_slot.restype = C.c_int32
    slot, seen = C.c_int64(-1), set()
    assert lib.faketest_ticket_slot(ctx, None, C.byref(slot)) == N.OK and slot.value == 0      # the compute stream came first
    for cycle in range(1100):                                  # more streams than a context has ticket slots, one at a time
        s = C.c_void_p()
        assert lib.mi355_stream_create(ctx, C.byref(s)) == N.OK
        assert lib.faketest_ticket_slot(ctx, s, C.byref(slot)) == N.OK, (cycle, lib.mi355_last_error(ctx))
        seen.add(slot.value)
        assert lib.mi355_stream_destroy(ctx, s) == N.OK
    assert seen == {1}                                         # the dead stream's slot is the next stream's
    live = [C.c_void_p() for _ in range(3)]
    for s in live:
        assert lib.mi355_stream_create(ctx, C.byref(s)) == N.OK and lib.faketest_ticket_slot(ctx, s, C.byref(slot)) == N.OK
        seen.add(slot.value)
    assert seen == {1, 2, 3}                                   # live streams never share one
    for s in live:
        assert lib.mi355_stream_destroy(ctx, s) == N.OK
    s, p, q = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.mi355_stream_create(ctx, C.byref(s)) == N.OK
    frees = _device(lib)["frees"]
    assert lib.faketest_scratch_get(ctx, s, 0, 1 << 20, C.byref(p)) == N.OK and lib.faketest_scratch_get(ctx, s, 2, 1 << 16, C.byref(q)) == N.OK
    assert lib.mi355_stream_destroy(ctx, s) == N.OK
    assert _device(lib)["frees"] == frees + 2 and lib.pooltest_inside_allocation(p, 1 << 20) == 0
    # pinned by a graph: retired with the stream, freed with the graph
    s2, graph = C.c_void_p(), C.c_void_p()
    assert lib.mi355_stream_create(ctx, C.byref(s2)) == N.OK
    assert lib.faketest_scratch_get(ctx, s2, 0, 1 << 20, C.byref(p)) == N.OK
    assert lib.mi355_graph_begin_capture(ctx, s2) == N.OK
    assert lib.faketest_scratch_get(ctx, s2, 0, 1 << 19, C.byref(q)) == N.OK and q.value == p.value
    assert lib.mi355_graph_end_capture(ctx, s2, C.byref(graph)) == N.OK
    frees = _device(lib)["frees"]
    assert lib.mi355_stream_destroy(ctx, s2) == N.OK
    assert _device(lib)["frees"] == frees and lib.pooltest_inside_allocation(p, 1 << 20) == 1
    assert lib.mi355_graph_destroy(ctx, graph) == N.OK
    assert _device(lib)["frees"] == frees + 1 and lib.pooltest_inside_allocation(p, 1 << 20) == 0
    # ... and so is the ticket slot a captured kernel counts on (advisor, round 4): a replay must never share its arrival words with
    # a new stream's launches -- the slot of a destroyed stream is handed out again only when the last graph carrying it is gone
    s3, s4, g3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.mi355_stream_create(ctx, C.byref(s3)) == N.OK and lib.faketest_ticket_slot(ctx, s3, C.byref(slot)) == N.OK
    pinned = slot.value
    assert lib.mi355_graph_begin_capture(ctx, s3) == N.OK
    assert lib.faketest_ticket_slot(ctx, s3, C.byref(slot)) == N.OK and slot.value == pinned      # a reduction inside the window
    assert lib.mi355_graph_end_capture(ctx, s3, C.byref(g3)) == N.OK
    assert lib.mi355_stream_destroy(ctx, s3) == N.OK
    assert lib.mi355_stream_create(ctx, C.byref(s4)) == N.OK and lib.faketest_ticket_slot(ctx, s4, C.byref(slot)) == N.OK
    assert slot.value != pinned                                # the graph is alive: its slot stays out of circulation
    assert lib.mi355_stream_destroy(ctx, s4) == N.OK and lib.mi355_graph_destroy(ctx, g3) == N.OK
    got = set()
    for _ in range(2):                                         # both free slots come back, the once-pinned one among them
        s5 = C.c_void_p()
        assert lib.mi355_stream_create(ctx, C.byref(s5)) == N.OK and lib.faketest_ticket_slot(ctx, s5, C.byref(slot)) == N.OK
        got.add(slot.value); live.append(s5)
    assert pinned in got
    for s5 in live[-2:]:
        assert lib.mi355_stream_destroy(ctx, s5) == N.OK
    # advisor, round 5: only the CAPTURE stream's slot is pinned by a window (another lane's reductions are real work, not graph nodes) ...
    cap, other, nxt, g6 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.mi355_stream_create(ctx, C.byref(cap)) == N.OK and lib.faketest_ticket_slot(ctx, cap, C.byref(slot)) == N.OK
    cap_slot = slot.value
    assert lib.mi355_stream_create(ctx, C.byref(other)) == N.OK
    assert lib.mi355_graph_begin_capture(ctx, cap) == N.OK
    assert lib.faketest_ticket_slot(ctx, other, C.byref(slot)) == N.OK                 # a reduction on another lane while the window is open
    other_slot = slot.value
    assert lib.faketest_ticket_slot(ctx, cap, C.byref(slot)) == N.OK and slot.value == cap_slot
    assert lib.mi355_stream_destroy(ctx, other) == N.OK
    assert lib.mi355_stream_create(ctx, C.byref(nxt)) == N.OK and lib.faketest_ticket_slot(ctx, nxt, C.byref(slot)) == N.OK
    assert slot.value == other_slot                              # not retired: free again at once
    # ... and a capture that FAILS pins nothing: the window's slot stays an ordinary slot of its stream and goes back with it
    lib.faketest_fail_end_capture(HIP_ERROR_INVALID_VALUE)
    assert lib.mi355_graph_end_capture(ctx, cap, C.byref(g6)) == N.E_EXECUTION and not g6.value
    assert lib.mi355_stream_destroy(ctx, cap) == N.OK and lib.mi355_stream_destroy(ctx, nxt) == N.OK
    back = set()
    for _ in range(2):
        s7 = C.c_void_p()
        assert lib.mi355_stream_create(ctx, C.byref(s7)) == N.OK and lib.faketest_ticket_slot(ctx, s7, C.byref(slot)) == N.OK
        back.add(slot.value); live.append(s7)
    assert back == {cap_slot, other_slot}                        # no slot leaked by the failed capture
    for s7 in live[-2:]:
        assert lib.mi355_stream_destroy(ctx, s7) == N.OK
    # the context's own streams are not the caller's to destroy
    own = C.c_void_p()
    assert lib.mi355_default_stream(ctx, C.byref(own)) == N.OK and lib.mi355_stream_destroy(ctx, own) == N.E_INVALID_ARGUMENT


def test_graph_destroy_waits_for_replays_and_reports_a_failed_wait(lib, ctx):
    mod, fn, graph, s2 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.mi355_module_load(ctx, b"FAKEHSACO", 9, C.byref(mod)) == N.OK
    assert lib.mi355_module_get_function(ctx, mod, b"k", C.byref(fn)) == N.OK
    assert lib.mi355_stream_create(ctx, C.byref(s2)) == N.OK
    one = (C.c_uint32 * 3)(1, 1, 1)
    assert lib.mi355_graph_begin_capture(ctx, None) == N.OK
    assert lib.mi355_launch(ctx, None, fn, one, one, 0, None, 0) == N.OK
    assert lib.mi355_graph_end_capture(ctx, None, C.byref(graph)) == N.OK
    assert lib.mi355_graph_replay(ctx, None, graph) == N.OK and lib.mi355_graph_replay(ctx, s2, graph) == N.OK
    before = _streams(lib)["syncs"]
    lib.faketest_fail_next(HIP_ERROR_LAUNCH_FAILURE, 0)              # the first wait finds a faulted stream
    assert lib.mi355_graph_destroy(ctx, graph) == N.OK
    assert _streams(lib)["syncs"] == before + 2                      # both streams it was replayed on were waited for
    assert lib.mi355_flush(ctx) == N.E_SERVER_UNHEALTHY
    rc, code, _, _, msg = _pop(lib, ctx)
    assert rc == N.OK and code == N.E_EXECUTION and "graph_destroy" in msg
    assert lib.mi355_stream_destroy(ctx, s2) == N.OK


def test_block_dropped_while_a_collective_is_in_flight_is_ordered_behind_the_comm_stream(lib, ctx):
    """A temporary handed to all_reduce / send and dropped before sync_collective: the pool orders the freeing stream behind the
    communication stream before the block can be reused (same stream: at once; other streams: through the block's event)."""
    comm = C.c_void_p()
    assert lib.mi355_comm_stream(ctx, C.byref(comm)) == N.OK
    a = _alloc(lib, ctx, 4096)
    w0 = _streams(lib)["waits"]
    assert lib.mi355_pool_free(ctx, None, a) == N.OK and _streams(lib)["waits"] == w0          # nothing in flight: no extra work
    b = _alloc(lib, ctx, 4096)
    lib.faketest_set_comm_dirty(ctx, 1)                              # a collective was issued and sync_collective has not run
    assert lib.mi355_pool_free(ctx, None, b) == N.OK
    log = _streams(lib)
    default = C.c_void_p()
    assert lib.mi355_default_stream(ctx, C.byref(default)) == N.OK
    assert log["waits"] == w0 + 1 and log["last_wait_stream"] == default.value               # the freeing stream waits on the comm fence
    lib.faketest_set_comm_dirty(ctx, 0)


def test_module_kernel_above_64_kib_of_lds_reports_the_limit_that_applies(lib, ctx):
    mod, fn = C.c_void_p(), C.c_void_p()
    assert lib.mi355_module_load(ctx, b"FAKEHSACO", 9, C.byref(mod)) == N.OK
    assert lib.mi355_module_get_function(ctx, mod, b"k", C.byref(fn)) == N.OK
    one = (C.c_uint32 * 3)(1, 1, 1)
    assert lib.mi355_launch(ctx, None, fn, one, one, 96 << 10, None, 0) == N.OK and lib.mi355_flush(ctx) == N.OK   # accepted by the driver
    lib.faketest_fail_next(0, HIP_ERROR_INVALID_VALUE)               # a driver that refuses the opt-in
    assert lib.mi355_launch(ctx, None, fn, one, one, 96 << 10, None, 0) == N.OK
    assert lib.mi355_flush(ctx) == N.E_SERVER_UNHEALTHY
    rc, code, req, mx, _ = _pop(lib, ctx)
    assert (rc, code, req, mx) == (N.OK, N.E_SHARED_MEMORY, 96 << 10, 64 << 10)


def test_context_calls_from_several_threads_are_serialised(lib, ctx):
    """Handles dropped by the garbage collector on another thread race client.empty() on the main one (ctypes releases the GIL
    inside foreign calls): the context's lock makes that safe.  4 threads x 2000 alloc / free pairs, then the books balance."""
    import threading
    errors = []

    def worker(seed):
        rng = np.random.default_rng(seed)
        mine = []
        for _ in range(2000):
            if mine and rng.random() < 0.5:
                if lib.mi355_pool_free(ctx, None, C.c_void_p(mine.pop(int(rng.integers(len(mine)))))) != N.OK:
                    errors.append("free")
            else:
                p = C.c_void_p()
                if lib.mi355_pool_alloc(ctx, None, int(rng.integers(1, 1 << 16)), C.byref(p)) != N.OK or not p.value:
                    errors.append("alloc")
                mine.append(p.value)
        for q in mine:
            lib.mi355_pool_free(ctx, None, C.c_void_p(q))
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    usage = N.MemoryUsage()
    assert not errors and lib.mi355_pool_usage(ctx, C.byref(usage)) == N.OK
    assert usage.number_allocs == 0 and usage.bytes_in_use == 0 and _device(lib)["bad_frees"] == 0
