"""The memory pool's logic (cubecl_amd/csrc/pool.cpp) on the host: the product source compiled with g++ against a fake HIP
runtime (tests/fake_hip/) whose streams complete when the test says so.  The reference tests its memory management the
same way, on a DummyServer over host memory (crates/cubecl-runtime/tests/dummy/, tests/integration_test.rs,
crates/cubecl-hip/tests/memory_pools.rs); the GPU versions of these cases are in tests/test_gpu_runtime.py
(test_pool_*).  Nothing here needs a device; nothing in the product uses tests/fake_hip."""
import ctypes as C
import random
import subprocess
from pathlib import Path

import pytest

from cubecl_amd import _native as N

ROOT = Path(__file__).resolve().parents[1]
FAKE = ROOT / "tests" / "fake_hip"
MiB = 1 << 20


@pytest.fixture(scope="module")
def lib():
    so = FAKE / "libpooltest.so"
    srcs = [ROOT / "cubecl_amd" / "csrc" / "pool.cpp", ROOT / "cubecl_amd" / "csrc" / "internal.hpp", FAKE / "fake_hip.cpp",
            FAKE / "hip" / "hip_runtime.h", ROOT / "include" / "mi355cube.h"]
    # -Bsymbolic: the fake hip* / mi355_* definitions inside this library win over the real ones when libmi355cube.so
    # (and with it libamdhip64) is already loaded in the test process
    if not so.exists() or so.stat().st_mtime < max(s.stat().st_mtime for s in srcs + [Path(__file__)]):
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-I", str(FAKE), "-o", str(so),
                        str(srcs[0]), str(srcs[2])], check=True)
    lib = C.CDLL(str(so))
    lib.pooltest_ctx_create.restype = C.c_void_p
    lib.pooltest_ctx_create.argtypes = [C.c_uint64]
    lib.pooltest_ctx_destroy.argtypes = [C.c_void_p]
    lib.pooltest_stream_create.restype = C.c_void_p
    lib.pooltest_stream_destroy.argtypes = [C.c_void_p]
    lib.pooltest_stream_complete.argtypes = [C.c_void_p, C.c_void_p]
    lib.pooltest_set_capturing.argtypes = [C.c_void_p, C.c_int32]
    lib.pooltest_set_capacity.argtypes = [C.c_uint64]
    lib.pooltest_last_error.restype = C.c_char_p
    lib.pooltest_last_error.argtypes = [C.c_void_p]
    lib.pooltest_counters.argtypes = [C.POINTER(C.c_uint64)]
    lib.pooltest_inside_allocation.argtypes = [C.c_void_p, C.c_uint64]
    lib.mi355_pool_alloc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.mi355_pool_free.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mi355_pool_cleanup.argtypes = [C.c_void_p, C.c_int32]
    lib.mi355_pool_mode.argtypes = [C.c_void_p, C.c_int32]
    lib.mi355_pool_usage.argtypes = [C.c_void_p, C.POINTER(N.MemoryUsage)]
    return lib


class Pool:
    def __init__(self, lib, max_page_size=72 << 30):
        self.lib, self.ctx = lib, lib.pooltest_ctx_create(max_page_size)
        lib.pooltest_set_capacity(288 << 30)

    def alloc(self, nbytes, stream=None, expect=N.OK):
        p = C.c_void_p()
        rc = self.lib.mi355_pool_alloc(self.ctx, stream, nbytes, C.byref(p))
        assert rc == expect, (rc, self.lib.pooltest_last_error(self.ctx))
        return p.value

    def free(self, ptr, stream=None, expect=N.OK):
        assert self.lib.mi355_pool_free(self.ctx, stream, ptr) == expect

    def usage(self):
        u = N.MemoryUsage()
        assert self.lib.mi355_pool_usage(self.ctx, C.byref(u)) == N.OK
        return u

    def device(self):
        out = (C.c_uint64 * 8)()
        self.lib.pooltest_counters(out)
        return dict(zip(("mallocs", "frees", "bad_frees", "live", "in_use", "events", "event_queries", "device_syncs"), out))

    def close(self):
        self.lib.pooltest_ctx_destroy(self.ctx)


@pytest.fixture()
def pool(lib):
    p = Pool(lib)
    before = p.device()
    yield p
    p.close()
    after = p.device()                      # destroy returns every page and event to the driver, with no bad free
    assert after["live"] == before["live"] and after["events"] == before["events"] and after["bad_frees"] == before["bad_frees"]


def test_slices_are_reused_without_driver_calls_and_usage_is_accounted(pool):
    d0 = pool.device()
    a = pool.alloc(1000)
    u = pool.usage()
    assert (u.number_allocs, u.bytes_in_use, u.bytes_padding, u.bytes_reserved, u.driver_allocs) == (1, 1000, 24, 2 * MiB, 1)
    pool.free(a)
    assert pool.alloc(1000) == a and pool.alloc(900) != a            # same class, most recently freed first; then a new slice
    u = pool.usage()
    assert (u.cache_hits, u.driver_allocs, u.driver_frees, u.number_allocs) == (1, 1, 0, 2)
    assert pool.device()["mallocs"] - d0["mallocs"] == 1              # one 2 MiB slab page served all of it
    # a steady-state loop never reaches the driver
    ptrs = [pool.alloc(300 * 1024) for _ in range(8)]
    calls = pool.device()["mallocs"]
    for _ in range(200):
        for p in ptrs:
            pool.free(p)
        ptrs = [pool.alloc(300 * 1024) for _ in range(8)]
    assert pool.device()["mallocs"] == calls and len(set(ptrs)) == 8
    pool.alloc(0)                                                     # empty allocation: NULL, not an error
    assert pool.usage().number_allocs == 10


def test_size_classes_pad_less_than_a_quarter_and_pages_hold_64_slices(pool):
    for nbytes in (1, 511, 512, 513, 640, 641, 4096, 5000, 65537, 1 << 20, (1 << 20) + 1, 32 * MiB):
        before = pool.usage()
        p = pool.alloc(nbytes)
        u = pool.usage()
        padding = u.bytes_padding - before.bytes_padding
        assert padding < max(0.25 * nbytes, 512) and u.bytes_in_use - before.bytes_in_use == nbytes
        assert pool.lib.pooltest_inside_allocation(p, nbytes + padding)     # the whole rounded slice lies in its page
    first = pool.alloc(3 * MiB)                                       # class 3 MiB: page = min(64 x 3 MiB, 256 MiB) = 192 MiB
    reserved = pool.usage().bytes_reserved
    rest = [pool.alloc(3 * MiB) for _ in range(63)]
    assert pool.usage().bytes_reserved == reserved                    # 64 slices out of one page
    extra = pool.alloc(3 * MiB)
    assert pool.usage().bytes_reserved == reserved + 192 * MiB and len({first, extra, *rest}) == 65


def test_reuse_is_stream_ordered(pool):
    lib = pool.lib
    s1, s2 = lib.pooltest_stream_create(), lib.pooltest_stream_create()
    a = pool.alloc(64 * 1024, s1)
    pool.free(a, s1)                                                  # event recorded on s1, not completed
    b = pool.alloc(64 * 1024, s2)
    assert b != a                                                     # another stream may not touch it yet
    assert pool.alloc(64 * 1024, s1) == a                             # the freeing stream gets it back at once
    pool.free(a, s1)
    lib.pooltest_stream_complete(pool.ctx, s1)
    assert pool.alloc(64 * 1024, s2) == a                             # once s1 has passed the free point, anyone may
    # exclusive pages follow the same rule
    big = pool.alloc(100 * MiB, s1)
    pool.free(big, s1)
    other = pool.alloc(100 * MiB, s2)
    assert other != big and pool.usage().driver_allocs == 3
    lib.pooltest_stream_complete(pool.ctx, s1)
    pool.free(other, s2)
    assert pool.alloc(100 * MiB, s2) in (big, other) and pool.usage().driver_allocs == 3
    for s in (s1, s2):
        lib.pooltest_stream_destroy(s)


def test_exclusive_pages_round_to_2_mib_and_are_reused_within_an_eighth(pool):
    a = pool.alloc(40 * MiB + 5)
    u = pool.usage()
    assert u.bytes_reserved == 42 * MiB and u.bytes_padding == 2 * MiB - 5
    pool.free(a)
    assert pool.alloc(38 * MiB) == a                                  # 42 <= 38 + 38 / 8
    pool.free(a)
    b = pool.alloc(36 * MiB)                                          # 42 > 36 + 4.5: a page of its own
    assert b != a and pool.usage().driver_allocs == 2
    assert pool.alloc(42 * MiB) == a and pool.usage().cache_hits == 2


def test_explicit_cleanup_releases_everything_idle_and_keeps_what_is_live(pool):
    keep = pool.alloc(10_000)
    gone = [pool.alloc(n) for n in (20_000, 50 * MiB, 70 * MiB)]
    for p in gone:
        pool.free(p)
    assert pool.usage().bytes_reserved == 2 * MiB + 2 * MiB + 50 * MiB + 70 * MiB
    assert pool.lib.mi355_pool_cleanup(pool.ctx, 1) == N.OK
    u = pool.usage()
    assert u.bytes_reserved == 2 * MiB and u.driver_frees == 3 and u.number_allocs == 1      # the page with the live slice stays
    assert pool.device()["device_syncs"] >= 1                         # "release the memory now" waits for the device first
    pool.free(keep)
    pool.lib.mi355_pool_cleanup(pool.ctx, 1)
    assert pool.usage().bytes_reserved == 0 and pool.device()["bad_frees"] == 0


def test_periodic_cleanup_returns_cached_pages_after_their_period_but_never_persistent_ones(pool):
    big = pool.alloc(64 * MiB)
    pool.free(big)
    pool.lib.pooltest_stream_complete(pool.ctx, None)
    assert pool.lib.mi355_pool_mode(pool.ctx, N.ALLOC_MODE_PERSISTENT) == N.OK
    weights = pool.alloc(1000)                                        # exact size in 256-byte granules, an exclusive page
    assert pool.usage().bytes_padding == 24 and pool.usage().bytes_reserved == 64 * MiB + 1024
    pool.free(weights)
    pool.lib.pooltest_stream_complete(pool.ctx, None)
    assert pool.lib.mi355_pool_mode(pool.ctx, N.ALLOC_MODE_AUTO) == N.OK and pool.lib.mi355_pool_mode(pool.ctx, 99) == N.E_INVALID_ARGUMENT
    small = pool.alloc(600)
    for i in range(6200):                                             # 5000 x (1 + 64 MiB / 1 GiB rounded) = 5000 reservations
        pool.free(small)
        small = pool.alloc(600)
        if i == 3000:
            assert pool.usage().driver_frees == 0                     # not yet due
    u = pool.usage()
    assert u.driver_frees == 1 and u.bytes_reserved == 2 * MiB + 1024  # the 64 MiB page went back; the persistent one stays
    pool.lib.mi355_pool_cleanup(pool.ctx, 1)
    assert pool.usage().driver_frees == 2                             # an explicit cleanup takes it too


def test_out_of_memory_releases_the_cache_and_retries_once(pool):
    pool.lib.pooltest_set_capacity(100 * MiB)
    a = pool.alloc(60 * MiB)
    pool.free(a)
    b = pool.alloc(50 * MiB)                                          # 60 cached + 50 does not fit: cleanup, then it does
    u = pool.usage()
    assert b is not None and u.driver_frees == 1 and u.bytes_reserved == 50 * MiB
    pool.alloc(80 * MiB, expect=N.E_OUT_OF_MEMORY)                    # a live block cannot be evicted: OutOfMemory, not BufferTooBig
    assert b"out of device memory" in pool.lib.pooltest_last_error(pool.ctx)
    small = Pool(pool.lib, max_page_size=1 << 30)
    small.alloc((1 << 30) + 1, expect=N.E_BUFFER_TOO_BIG)             # above max_page_size: BufferTooBig without asking the driver
    small.close()
    pool.lib.pooltest_set_capacity(288 << 30)


def test_inside_a_capture_window_only_the_cache_serves(pool):
    s2 = pool.lib.pooltest_stream_create()
    a, b = pool.alloc(8 * MiB), pool.alloc(8 * MiB)
    pool.free(a)
    pool.free(b, s2)
    d0 = pool.device()
    pool.lib.pooltest_set_capturing(pool.ctx, 1)
    assert pool.alloc(8 * MiB) == a                                   # same stream: no event query needed
    c = pool.alloc(8 * MiB)                                           # b was freed on another stream and may not be looked at:
    assert c not in (a, b)                                            # the next slice of the page already held is taken instead
    pool.alloc(100 * MiB, expect=N.E_UNSUPPORTED)                     # a fresh page would need the driver
    assert b"capture" in pool.lib.pooltest_last_error(pool.ctx)
    pool.alloc(5000, expect=N.E_UNSUPPORTED)                          # likewise the first slice of a class without a page
    pool.free(a)                                                      # no event is recorded inside the window ...
    assert pool.lib.mi355_pool_cleanup(pool.ctx, 1) == N.OK           # ... and nothing is released in it
    d1 = pool.device()
    assert d1["mallocs"] == d0["mallocs"] and d1["frees"] == d0["frees"] and d1["event_queries"] == d0["event_queries"]
    pool.lib.pooltest_set_capturing(pool.ctx, 0)
    assert pool.alloc(8 * MiB) == a                                   # event-less block: its own stream still reuses it
    pool.lib.pooltest_stream_destroy(s2)


def test_unknown_pointers_and_double_frees_are_refused(pool):
    a = pool.alloc(4096)
    pool.free(a + 16, expect=N.E_NOT_FOUND)
    pool.free(a)
    pool.free(a, expect=N.E_NOT_FOUND)
    pool.free(None)                                                   # freeing NULL is a no-op
    assert pool.usage().number_allocs == 0


def test_randomised_alloc_free_never_hands_out_overlapping_live_blocks(pool):
    rng = random.Random(0x5EED)
    streams = [None, pool.lib.pooltest_stream_create(), pool.lib.pooltest_stream_create()]
    live = {}                                                         # ptr -> (bytes, stream)
    for step in range(4000):
        if live and (rng.random() < 0.45 or len(live) > 300):
            p = rng.choice(list(live))
            nbytes, s = live.pop(p)
            pool.free(p, s)
        else:
            nbytes = rng.choice((rng.randint(1, 4096), rng.randint(4096, 1 << 20), rng.randint(1 << 20, 48 * MiB)))
            s = rng.choice(streams)
            p = pool.alloc(nbytes, s)
            assert p not in live and pool.lib.pooltest_inside_allocation(p, nbytes)
            live[p] = (nbytes, s)
        if step % 97 == 0:
            pool.lib.pooltest_stream_complete(pool.ctx, rng.choice(streams))
        if step % 1000 == 999:
            pool.lib.mi355_pool_cleanup(pool.ctx, rng.randint(0, 1))
    spans = sorted((p, p + n) for p, (n, _) in live.items())
    assert all(a_end <= b_start for (_, a_end), (b_start, _) in zip(spans, spans[1:]))
    u = pool.usage()
    assert u.number_allocs == len(live) and u.bytes_in_use == sum(n for n, _ in live.values())
    for p, (_, s) in live.items():
        pool.free(p, s)
    assert pool.usage().bytes_in_use == 0 and pool.usage().bytes_padding == 0
    for s in streams[1:]:
        pool.lib.pooltest_stream_destroy(s)
