"""The host mirror of the reference surface -- `ComputeClient`, `Handle`, pitched tensors, the error model, `profile`,
graph capture, `to_client`, `KernelArguments` -- driven end to end on the CPU: the same Python classes the GPU tests use,
over the product's runtime / pool sources compiled against the fake HIP runtime (tests/fake_hip/).  The fake library is
injected into a `_Server` built by hand here; `Mi355Runtime.client()` itself never looks for anything but the real
libmi355cube.so (tests/test_abi_cpu.py::test_missing_library_raises).  Counterpart of the reference's client tests on its
DummyServer (crates/cubecl-runtime/tests/integration_test.rs)."""
import ctypes as C

import numpy as np
import pytest

from cubecl_amd import (AddressType, CubeCount, CubeDim, DeviceId, ElemType, InfoBuilder, KernelArguments, ServerError)
from cubecl_amd import _native as N
from cubecl_amd.info import buffer_len
from cubecl_amd.runtime import ComputeClient, _Server
from test_runtime_cpu import build_runtime_lib


@pytest.fixture(scope="module")
def fake():
    lib = C.CDLL(str(build_runtime_lib()))
    for name, (restype, argtypes) in N.PROTOTYPES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype, getattr(lib, name).argtypes = restype, argtypes
    lib.faketest_set_device.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
    lib.faketest_expect_params.argtypes = [C.c_uint32]
    lib.faketest_launch_log.argtypes = [C.POINTER(C.c_uint64)]
    lib.faketest_set_device(b"gfx950:sramecc+:xnack-", 64, 2)
    return lib


def _client(lib, index=0) -> ComputeClient:
    return ComputeClient(_Server(DeviceId(0, index), lib=lib))


@pytest.fixture()
def client(fake):
    c = _client(fake)
    yield c
    c._s.close()


def _log(lib):
    out = (C.c_uint64 * 20)()
    lib.faketest_launch_log(out)
    return list(out)


def test_buffers_handles_and_pitched_tensors_round_trip(client):
    x = np.arange(64, dtype=np.float32)
    h = client.create_from_slice(x)
    assert np.array_equal(client.read_one(h).view(np.float32), x)
    window = h.offset_start_by(16).offset_end_by(32)                         # bytes [16, 224): elements 4 .. 55
    assert window.size_in_used() == 208 and np.array_equal(client.read_one(window).view(np.float32), x[4:56])
    assert buffer_len(window, 4) == 52 and buffer_len(h, 4, vector_size=4) == 16
    client.write(window, np.full(52, -1.0, dtype=np.float32))
    back = client.read_one(h).view(np.float32)
    assert np.array_equal(back[:4], x[:4]) and np.all(back[4:56] == -1.0) and np.array_equal(back[56:], x[56:])
    assert client.read_one(client.empty(0)).size == 0                        # empty read (crates/cubecl-hip/tests/empty_read.rs)
    # empty_tensor applies the pitched layout policy: rows of 25 f32 (100 bytes) sit on a 128-byte pitch
    layout = client.empty_tensor((3, 5, 25), 4)
    assert layout.strides == (5 * 32, 32, 1) and layout.memory.size_in_used() == 15 * 128
    data = np.arange(375, dtype=np.float32).reshape(3, 5, 25)
    desc = layout.memory.copy_descriptor((3, 5, 25), layout.strides, 4)
    client.write_tensor(desc, data)
    assert np.array_equal(client.read_tensor(desc).view(np.float32).reshape(3, 5, 25), data)
    assert client.empty_tensor((4, 64), 4).strides == (64, 1) and client.empty_tensor((7,), 2).strides == (1,)
    with pytest.raises(ServerError) as e:
        client.read_tensor(layout.memory.copy_descriptor((3, 5, 25), (1, 3, 15), 4))
    assert e.value.kind == "UnsupportedStrides"
    assert client.io_optimized_vector_sizes(2) == [8, 4, 2, 1] and client.io_optimized_vector_sizes(4) == [4, 2, 1]
    assert client.features()["plane"] == {"Ops", "NonUniformControlFlow"}


def test_pool_is_what_client_empty_allocates_from(client):
    before = client.memory_usage()
    a = client.empty(1 << 20)
    ptr = a.device_ptr()
    u = client.memory_usage()
    assert u["number_allocs"] == before["number_allocs"] + 1 and u["bytes_in_use"] == before["bytes_in_use"] + (1 << 20)
    del a                                                                    # last handle gone: back to the pool, not to the driver
    assert client.memory_usage()["number_allocs"] == before["number_allocs"]
    assert client.empty(1 << 20).device_ptr() == ptr                         # and reused by the next request of the class
    client.memory_cleanup()
    u = client.memory_usage()
    assert u["bytes_reserved"] == 0 and u["device_bytes_total"] == 288 << 30
    client.allocation_mode(N.ALLOC_MODE_PERSISTENT)
    w = client.empty(1000)
    assert client.memory_usage()["bytes_padding"] == 24                      # exact size in 256-byte granules
    client.allocation_mode(N.ALLOC_MODE_AUTO)
    del w
    with pytest.raises(ServerError) as e:
        client.empty(100 << 30)
    assert e.value.kind == "BufferTooBig"


def test_launch_errors_surface_as_server_unhealthy_with_the_reference_taxonomy(client, fake):
    mod = client.load_module(b"FAKEHSACO")
    fn = client.get_function(mod, "abi_axpb")
    with pytest.raises(ServerError) as e:
        client.load_module(b"garbage!!!")
    assert e.value.kind == "CompilationError"
    with pytest.raises(ServerError) as e:
        client.get_function(mod, "missing")
    assert e.value.kind == "NotFound"
    src, dst = client.create_from_slice(np.arange(1000, dtype=np.uint32)), client.empty(4000)
    # the info buffer the reference's launcher would build for (u32 scale, u32 bias, two arrays), last pointer of the launch
    b = InfoBuilder()
    b.scalars.push(3, "u32"); b.scalars.push(7, "u32")
    b.metadata.register_buffer(buffer_len(src, 4), AddressType.U32); b.metadata.register_buffer(buffer_len(dst, 4), AddressType.U32)
    args = KernelArguments().with_buffers([src, dst]).with_info(b.finish())
    n0 = _log(fake)[0]
    fake.faketest_expect_params(3)
    args.launch(client, fn, CubeCount.new_1d(4), CubeDim.new(client, 1000))
    log = _log(fake)
    assert log[0] == n0 + 1 and log[3:9] == [4, 1, 1, 64, 8, 1]              # CubeDim::new: 8 planes of 64
    assert log[11:13] == [src.device_ptr(), dst.device_ptr()]
    info_words = (C.c_uint32 * 4).from_address(log[13])                      # "device" memory is host memory here
    assert list(info_words) == [3, 7, 1000, 1000]
    client.launch(fn, CubeCount.Static(0, 1, 1), CubeDim.new_1d(64), [src, dst])     # zero cube count: no-op
    assert _log(fake)[0] == n0 + 1
    client.flush()
    client.launch(fn, CubeCount.new_single(), CubeDim.new_1d(64), [src, dst], shared_mem_bytes=(160 << 10) + 8)
    client.launch(fn, CubeCount.new_single(), CubeDim.new_2d(64, 32), [src, dst])
    with pytest.raises(ServerError) as e:
        client.flush()
    assert e.value.kind == "ServerUnhealthy" and [x.kind for x in e.value.errors] == ["TooManyResources(SharedMemory)", "TooManyResources(Units)"]
    assert (e.value.errors[0].requested, e.value.errors[0].max) == ((160 << 10) + 8, 160 << 10) and e.value.errors[1].requested == 2048
    client.flush()                                                           # drained: healthy again
    client.sync()


def test_profile_capture_replay_and_to_client(client, fake):
    result, nanos = client.profile(lambda: 42, "answer")
    assert (result, nanos) == (42, 1_500_000)
    fn = client.get_function(client.load_module(b"FAKEHSACO"), "k")
    before = _log(fake)[0]
    graph = client.capture(lambda: [client.launch(fn, CubeCount.new_single(), CubeDim.new_single(), []) for _ in range(4)])
    assert _log(fake)[0] == before                                           # captured, not executed
    client.replay(graph); client.replay(graph)
    assert _log(fake)[0] == before + 8
    client.graph_destroy(graph)
    other = _client(fake, 1)
    try:
        data = np.array([0.0, 1.0, 2.0, 3.0, 4.0, 5.0], dtype=np.float32)   # runtime_tests/to_client.rs:28-29
        moved = client.to_client(client.create_from_slice(data), other, ElemType.F32)
        assert np.array_equal(other.read_one(moved).view(np.float32), data)
        assert client.device_key() == "mi355_dev0" and other.device_key() == "mi355_dev1"
    finally:
        del moved
        other._s.close()


def test_advertised_types_atomics_and_mma_properties_follow_the_reference_registration(client):
    """Row a18: what a backend must advertise so that the type-gated tests and launchers run instead of skipping --
    register_supported_types (crates/cubecl-cpp/src/shared/base.rs:322-375), parsed from the reference source when it is
    present (this container), and pinned literally so that the GPU box, which has no /root/reference, checks the same table."""
    import re
    from pathlib import Path
    from cubecl_amd import ElemType
    f = client.features()
    ALL = frozenset({"Conversion", "Arithmetic", "DotProduct", "Buffer"})
    full = {"INDEX", "U8", "U16", "U32", "U64", "I8", "I16", "I32", "I64", "BF16", "F16", "F32", "FLEX32", "F64", "BOOL"}
    want = {ElemType[n]: ALL for n in full}
    want.update({ElemType.F8E4M3: frozenset({"Conversion", "Buffer"}), ElemType.F8E5M2: frozenset({"Conversion", "Buffer"})})
    assert f["type_usage"] == want
    restricted = frozenset({"Add", "LoadStore", "Exchange"})
    every = frozenset({"LoadStore", "Exchange", "Add", "MinMax", "Bitwise", "CompareExchange"})
    assert f["atomic_type_usage"] == {ElemType.I32: every, ElemType.U32: every, ElemType.I64: restricted, ElemType.U64: restricted,
                                      ElemType.F32: restricted}
    assert f["address_types"] == {"U32", "U64"}
    src = Path("/root/reference/crates/cubecl-cpp/src/shared/base.rs")
    if src.exists():                                          # hold the literal table above to the reference's own text
        body = src.read_text()
        body = body[body.index("pub fn register_supported_types"):]
        body = body[: body.index("\n}\n")]
        sup = body[body.index("let supported_types"): body.index("let supported_atomic_types")]
        names = set(re.findall(r"(?:UIntKind|IntKind|FloatKind)::(\w+)", sup)) | set(re.findall(r"ElemType::(Index|Bool)", sup))
        assert {n.upper() for n in names} == full
        atom = body[body.index("let supported_atomic_types"): body.index("for ty in supported_types")]
        assert {n.upper() for n in re.findall(r"(?:UIntKind|IntKind|FloatKind)::(\w+)", atom)} == {"I32", "I64", "U32", "U64", "F32"}
        assert "for ty in [FloatKind::E4M3, FloatKind::E5M2]" in body and "TypeUsage::Conversion | TypeUsage::Buffer" in body
        assert "AtomicUsage::Add | AtomicUsage::LoadStore | AtomicUsage::Exchange" in body
        feats = Path("/root/reference/crates/cubecl-ir/src/features.rs").read_text()
        usage = feats[feats.index("pub enum TypeUsage"):]
        assert re.findall(r"^    (\w+),", usage[: usage.index("}")], re.M) == ["Conversion", "Arithmetic", "DotProduct", "Buffer"]
        atomic = feats[feats.index("pub enum AtomicUsage"):]
        assert re.findall(r"^    (\w+),", atomic[: atomic.index("}")], re.M) == ["LoadStore", "Exchange", "Add", "MinMax", "Bitwise",
                                                                                 "CompareExchange"]
    # TargetProperties.mma for MFMA (fields of cubecl_ir::MmaProperties, runtime_properties.rs:19-39): wave64, k-contiguous
    # operand registers (8 x 16-bit, 16 x 8-bit, one f32), accumulators 4 rows down a column, nothing duplicated
    m = client.target_properties()["mma"]
    assert (m["register_size_bits"], m["const_plane_size"]) == (32, 64)
    assert (m["register_layout_a"], m["register_layout_b"], m["register_layout_acc"]) == ("RowMajor", "ColMajor", "ColMajor")
    assert (m["register_duplication_a"], m["register_duplication_b"], m["register_duplication_acc"]) == (1, 1, 1)
    ce = m["contiguous_elements"]
    assert (ce("A", 16), ce("B", 16), ce("A", 8), ce("A", 32), ce("Accumulator", 32)) == (8, 8, 16, 1, 4)
    if src.exists():
        rp = Path("/root/reference/crates/cubecl-ir/src/runtime_properties.rs").read_text()
        st = rp[rp.index("pub struct MmaProperties"):]
        fields = re.findall(r"pub (\w+):", st[: st.index("\n}")])
        assert set(fields) - {"contiguous_elements"} == set(m) - {"contiguous_elements"} and "contiguous_elements" in fields


def _stream_log(lib):
    lib.faketest_stream_log.argtypes = [C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 4)()
    lib.faketest_stream_log(out)
    return list(out)


def test_logical_streams_wait_for_each_other_only_when_a_binding_crosses(client, fake):
    """ComputeClient::set_stream + MultiStream::resolve (crates/cubecl-runtime/src/stream/event.rs:20-78, :150-330) on the
    Python mirror: a StreamId maps to a lane (`id % max_streams`) with its own mi355_stream; an operation whose binding was
    created on another lane records an event there and makes the issuing lane's stream wait for it -- once per new
    cursor, not per use -- and the freeing lane waits for the borrower before the memory goes back to the pool."""
    a = client.with_stream(1)
    b = client.with_stream(2)
    assert (client.stream_id(), a.stream_id(), b.stream_id()) == (0, 1, 2)
    assert client.with_stream(1 + 128).stream_id() == 1                      # stream_index = id % max_streams
    assert a.stream.value and b.stream.value and a.stream.value != b.stream.value and not client.stream.value
    assert client.with_stream(1).stream.value == a.stream.value             # one native stream per lane

    w0 = _stream_log(fake)[1]
    x = a.create_from_slice(np.arange(256, dtype=np.float32))               # lane 1 binds and writes
    assert x.memory.lane == 1 and _stream_log(fake)[1] == w0                 # same lane: no wait
    assert np.array_equal(a.read_one(x).view(np.float32), np.arange(256, dtype=np.float32))
    assert _stream_log(fake)[1] == w0

    got = b.read_one(x)                                                      # lane 2 reads lane 1's buffer
    log = _stream_log(fake)
    assert np.array_equal(got.view(np.float32), np.arange(256, dtype=np.float32))
    assert log[1] == w0 + 1 and log[2] == b.stream.value                     # one hipStreamWaitEvent, on lane 2's stream
    b.read_one(x)                                                            # already waited for that cursor
    assert _stream_log(fake)[1] == w0 + 1
    a.write(x, np.ones(256, dtype=np.float32))                               # lane 1 touches it again: its cursor moves,
    log = _stream_log(fake)                                                  # and it first waits for the borrower (lane 2)
    assert log[1] == w0 + 2 and log[2] == a.stream.value
    assert np.array_equal(b.read_one(x).view(np.float32), np.ones(256, dtype=np.float32))
    assert _stream_log(fake)[1] == w0 + 3                                    # ... so lane 2 waits again
    assert b._lane.waits == 2 and a._lane.waits == 1

    # a kernel launch resolves every binding: two foreign lanes -> two waits on the launching lane
    y = b.create_from_slice(np.zeros(64, dtype=np.float32))
    c = client.with_stream(3)
    mod = c.load_module(b"FAKEHSACO")
    fn = c.get_function(mod, "k")
    from cubecl_amd.runtime import CubeCount, CubeDim
    before = _stream_log(fake)[1]
    fake.faketest_expect_params(2)
    c.launch(fn, CubeCount(1, 1, 1), CubeDim(64, 1, 1), [x, y])
    log = _stream_log(fake)
    assert log[1] == before + 2 and log[2] == c.stream.value
    c.launch(fn, CubeCount(1, 1, 1), CubeDim(64, 1, 1), [x, y])
    assert _stream_log(fake)[1] == before + 2

    # recycling: lane 1 frees x, which lanes 2 and 3 borrowed -> lane 1's stream waits for both before mi355_pool_free
    assert x.memory.users == {2, 3}
    before = _stream_log(fake)[1]
    stream_a = a.stream.value
    del x, got
    import gc
    gc.collect()
    log = _stream_log(fake)
    assert log[1] == before + 2 and log[2] == stream_a


def test_owner_lane_waits_for_borrowers_before_touching_its_memory_again(client, fake):
    """Write-after-read across logical streams: lane 2 read lane 1's buffer; when lane 1 writes it again, lane 1's stream first
    waits (on the device) for what lane 2 has issued -- once, then the buffer counts as lane 1's alone again."""
    a, b = client.with_stream(1), client.with_stream(2)
    x = a.create_from_slice(np.zeros(64, dtype=np.float32))
    b.read_one(x)                                                 # lane 2 borrows
    assert x.memory.users == {2}
    before = _stream_log(fake)[1]
    a.write(x, np.ones(64, dtype=np.float32))                     # the owner overwrites
    log = _stream_log(fake)
    assert log[1] == before + 1 and log[2] == a.stream.value and x.memory.users is None
    a.write(x, np.ones(64, dtype=np.float32))
    assert _stream_log(fake)[1] == before + 1
