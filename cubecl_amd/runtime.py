"""Host-side mirror of the reference's runtime surface for the MI355X backend.

Names, argument meaning and error behaviour follow the reference (file:line under the
tracel-ai/cubecl checkout) so the parity tests read like the reference's own tests:

  Runtime           crates/cubecl-runtime/src/runtime.rs:14-52      -> Mi355Runtime
  ComputeClient     crates/cubecl-runtime/src/client.rs:44-48       -> ComputeClient
  Handle            crates/cubecl-runtime/src/server/handle.rs:10   -> Handle
  CopyDescriptor    crates/cubecl-runtime/src/server/base.rs:1102   -> CopyDescriptor
  CubeCount/CubeDim crates/cubecl-runtime/src/server/base.rs:1148,1251
  ReduceOperation   crates/cubecl-runtime/src/server/base.rs:623-628
  ServerError       crates/cubecl-runtime/src/server/base.rs:286-332
  DeviceId          crates/cubecl-common/src/device/base.rs:6

Everything here forwards to libmi355cube.so through the C ABI (include/mi355cube.h).  PyTorch
is not involved; there is no CPU fallback.  (The Rust shim a maintainer would compile instead
of this module lives in rust/cubecl-mi355; see INTEGRATION.md.)
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _native as N


# ----------------------------------------------------------------------------------------------
# value types
# ----------------------------------------------------------------------------------------------
class ElemType(enum.IntEnum):
    """cubecl_ir::ElemType restricted to what this path moves."""
    F32 = N.DTYPE_F32
    BF16 = N.DTYPE_BF16
    F16 = N.DTYPE_F16
    F64 = N.DTYPE_F64
    I32 = N.DTYPE_I32
    U32 = N.DTYPE_U32
    I64 = N.DTYPE_I64
    U64 = N.DTYPE_U64
    U8 = N.DTYPE_U8
    I8 = N.DTYPE_I8
    F8E4M3 = N.DTYPE_F8E4M3          # FloatKind::E4M3 (crates/cubecl-ir/src/types/scalar.rs)
    F8E5M2 = N.DTYPE_F8E5M2          # FloatKind::E5M2
    F4E2M1X2 = N.DTYPE_F4E2M1X2      # FloatKind::E2M1 packed in pairs (e2m1x2, fp4.rs:19-28): one byte per two elements
    UE8M0 = N.DTYPE_UE8M0            # FloatKind::UE8M0, the MX block scale
    # advertised for generated kernels (features().type_usage); this library's own entry points do not take them
    I16 = N.DTYPE_I16
    U16 = N.DTYPE_U16
    BOOL = N.DTYPE_BOOL
    FLEX32 = N.DTYPE_FLEX32
    INDEX = N.DTYPE_INDEX

    def size(self) -> int:
        return N.DTYPE_SIZE[int(self)]


class ReduceOperation(enum.IntEnum):
    """server/base.rs:623-628 (Sum, Mean) + Max/Min (API delta for argmax)."""
    Sum = N.REDUCE_SUM
    Mean = N.REDUCE_MEAN
    Max = N.REDUCE_MAX
    Min = N.REDUCE_MIN


@dataclass(frozen=True, order=True)
class DeviceId:
    type_id: int = 0
    index_id: int = 0


@dataclass(frozen=True)
class CubeDim:
    """server/base.rs:1251-1327."""
    x: int = 1
    y: int = 1
    z: int = 1

    @staticmethod
    def new(client, working_units: int) -> "CubeDim":
        """:1261-1295 -- (plane_size, planes): a power-of-two number of planes, at most 8 and at most what
        `working_units` fill, capped by max_units_per_cube; at least one plane."""
        p = client.properties()
        plane = int(p.plane_size_max)
        plane_count_max = max(1, int(working_units) // plane)
        planes = 1 << min(3, plane_count_max.bit_length() - 1)
        return CubeDim(plane, max(min(int(p.max_units_per_cube) // plane, planes), 1), 1)

    @staticmethod
    def new_single() -> "CubeDim":
        return CubeDim(1, 1, 1)

    @staticmethod
    def new_1d(x: int) -> "CubeDim":
        return CubeDim(x, 1, 1)

    @staticmethod
    def new_2d(x: int, y: int) -> "CubeDim":
        return CubeDim(x, y, 1)

    @staticmethod
    def new_3d(x: int, y: int, z: int) -> "CubeDim":
        return CubeDim(x, y, z)

    def num_elems(self) -> int:
        return self.x * self.y * self.z

    def can_contain(self, other: "CubeDim") -> bool:
        return self.x >= other.x and self.y >= other.y and self.z >= other.z


@dataclass(frozen=True)
class CubeCount:
    """CubeCount::Static(x, y, z) (server/base.rs:1148-1153; the Dynamic variant needs the IR's indirect dispatch)."""
    x: int = 1
    y: int = 1
    z: int = 1

    @staticmethod
    def Static(x: int, y: int = 1, z: int = 1) -> "CubeCount":
        return CubeCount(x, y, z)

    @staticmethod
    def new_single() -> "CubeCount":
        return CubeCount(1, 1, 1)

    @staticmethod
    def new_1d(x: int) -> "CubeCount":
        return CubeCount(x, 1, 1)

    @staticmethod
    def new_2d(x: int, y: int) -> "CubeCount":
        return CubeCount(x, y, 1)

    @staticmethod
    def new_3d(x: int, y: int, z: int) -> "CubeCount":
        return CubeCount(x, y, z)

    def is_empty(self) -> bool:
        """:1219-1224 -- a launch with an empty count is a no-op (client.rs:880-884)."""
        return self.x == 0 or self.y == 0 or self.z == 0

    def __str__(self) -> str:
        return f"({self.x}, {self.y}, {self.z})"


def cube_count_spread(max_cube_count: Sequence[int], num_cubes: int) -> tuple:
    """server/base.rs:1347-1374: a cube count over the limit in x is halved (rounding up) into y, then y into z."""
    limits = list(max_cube_count)
    count = [int(num_cubes), 1, 1]
    for i in range(2):
        if count[i] <= limits[i]:
            break
        while count[i] > limits[i]:
            count[i] = (count[i] + 1) // 2
            count[i + 1] *= 2
    return tuple(count)


@dataclass(frozen=True)
class CubeCountSelection:
    """server/base.rs:1156-1197: `Exact(count)` when the spread count is the requested one, else `Approx(count, actual)`
    -- some cubes are idle and the kernel has to check bounds."""
    count: CubeCount
    num_cubes_actual: int
    exact: bool

    @staticmethod
    def new(client, num_cubes: int) -> "CubeCountSelection":
        p = client.properties()
        c = cube_count_spread(tuple(p.max_cube_count), num_cubes)
        actual = c[0] * c[1] * c[2]
        return CubeCountSelection(CubeCount(*c), actual, actual == num_cubes)

    def has_idle(self) -> bool:
        return not self.exact

    def cube_count(self) -> CubeCount:
        return self.count


class ServerError(RuntimeError):
    """ServerError / LaunchError / IoError carried across the C ABI as (code, message)."""

    def __init__(self, code: int, message: str, errors: Optional[list] = None):
        self.code = code
        self.kind = N.ERROR_NAMES.get(code, f"code {code}")
        self.errors = errors or []      # ServerUnhealthy { errors }
        detail = "".join(f"\n  - {e.kind}: {e}" for e in self.errors)
        super().__init__(f"{self.kind}: {message}{detail}")
        self.message = message
        self.requested = 0
        self.max = 0


# ----------------------------------------------------------------------------------------------
# handles
# ----------------------------------------------------------------------------------------------
class _Memory:
    """A reservation from the server's memory pool (ManagedMemoryHandle); goes back to the pool when the last
    Handle drops, on the stream the client works on, so the pool may hand it out again stream-ordered."""

    __slots__ = ("server", "ptr", "size", "stream", "lane", "cursor", "users", "__weakref__")

    def __init__(self, server: "_Server", ptr: int, size: int, stream=None, lane: int = 0, cursor: int = 0):
        self.server, self.ptr, self.size, self.stream = server, ptr, size, stream
        # the logical stream that bound the memory and that stream's cursor at its last use there: what another stream
        # compares with what it has already waited for (MultiStream::resolve, stream/event.rs)
        self.lane, self.cursor = lane, cursor
        self.users = None        # other lanes that were handed this memory (they must be done before it is recycled)

    def __del__(self):
        try:
            if self.ptr and self.server is not None and self.server.ctx:
                srv = self.server
                for index in self.users or ():
                    # the reverse hazard of a cross-lane use (the reference pins the binding until a GC fence): the freeing
                    # lane waits, on the device, for what the borrowing lane has issued
                    ev = C.c_void_p()
                    if srv.lib.mi355_event_create(srv.ctx, C.byref(ev)) == N.OK:
                        srv.lib.mi355_event_record(srv.ctx, ev, srv.lanes[index].sys)
                        srv.lib.mi355_stream_wait_event(srv.ctx, self.stream, ev)
                        srv.lib.mi355_event_destroy(srv.ctx, ev)
                srv.lib.mi355_pool_free(srv.ctx, self.stream, C.c_void_p(self.ptr))
        except Exception:
            pass


@dataclass
class Handle:
    """server::Handle: ref-counted memory + byte offsets (handle.rs:10-21). Not a raw pointer."""
    memory: _Memory
    offset_start: Optional[int] = None
    offset_end: Optional[int] = None
    size: int = 0

    def clone(self) -> "Handle":
        return Handle(self.memory, self.offset_start, self.offset_end, self.size)

    def offset_start_by(self, offset: int) -> "Handle":
        return Handle(self.memory, (self.offset_start or 0) + offset, self.offset_end, self.size)

    def offset_end_by(self, offset: int) -> "Handle":
        return Handle(self.memory, self.offset_start, (self.offset_end or 0) + offset, self.size)

    def size_in_used(self) -> int:
        return self.size - (self.offset_start or 0) - (self.offset_end or 0)

    def binding(self) -> "Handle":
        return self

    # resolved server-side, like command.resource() (crates/cubecl-hip/src/compute/command.rs:57-62)
    def device_ptr(self) -> int:
        return self.memory.ptr + (self.offset_start or 0)

    def copy_descriptor(self, shape: Sequence[int], strides: Sequence[int], elem_size: int) -> "CopyDescriptor":
        return CopyDescriptor(self, tuple(shape), tuple(strides), elem_size)


@dataclass
class CopyDescriptor:
    handle: Handle
    shape: tuple
    strides: tuple
    elem_size: int


@dataclass
class MemoryLayout:
    """server::MemoryLayout { memory, strides } returned by empty_tensor / create_tensor."""
    memory: Handle
    strides: tuple


def contiguous_strides(shape: Sequence[int]) -> tuple:
    strides, acc = [], 1
    for d in reversed(tuple(shape)):
        strides.append(acc)
        acc *= d
    return tuple(reversed(strides))


def has_pitched_row_major_strides(shape: Sequence[int], strides: Sequence[int]) -> bool:
    """crates/cubecl-zspace/src/striding/layout_validation.rs:84-98: contiguous except for a
    padded second-to-last stride."""
    shape, strides = tuple(shape), tuple(strides)
    if len(shape) != len(strides):
        return False
    if len(shape) == 0:
        return True
    if strides[-1] != 1:
        return False
    if len(shape) == 1:
        return True
    if strides[-2] < shape[-1]:
        return False
    acc = strides[-2]
    for i in range(len(shape) - 3, -1, -1):
        acc *= shape[i + 1]
        if strides[i] != acc:
            return False
    return True


# ----------------------------------------------------------------------------------------------
# server + client
# ----------------------------------------------------------------------------------------------
MAX_STREAMS = 128   # StreamingConfig::max_streams default (config/streaming.rs:43-45)


class _Lane:
    """One logical stream of a server (StreamWrapper, stream/event.rs): the native stream, a cursor that advances with
    every operation resolved on it, and for every other lane the cursor of that lane this one has already waited for."""

    __slots__ = ("index", "sys", "cursor", "last_synced", "waits")

    def __init__(self, index: int, sys):
        self.index, self.sys, self.cursor, self.last_synced, self.waits = index, sys, 0, {}, 0


class _Server:
    """One per DeviceId (HipServer analogue); owns the C context."""

    def __init__(self, device: DeviceId, lib=None):
        self.lib = lib if lib is not None else N.load()     # `lib`: a differently built copy of the same ABI (tests)
        self.device = device
        ctx = C.c_void_p()
        rc = self.lib.mi355_ctx_create(device.index_id, C.byref(ctx))
        if rc != N.OK:
            raise ServerError(rc, self.lib.mi355_last_global_error().decode())
        self.ctx = ctx
        props = N.DeviceProps()
        self.check(self.lib.mi355_device_props(self.ctx, C.byref(props)))
        self.props = props
        self.comms: dict = {}
        self.lanes: dict = {0: _Lane(0, C.c_void_p(None))}     # lane 0 = the context's own compute stream

    def lane(self, stream_id: int) -> "_Lane":
        """The lane of a logical StreamId: `stream_id % MAX_STREAMS` (stream_index, stream/event.rs), its mi355_stream
        created on first use (EventStreamBackend::create_stream)."""
        index = int(stream_id) % MAX_STREAMS
        lane = self.lanes.get(index)
        if lane is None:
            sys = C.c_void_p()
            self.check(self.lib.mi355_stream_create(self.ctx, C.byref(sys)))
            lane = self.lanes[index] = _Lane(index, sys)
        return lane

    def check(self, rc: int) -> None:
        if rc == N.OK:
            return
        msg = self.lib.mi355_last_error(self.ctx).decode(errors="replace")
        if rc == N.E_SERVER_UNHEALTHY:
            errors = []
            while True:
                code, req, mx = C.c_int32(), C.c_uint64(), C.c_uint64()
                buf = C.create_string_buffer(512)
                if self.lib.mi355_error_pop(self.ctx, C.byref(code), C.byref(req), C.byref(mx), buf, 512) != N.OK:
                    break
                err = ServerError(code.value, buf.value.decode(errors="replace"))
                err.requested, err.max = req.value, mx.value
                errors.append(err)
            raise ServerError(rc, msg, errors)
        raise ServerError(rc, msg)

    def close(self) -> None:
        if self.ctx:
            for comm in self.comms.values():
                self.lib.mi355_comm_destroy(self.ctx, comm)
            self.comms.clear()
            for lane in self.lanes.values():
                if lane.sys:
                    self.lib.mi355_stream_destroy(self.ctx, lane.sys)
            self.lanes = {0: _Lane(0, C.c_void_p(None))}
            self.lib.mi355_ctx_destroy(self.ctx)
            self.ctx = None


_SERVERS: dict = {}


class Mi355Runtime:
    """`impl Runtime` for MI355X (counterpart of HipRuntime, crates/cubecl-hip/src/runtime.rs:254-328)."""

    @staticmethod
    def name() -> str:
        return "mi355"

    @staticmethod
    def require_array_lengths() -> bool:
        return True  # crates/cubecl-hip/src/runtime.rs:267-269

    @staticmethod
    def max_cube_count() -> tuple:
        return (2 ** 31 - 1, 65535, 65535)  # :271-273

    @staticmethod
    def can_read_tensor(shape: Sequence[int], strides: Sequence[int]) -> bool:
        return has_pitched_row_major_strides(shape, strides)  # :275-280

    @staticmethod
    def enumerate_devices() -> list:
        lib = N.load()
        n = C.c_int32(0)
        lib.mi355_device_count(C.byref(n))
        return [DeviceId(0, i) for i in range(n.value)]

    @staticmethod
    def client(device: DeviceId = DeviceId(0, 0)) -> "ComputeClient":
        server = _SERVERS.get(device)
        if server is None or not server.ctx:
            server = _Server(device)
            _SERVERS[device] = server
        return ComputeClient(server)


class ComputeClient:
    """ComputeClient<R> (client.rs:169-1369) over the C ABI.  Calls are issued directly on the
    calling thread: the C context is single-threaded per device, as the runner-thread contract
    requires."""

    def __init__(self, server: _Server):
        self._s = server
        self.lib = server.lib
        self.ctx = server.ctx
        self._lane = server.lanes[0]    # StreamId 0: the context's own compute stream (NULL in the C ABI)

    # -- logical streams -------------------------------------------------------------------------
    @property
    def stream(self):
        """The mi355_stream of the lane this client currently issues on."""
        return self._lane.sys

    def stream_id(self) -> int:
        return self._lane.index

    def set_stream(self, stream_id: int) -> "ComputeClient":
        """ComputeClient::set_stream (client.rs:217): everything this client issues from now on goes to the lane of
        `stream_id`.  Memory keeps the lane that created it; using it from another lane inserts the device-side wait
        (see `on`)."""
        self._lane = self._s.lane(stream_id)
        return self

    def with_stream(self, stream_id: int) -> "ComputeClient":
        """A second client of the same server pinned to another logical stream (what a thread with its own
        StreamId::current() sees in the reference)."""
        other = ComputeClient(self._s)
        return other.set_stream(stream_id)

    def on(self, *handles):
        """MultiStream::resolve (stream/event.rs) for one operation: advance this lane's cursor; for every binding that
        lives on another lane and whose cursor this lane has not yet waited for, record an event behind the origin lane's
        work (EventStreamBackend::flush) and make this lane's stream wait for it on the device
        (EventStreamBackend::wait_event = mi355_stream_wait_event).  Returns the stream to issue on."""
        lane = self._lane
        lane.cursor += 1
        origins = {}
        for h in handles:
            mem = getattr(getattr(h, "handle", h), "memory", None)
            if mem is None or mem.server is not self._s:
                continue
            if mem.lane == lane.index:
                mem.cursor = lane.cursor              # last use on its own lane
                if mem.users:
                    # the owner is about to touch memory other lanes were handed: whatever they have issued on it comes
                    # first (write-after-read across lanes; the reference leaves this one to the caller)
                    for index in mem.users:
                        origins[index] = self._s.lanes[index]
                    mem.users = None
            else:
                if mem.users is None:
                    mem.users = set()
                mem.users.add(lane.index)
                if lane.last_synced.get(mem.lane, -1) < mem.cursor:
                    origins[mem.lane] = self._s.lanes[mem.lane]
        for index, origin in origins.items():
            ev = C.c_void_p()
            self._s.check(self.lib.mi355_event_create(self.ctx, C.byref(ev)))
            self._s.check(self.lib.mi355_event_record(self.ctx, ev, origin.sys))
            self._s.check(self.lib.mi355_stream_wait_event(self.ctx, lane.sys, ev))
            self._s.check(self.lib.mi355_event_destroy(self.ctx, ev))
            lane.last_synced[index] = origin.cursor   # everything the origin lane has issued so far is covered
            lane.waits += 1
        return lane.sys

    # -- properties --------------------------------------------------------------------------
    def properties(self) -> N.DeviceProps:
        return self._s.props

    def features(self) -> dict:
        p = self._s.props
        cfgs = [(c.a_type, c.b_type, c.cd_type, c.m, c.n, c.k) for c in p.mma_configs[: p.num_mma_configs]]
        scaled = [(c.a_type, c.b_type, c.cd_type, c.scales_type, c.m, c.n, c.k, c.scales_factor)
                  for c in p.scaled_mma_configs[: p.num_scaled_mma_configs]]
        def names(bits, table):
            return frozenset(n for n, b in table.items() if bits & b)
        # register_supported_types (crates/cubecl-cpp/src/shared/base.rs:322-375): every ElemType a kernel may use with its
        # TypeUsage set, atomics with their AtomicUsage set, the two address types
        type_usage = {ElemType(e.dtype): names(e.usage, N.TYPE_USAGE) for e in p.type_usage[: p.num_type_usage]}
        atomic_usage = {ElemType(e.dtype): names(e.usage, N.ATOMIC_USAGE) for e in p.atomic_usage[: p.num_atomic_usage]}
        address = {n for n, b in (("U32", N.ADDRESS_TYPE_U32), ("U64", N.ADDRESS_TYPE_U64)) if p.address_types & b}
        return {"plane": {"Ops", "NonUniformControlFlow"} if p.plane_ops else set(), "cmma": set(cfgs), "mma": set(cfgs),
                "scaled_mma": set(scaled),          # features.matmul.scaled_mma (ScaledMmaConfig, cmma.rs:1493-1505)
                "type_usage": type_usage, "atomic_type_usage": atomic_usage, "address_types": address}

    def target_properties(self) -> dict:
        """Runtime::target_properties().mma for this device's matrix cores (runtime.rs:14-52; MmaProperties,
        crates/cubecl-ir/src/runtime_properties.rs:19-39).  `contiguous_elements(ident, elem_bits)` plays the role of the
        reference's ContiguousElements closure."""
        m = self._s.props.mma_properties
        lay = {N.LAYOUT_ROW_MAJOR: "RowMajor", N.LAYOUT_COL_MAJOR: "ColMajor"}

        def contiguous_elements(ident: str, elem_bits: int) -> int:
            if ident == "Accumulator":
                return m.contiguous_elements_acc
            return max(m.contiguous_elements_ab_bits // elem_bits, 1) if elem_bits < 32 else 1
        return {"mma": {"register_size_bits": m.register_size_bits, "const_plane_size": m.const_plane_size,
                        "register_layout_a": lay[m.register_layout_a], "register_layout_b": lay[m.register_layout_b],
                        "register_layout_acc": lay[m.register_layout_acc], "register_duplication_a": m.register_duplication_a,
                        "register_duplication_b": m.register_duplication_b, "register_duplication_acc": m.register_duplication_acc,
                        "contiguous_elements": contiguous_elements}}

    def io_optimized_vector_sizes(self, elem_size: int) -> list:
        width = self._s.props.load_width_bits // 8  # client.rs:1339
        out, v = [], max(width // elem_size, 1)
        while v >= 1:
            out.append(v)
            v //= 2
        return out

    def device_key(self) -> str:
        """client.rs:1353-1356: stable per-device identity that keys the device-level measurement caches."""
        return f"{Mi355Runtime.name()}_dev{self._s.device.index_id}"

    def measure_throughput(self, key, kernel_config, *, cache_enabled: Optional[bool] = None):
        """client.rs:1358-1368: peak of one probe under the reference's sampling protocol (cubecl_amd/roofline.py),
        cached per device."""
        from .roofline import ThroughputBenchmarker, ThroughputCache
        return ThroughputBenchmarker(ThroughputCache.get_for_device(self.device_key()), cache_enabled).measure(key, kernel_config)

    # -- memory ------------------------------------------------------------------------------
    def empty(self, size: int) -> Handle:
        """client.empty: a reservation from the memory pool (memory_manage.rs:1084), not a driver allocation."""
        ptr = C.c_void_p()
        lane = self._lane
        lane.cursor += 1
        self._s.check(self.lib.mi355_pool_alloc(self.ctx, lane.sys, size, C.byref(ptr)))
        return Handle(_Memory(self._s, ptr.value or 0, size, lane.sys, lane.index, lane.cursor), None, None, size)

    def create_from_slice(self, data) -> Handle:
        buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        handle = self.empty(buf.nbytes)
        self.write(handle, buf)
        return handle

    create = create_from_slice

    def write(self, handle: Handle, data) -> None:
        buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        if buf.nbytes > handle.size_in_used():
            raise ServerError(N.E_INVALID_ARGUMENT, "write larger than the handle")
        self._s.check(self.lib.mi355_write(self.ctx, self.on(handle), C.c_void_p(handle.device_ptr()),
                                           buf.ctypes.data_as(C.c_void_p), buf.nbytes))
        # the host buffer must outlive the copy (command.rs:402): complete it before returning
        self._s.check(self.lib.mi355_sync(self.ctx, self.stream))

    def empty_tensor(self, shape: Sequence[int], elem_size: int) -> MemoryLayout:
        """PitchedMemoryLayoutPolicy::apply (allocator.rs:21-72)."""
        shape = tuple(int(d) for d in shape)
        if len(shape) < 2:
            n = int(np.prod(shape)) if shape else 1
            return MemoryLayout(self.empty(n * elem_size), contiguous_strides(shape))
        width = shape[-1] * elem_size
        pitch = C.c_uint64()
        self._s.check(self.lib.mi355_pitched_row_bytes(self.ctx, width, C.byref(pitch)))
        height = int(np.prod(shape[:-1]))
        strides = [1] * len(shape)
        strides[-2] = pitch.value // elem_size
        for i in range(len(shape) - 3, -1, -1):
            strides[i] = strides[i + 1] * shape[i + 1]
        return MemoryLayout(self.empty(max(height, 1) * pitch.value), tuple(strides))

    def create_tensor(self, data: np.ndarray) -> MemoryLayout:
        layout = self.empty_tensor(data.shape, data.dtype.itemsize)
        self.write_tensor(CopyDescriptor(layout.memory, data.shape, layout.strides, data.dtype.itemsize), data)
        return layout

    def write_tensor(self, desc: CopyDescriptor, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data)
        if not has_pitched_row_major_strides(desc.shape, desc.strides):
            raise ServerError(N.E_UNSUPPORTED_STRIDES, f"unsupported strides {desc.strides} for shape {desc.shape}")
        if len(desc.shape) < 2 or desc.strides[-2] == desc.shape[-1]:
            return self.write(desc.handle, data)
        width = desc.shape[-1] * desc.elem_size
        rows = int(np.prod(desc.shape[:-1]))
        self._s.check(self.lib.mi355_write_2d(self.ctx, self.on(desc.handle), C.c_void_p(desc.handle.device_ptr()),
                                              desc.strides[-2] * desc.elem_size, data.ctypes.data_as(C.c_void_p),
                                              width, width, rows))
        self._s.check(self.lib.mi355_sync(self.ctx, self.stream))

    def read_one(self, handle: Handle) -> np.ndarray:
        """Returns the bytes; raises ServerError (incl. queued launch errors) like read_one."""
        n = handle.size_in_used()
        out = np.empty(n, dtype=np.uint8)
        self._s.check(self.lib.mi355_read(self.ctx, self.on(handle), out.ctypes.data_as(C.c_void_p),
                                          C.c_void_p(handle.device_ptr()), n))
        return out

    read_one_unchecked = read_one

    def read(self, handles: Iterable[Handle]) -> list:
        return [self.read_one(h) for h in handles]

    def read_tensor(self, desc: CopyDescriptor) -> np.ndarray:
        """read_one_tensor: gathers a (possibly pitched) tensor into contiguous bytes."""
        if not has_pitched_row_major_strides(desc.shape, desc.strides):
            raise ServerError(N.E_UNSUPPORTED_STRIDES, f"unsupported strides {desc.strides} for shape {desc.shape}")
        n = int(np.prod(desc.shape)) * desc.elem_size
        out = np.empty(n, dtype=np.uint8)
        if n == 0:
            return out
        if len(desc.shape) < 2 or desc.strides[-2] == desc.shape[-1]:
            self._s.check(self.lib.mi355_read(self.ctx, self.on(desc.handle), out.ctypes.data_as(C.c_void_p),
                                              C.c_void_p(desc.handle.device_ptr()), n))
            return out
        width = desc.shape[-1] * desc.elem_size
        rows = int(np.prod(desc.shape[:-1]))
        self._s.check(self.lib.mi355_read_2d(self.ctx, self.on(desc.handle), out.ctypes.data_as(C.c_void_p), width,
                                             C.c_void_p(desc.handle.device_ptr()), desc.strides[-2] * desc.elem_size,
                                             width, rows))
        return out

    # -- execution ---------------------------------------------------------------------------
    def launch(self, function, cube_count: CubeCount, cube_dim: CubeDim, resources: Sequence[Handle],
               info: Optional[Handle] = None, shared_mem_bytes: int = 0) -> None:
        """ComputeClient::launch for an externally built kernel: one pointer per binding, the
        `info` buffer last (crates/cubecl-cpp/src/hip/signature.rs:28-62).  Fire and forget:
        errors surface at flush/sync/read."""
        ptrs = [h.device_ptr() for h in resources]
        if info is not None:
            ptrs.append(info.device_ptr())
        arr = (C.c_void_p * max(len(ptrs), 1))(*ptrs)
        grid = (C.c_uint32 * 3)(cube_count.x, cube_count.y, cube_count.z)
        block = (C.c_uint32 * 3)(cube_dim.x, cube_dim.y, cube_dim.z)
        stream = self.on(*resources, *([info] if info is not None else []))
        self._s.check(self.lib.mi355_launch(self.ctx, stream, function, grid, block, shared_mem_bytes, arr, len(ptrs)))

    def load_module(self, image: bytes):
        mod = C.c_void_p()
        self._s.check(self.lib.mi355_module_load(self.ctx, image, len(image), C.byref(mod)))
        return mod

    def get_function(self, module, name: str):
        fn = C.c_void_p()
        self._s.check(self.lib.mi355_module_get_function(self.ctx, module, name.encode(), C.byref(fn)))
        return fn

    def to_client(self, src: Handle, dst_client: "ComputeClient", dtype: ElemType = None) -> Handle:
        """client.to_client(src, &dst_client, dtype) (client.rs:733-751): the bytes `src` has in use, on the other
        client's device.  A peer copy over xGMI, stream-ordered on both sides (no host round trip, no sync)."""
        nbytes = src.size_in_used()
        out = dst_client.empty(nbytes)
        self._s.check(self.lib.mi355_copy_to_ctx(self.ctx, self.on(src), C.c_void_p(src.device_ptr()), dst_client.ctx,
                                                 dst_client.on(out), C.c_void_p(out.device_ptr()), nbytes))
        return out

    def send(self, src: Handle, dtype: ElemType, device_ids: Sequence[DeviceId], peer: DeviceId) -> None:
        """ServerCommunication::send (server/base.rs:694-713) to the rank of `peer` in the sorted id list."""
        key = tuple(sorted(device_ids))
        n = src.size_in_used() // ElemType(dtype).size()
        self._s.check(self.lib.mi355_send(self.ctx, self._s.comms[key], self.on(src), C.c_void_p(src.device_ptr()), n,
                                          int(dtype), key.index(peer)))

    def recv(self, dst: Handle, dtype: ElemType, device_ids: Sequence[DeviceId], peer: DeviceId) -> None:
        """ServerCommunication::recv (server/base.rs:715-736)."""
        key = tuple(sorted(device_ids))
        n = dst.size_in_used() // ElemType(dtype).size()
        self._s.check(self.lib.mi355_recv(self.ctx, self._s.comms[key], self.on(dst), C.c_void_p(dst.device_ptr()), n,
                                          int(dtype), key.index(peer)))

    def flush(self) -> None:
        self._s.check(self.lib.mi355_flush(self.ctx))

    def sync(self) -> None:
        self._s.check(self.lib.mi355_sync(self.ctx, self.stream))

    def memory_usage(self) -> dict:
        """client.memory_usage(): MemoryUsage of the pool (memory_management/base.rs:8-28) plus the device totals."""
        u = N.MemoryUsage()
        self._s.check(self.lib.mi355_pool_usage(self.ctx, C.byref(u)))
        free, total = C.c_uint64(), C.c_uint64()
        self._s.check(self.lib.mi355_mem_info(self.ctx, C.byref(free), C.byref(total)))
        out = {name: int(getattr(u, name)) for name, _ in N.MemoryUsage._fields_ if name != "reserved"}
        out.update(device_bytes_free=free.value, device_bytes_total=total.value)
        return out

    def memory_cleanup(self) -> None:
        """client.memory_cleanup(): give everything the pool holds and nobody uses back to the driver."""
        self._s.check(self.lib.mi355_pool_cleanup(self.ctx, 1))

    def allocation_mode(self, mode: int) -> None:
        """client.allocation_mode(MemoryAllocationMode): N.ALLOC_MODE_AUTO / N.ALLOC_MODE_PERSISTENT."""
        self._s.check(self.lib.mi355_pool_mode(self.ctx, mode))

    def profile(self, fn, name: str = ""):
        """client.profile(closure, name): returns (result, device nanoseconds)."""
        token, nanos = C.c_uint64(), C.c_uint64()
        self._s.check(self.lib.mi355_profile_start(self.ctx, self.stream, C.byref(token)))
        result = fn()
        self._s.check(self.lib.mi355_profile_stop(self.ctx, self.stream, token, C.byref(nanos)))
        return result, nanos.value

    # -- graph capture (ComputeServer::begin_capture / end_capture / replay, server/base.rs:472-532) -----
    def capture(self, fn):
        """Runs `fn` (which enqueues work on this client's stream) inside a capture window and returns the
        instantiated graph.  Warm the sequence up once before capturing (base.rs:453-470)."""
        self._s.check(self.lib.mi355_graph_begin_capture(self.ctx, self.stream))
        try:
            fn()
        finally:
            g = C.c_void_p()
            rc = self.lib.mi355_graph_end_capture(self.ctx, self.stream, C.byref(g))
        self._s.check(rc)
        return g

    def replay(self, graph) -> None:
        self._s.check(self.lib.mi355_graph_replay(self.ctx, self.stream, graph))

    def graph_destroy(self, graph) -> None:
        self._s.check(self.lib.mi355_graph_destroy(self.ctx, graph))

    # -- collectives (ServerCommunication) ------------------------------------------------------
    def comm_init(self, device_ids: Sequence[DeviceId], unique_id: bytes, rank: Optional[int] = None) -> None:
        """comm_init: one communicator per sorted device set; rank = position of this device
        (crates/cubecl-cuda/src/compute/server.rs:669-703)."""
        ids = tuple(sorted(device_ids))
        if ids in self._s.comms:
            return
        if rank is None:
            rank = ids.index(self._s.device)
        uid = (C.c_uint8 * N.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        comm = C.c_void_p()
        self._s.check(self.lib.mi355_comm_init(self.ctx, uid, rank, len(ids), C.byref(comm)))
        self._s.comms[ids] = comm

    @staticmethod
    def comm_unique_id() -> bytes:
        lib = N.load()
        uid = (C.c_uint8 * N.UNIQUE_ID_BYTES)()
        rc = lib.mi355_comm_unique_id(uid)
        if rc != N.OK:
            raise ServerError(rc, lib.mi355_last_global_error().decode())
        return bytes(uid)

    def all_reduce(self, src: Handle, dst: Handle, dtype: ElemType, device_ids: Sequence[DeviceId],
                   op: ReduceOperation) -> None:
        ids = tuple(sorted(device_ids))
        comm = self._s.comms.get(ids)
        if comm is None:
            raise ServerError(N.E_COMM, "all_reduce before comm_init for this device set")
        count = src.size_in_used() // ElemType(dtype).size()  # get_nccl_dtype_count
        self._s.check(self.lib.mi355_all_reduce(self.ctx, comm, self.on(src, dst), C.c_void_p(src.device_ptr()),
                                                C.c_void_p(dst.device_ptr()), count, int(dtype), int(op)))

    def all_gather(self, src: Handle, dst: Handle, dtype: ElemType, device_ids: Sequence[DeviceId]) -> None:
        ids = tuple(sorted(device_ids))
        comm = self._s.comms.get(ids)
        if comm is None:
            raise ServerError(N.E_COMM, "all_gather before comm_init for this device set")
        count = src.size_in_used() // ElemType(dtype).size()
        self._s.check(self.lib.mi355_all_gather(self.ctx, comm, self.on(src, dst), C.c_void_p(src.device_ptr()),
                                                C.c_void_p(dst.device_ptr()), count, int(dtype)))

    def sync_collective(self) -> None:
        self._s.check(self.lib.mi355_sync_collective(self.ctx, self.stream))

    def sum_argmax_exchange(self, record: Handle, gathered: Handle, index_base, out_sum: Optional[Handle], out_value: Optional[Handle],
                            out_index: Optional[Handle], device_ids: Sequence[DeviceId]) -> None:
        """The exchange step of the sharded fused sum + argmax as ONE library call (`mi355_sum_argmax_exchange`: all-gather of the
        16-byte record, comm -> compute fence, combine kernel); the Rust server's `sum_argmax_exchange`
        (rust/cubecl-mi355/src/comm.rs).  `index_base`: one global start index per rank (host values)."""
        ids = tuple(sorted(device_ids))
        comm = self._s.comms.get(ids)
        if comm is None:
            raise ServerError(N.E_COMM, "sum_argmax_exchange before comm_init for this device set")
        base = (C.c_uint64 * max(len(ids), 1))(*[int(b) for b in (index_base or [0] * len(ids))])
        ptr = lambda h: C.c_void_p(h.device_ptr()) if h is not None else None
        self._s.check(self.lib.mi355_sum_argmax_exchange(self.ctx, comm, self.on(record, gathered, out_sum, out_value, out_index),
                                                         ptr(record), ptr(gathered), base, ptr(out_sum), ptr(out_value), ptr(out_index)))
