"""The info buffer of a launch: scalars + static metadata + dynamic metadata.

Mirrors
  * crates/cubecl-core/src/codegen/scalars.rs:10-64     `ScalarBuilder`: scalar arguments grouped by element type in the
                                                        type's sort order, each group zero-padded to 8 bytes
  * crates/cubecl-core/src/codegen/metadata.rs:36-157   `MetadataBuilder`: buffer lengths of every binding, then a shape
                                                        offset and a stride offset per tensor (static part); all shapes,
                                                        then all strides (dynamic part); in u32 or u64 per the address type
  * crates/cubecl-core/src/codegen/info.rs:8-34         `InfoBuilder.finish`: [scalars | static | dynamic] as u64 words,
                                                        `dynamic_metadata_offset` = words before the dynamic part
  * crates/cubecl-runtime/src/server/base.rs:1027-1098  `KernelArguments`, `MetadataBindingInfo` (+ `custom` for kernels
                                                        compiled outside the reference's code generator)

This is the host half of the device ABI of a generated kernel (crates/cubecl-cpp/src/shared/kernel.rs:60-93: `info_st` =
the static part as a struct, the dynamic part read at `info + dynamic_metadata_offset`): the buffer travels as the LAST
pointer of the launch (crates/cubecl-cpp/src/hip/signature.rs:28-62; `ComputeClient.launch(..., info=...)` here).
Pure host logic; `upload` is the only call that touches a device.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

INFO_ALIGN = 8                  # cubecl-ir/src/metadata.rs:8   bytes
METADATA_BASE_LEN = 1           # :12  words per binding  (its length)
METADATA_EXT_LEN = 2            # :17  words per tensor   (shape offset, stride offset)

# cubecl_ir::ElemType's derived ordering (crates/cubecl-ir/src/type.rs:26-49, :74-79, :99-104, :125-131):
# Index < Float(E2M1 ... F64) < Int(I8 ... I64) < UInt(U8 ... U64) < Bool.  Values: (rank in that order, byte size, struct code).
SCALAR_TYPES: Dict[str, Tuple[int, int, Optional[str]]] = {
    "index": (0, 0, None),      # usize: takes the launch's address type
    "e2m1": (1, 1, "B"), "e2m1x2": (2, 1, "B"), "e2m3": (3, 1, "B"), "e3m2": (4, 1, "B"), "e4m3": (5, 1, "B"), "e5m2": (6, 1, "B"),
    "ue8m0": (7, 1, "B"), "f16": (8, 2, "e"), "bf16": (9, 2, "H"), "flex32": (10, 4, "f"), "f32": (11, 4, "f"), "tf32": (12, 4, "f"),
    "f64": (13, 8, "d"),
    "i8": (14, 1, "b"), "i16": (15, 2, "h"), "i32": (16, 4, "i"), "i64": (17, 8, "q"),
    "u8": (18, 1, "B"), "u16": (19, 2, "H"), "u32": (20, 4, "I"), "u64": (21, 8, "Q"),
    "bool": (22, 1, "?"),
}


class AddressType:
    """cubecl_ir::AddressType: the integer type kernels index with."""
    U32 = 4
    U64 = 8


@dataclass
class MetadataBindingInfo:
    """server/base.rs:1086-1098."""
    data: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.uint64))     # u64 words
    dynamic_metadata_offset: int = 0                                                   # in u64 words

    @staticmethod
    def custom(data: Sequence[int]) -> "MetadataBindingInfo":
        """For externally compiled kernels: the words are whatever that kernel's `info` struct expects."""
        return MetadataBindingInfo(np.asarray(data, dtype=np.uint64), 0)

    def to_bytes(self) -> bytes:
        return self.data.astype("<u8").tobytes()

    def upload(self, client):
        """A device buffer holding the words (the reference copies it pinned -> device per launch and may cache it by
        content, crates/cubecl-hip/src/compute/server.rs:128-148).  None for an empty info."""
        return client.create_from_slice(self.data.astype("<u8")) if self.data.size else None


class ScalarBuilder:
    """scalars.rs:10-64."""

    def __init__(self):
        self._groups: Dict[str, bytearray] = {}

    def push(self, value, dtype: str) -> None:
        rank, size, code = SCALAR_TYPES[dtype]
        if code is None:
            raise ValueError("an index scalar has no fixed width: push it as u32 or u64 (the launch's address type)")
        if dtype == "bf16":                 # value is the raw bit pattern (numpy has no bfloat16)
            value = int(value) & 0xFFFF
        self.push_raw(struct.pack("<" + code, value), dtype)

    def push_raw(self, data: bytes, dtype: str) -> None:
        if dtype not in SCALAR_TYPES:
            raise KeyError(dtype)
        self._groups.setdefault(dtype, bytearray()).extend(data)

    def _sorted(self) -> List[bytearray]:
        return [self._groups[k] for k in sorted(self._groups, key=lambda k: SCALAR_TYPES[k][0])]

    def len_aligned(self) -> int:
        """Words the scalars take: every type group rounded up to a whole word."""
        return sum(-(-len(v) // INFO_ALIGN) for v in self._groups.values())

    def finish(self) -> bytes:
        out = bytearray()
        for values in self._sorted():
            if not values:
                continue
            out += values + bytes(-len(values) % INFO_ALIGN)
        self._groups.clear()
        return bytes(out)


class MetadataBuilder:
    """metadata.rs:36-163.  One state per address type, as in the reference: a launch uses one of them."""

    def __init__(self):
        self._state = {AddressType.U32: ([], [], [], []), AddressType.U64: ([], [], [], [])}    # lens, shapes, strides, offsets

    def register_buffer(self, buffer_len: int, address_type: int) -> None:
        self._state[address_type][0].append(self._fit(buffer_len, address_type))

    def register_tensor(self, buffer_len: int, shape: Sequence[int], strides: Sequence[int], address_type: int) -> None:
        lens, shapes, strds, offsets = self._state[address_type]
        if len(shape) != len(strides):
            raise ValueError("shape and strides differ in rank")
        lens.append(self._fit(buffer_len, address_type))
        offsets.append(len(shapes))
        shapes.extend(self._fit(s, address_type) for s in shape)
        strds.extend(self._fit(s, address_type) for s in strides)

    @staticmethod
    def _fit(value: int, address_type: int) -> int:
        return int(value) & ((1 << (8 * address_type)) - 1)        # `as u32` truncates in the reference too

    def static_len(self, address_type: int) -> int:
        lens, _, _, offsets = self._state[address_type]
        return len(lens) * METADATA_BASE_LEN + len(offsets) * METADATA_EXT_LEN

    def dynamic_len(self, address_type: int) -> int:
        _, shapes, strides, _ = self._state[address_type]
        return len(shapes) + len(strides)

    def finish(self, address_type: int) -> Tuple[bytes, bytes]:
        """(static part, dynamic part), each zero-padded to whole words."""
        lens, shapes, strides, offsets = self._state[address_type]
        dt = "<u4" if address_type == AddressType.U32 else "<u8"
        stride_base = len(shapes)
        static = np.asarray(lens + offsets + [stride_base + o for o in offsets], dtype=dt).tobytes()
        dynamic = np.asarray(shapes + strides, dtype=dt).tobytes()
        for part in (lens, shapes, strides, offsets):
            part.clear()
        return static + bytes(-len(static) % INFO_ALIGN), dynamic + bytes(-len(dynamic) % INFO_ALIGN)


class InfoBuilder:
    """info.rs:8-36."""

    def __init__(self):
        self.scalars = ScalarBuilder()
        self.metadata = MetadataBuilder()

    def finish(self, address_type: int = AddressType.U32) -> MetadataBindingInfo:
        packing = INFO_ALIGN // address_type
        scalars_size = self.scalars.len_aligned()
        static_size = -(-self.metadata.static_len(address_type) // packing)
        dynamic_size = -(-self.metadata.dynamic_len(address_type) // packing)
        scalars = self.scalars.finish()
        static, dynamic = self.metadata.finish(address_type)
        assert (len(scalars), len(static), len(dynamic)) == (8 * scalars_size, 8 * static_size, 8 * dynamic_size)
        return MetadataBindingInfo(np.frombuffer(scalars + static + dynamic, dtype="<u8").astype(np.uint64),
                                   scalars_size + static_size)


def buffer_len(handle, elem_size: int, vector_size: int = 1) -> int:
    """What the generated launchers register as a binding's length (crates/cubecl-core/src/compute/launcher.rs:117,
    :144): the bytes IN USE -- offsets excluded -- over the size of one (vector) element.  Pinned by
    runtime_tests/metadata.rs:199-269: a 64-element allocation seen through a discontiguous view still has length 64;
    32 elements as vectors of 4 have length 8; 256 elements with 64 cut off at either end, as vectors of 2, have length 64."""
    return handle.size_in_used() // (elem_size * vector_size)


@dataclass
class KernelArguments:
    """server/base.rs:1027-1071: the bindings of a launch and its info."""
    resources: list = field(default_factory=list)
    info: MetadataBindingInfo = field(default_factory=MetadataBindingInfo)

    def with_buffer(self, binding) -> "KernelArguments":
        self.resources.append(binding)
        return self

    def with_buffers(self, bindings) -> "KernelArguments":
        self.resources.extend(bindings)
        return self

    def with_info(self, info: MetadataBindingInfo) -> "KernelArguments":
        self.info = info
        return self

    def launch(self, client, function, cube_count, cube_dim, shared_mem_bytes: int = 0) -> None:
        """`ComputeClient::launch(kernel, count, arguments)` for an externally built kernel: one pointer per binding in
        order, the info buffer last when there is one."""
        client.launch(function, cube_count, cube_dim, self.resources, self.info.upload(client), shared_mem_bytes)
