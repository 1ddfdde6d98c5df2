"""TensorHandle (crates/cubecl-std/src/tensor/handle.rs:13-150) and the batched-matrix layout
classifier (crates/cubecl-std/src/tensor/matrix_batch_layout.rs:21-79)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _native as N
from .runtime import (ComputeClient, CopyDescriptor, ElemType, Handle, ServerError, contiguous_strides)

_NP = {ElemType.F32: np.float32, ElemType.F64: np.float64, ElemType.I32: np.int32, ElemType.U32: np.uint32,
       ElemType.I64: np.int64, ElemType.U64: np.uint64, ElemType.U8: np.uint8, ElemType.I8: np.int8,
       ElemType.BF16: np.uint16, ElemType.F16: np.float16,
       ElemType.F8E4M3: np.uint8, ElemType.F8E5M2: np.uint8,     # bit patterns (numpy has no bfloat16 / fp8)
       ElemType.F4E2M1X2: np.uint8, ElemType.UE8M0: np.uint8}    # packed e2m1 pairs (shape counts BYTES), ue8m0 scales
_BITS_ONLY = (ElemType.BF16, ElemType.F8E4M3, ElemType.F8E5M2, ElemType.F4E2M1X2, ElemType.UE8M0)


@dataclass
class TensorHandle:
    """{handle, metadata{shape, strides}, dtype}; strides in ELEMENTS, row-major, last stride 1."""
    handle: Handle
    shape: tuple
    strides: tuple
    dtype: ElemType

    # -- constructors (handle.rs:58-105, :155-192) ---------------------------------------------
    @staticmethod
    def new(handle: Handle, shape: Sequence[int], strides: Sequence[int], dtype: ElemType) -> "TensorHandle":
        return TensorHandle(handle, tuple(shape), tuple(strides), ElemType(dtype))

    @staticmethod
    def empty(client: ComputeClient, shape: Sequence[int], dtype: ElemType) -> "TensorHandle":
        layout = client.empty_tensor(shape, ElemType(dtype).size())  # possibly pitched strides
        return TensorHandle(layout.memory, tuple(shape), tuple(layout.strides), ElemType(dtype))

    @staticmethod
    def new_contiguous(shape: Sequence[int], handle: Handle, dtype: ElemType) -> "TensorHandle":
        return TensorHandle(handle, tuple(shape), contiguous_strides(shape), ElemType(dtype))

    @staticmethod
    def zeros(client: ComputeClient, shape: Sequence[int], dtype: ElemType) -> "TensorHandle":
        t = TensorHandle.empty(client, shape, dtype)
        client._s.check(client.lib.mi355_memset(client.ctx, client.on(t.handle), C.c_void_p(t.handle.device_ptr()), 0,
                                                t.handle.size_in_used()))
        return t

    @staticmethod
    def from_numpy(client: ComputeClient, array: np.ndarray, dtype: Optional[ElemType] = None) -> "TensorHandle":
        """Upload; `dtype` BF16 expects uint16 bit patterns, F8E4M3 / F8E5M2 uint8 bit patterns."""
        array = np.ascontiguousarray(array)
        if dtype is None:
            dtype = {np.dtype(v): k for k, v in _NP.items() if k not in _BITS_ONLY}[array.dtype]
        handle = client.create_from_slice(array)
        return TensorHandle.new_contiguous(array.shape, handle, dtype)

    @staticmethod
    def uniform(client: ComputeClient, shape: Sequence[int], dtype: ElemType, seed: int, tensor_id: int,
                lo: float, hi: float) -> "TensorHandle":
        """On-device counter-based fill, bit-identical to oracle_fill_uniform_f32 (+RNE cast)."""
        n = int(np.prod(shape))
        handle = client.empty(n * ElemType(dtype).size())
        client._s.check(client.lib.mi355_fill_uniform(client.ctx, client.on(handle), C.c_void_p(handle.device_ptr()),
                                                      int(dtype), n, seed, tensor_id, lo, hi))
        return TensorHandle.new_contiguous(shape, handle, dtype)

    # -- accessors -----------------------------------------------------------------------------
    def rank(self) -> int:
        return len(self.shape)

    def num_elems(self) -> int:
        return int(np.prod(self.shape)) if self.shape else 1

    def can_mut(self) -> bool:
        return True

    def binding(self) -> "TensorHandle":
        return self

    into_arg = binding

    def into_copy_descriptor(self) -> CopyDescriptor:
        return CopyDescriptor(self.handle, self.shape, self.strides, self.dtype.size())

    def is_contiguous(self) -> bool:
        return tuple(self.strides) == contiguous_strides(self.shape)

    def is_contiguous_pitched(self) -> bool:
        """contiguous/base.rs:475-501: only the second-to-last stride may carry padding."""
        rank = len(self.shape)
        if rank == 0:
            return True
        if self.strides[-1] != 1:
            return False
        if rank <= 1:
            return True
        if sorted(self.strides, reverse=True) != list(self.strides):
            return False
        return all(self.strides[i] == self.shape[i + 1] * self.strides[i + 1] for i in range(rank - 2))

    def permute(self, axes: Sequence[int]) -> "TensorHandle":
        """A view with the axes re-ordered (no data movement): what Burn's swap_dims / permute hand to the launchers."""
        if sorted(axes) != list(range(len(self.shape))):
            raise ValueError(f"permute: {axes} is not a permutation of {len(self.shape)} axes")
        return TensorHandle(self.handle, tuple(self.shape[a] for a in axes), tuple(self.strides[a] for a in axes), self.dtype)

    def to_numpy(self, client: ComputeClient) -> np.ndarray:
        raw = client.read_tensor(self.into_copy_descriptor())
        return raw.view(_NP[self.dtype]).reshape(self.shape)

    def device_ptr(self) -> int:
        return self.handle.device_ptr()


# ---- matrix_batch_layout (matrix_batch_layout.rs:21-79) -------------------------------------------
@dataclass(frozen=True)
class MatrixBatchLayout:
    kind: str                  # "Contiguous" | "MildlyPermuted" | "HighlyPermuted"
    transposed: bool = False
    batch_swap: bool = False


def matrix_batch_layout(strides: Sequence[int]) -> MatrixBatchLayout:
    """Classify a batched matrix operand from its strides alone."""
    strides = tuple(strides)
    rank = len(strides)
    if rank <= 1:
        return MatrixBatchLayout("Contiguous")
    row_stride, col_stride = strides[-2], strides[-1]
    if row_stride == 0 or col_stride == 0:
        return MatrixBatchLayout("HighlyPermuted")
    transposed = row_stride < col_stride
    batch_swap = False
    previous = row_stride
    for d in range(rank - 2):
        current = strides[rank - 3 - d]
        if current < row_stride or current < col_stride:
            if current == 0:
                batch_swap = True      # broadcast batch dimension
            else:
                return MatrixBatchLayout("HighlyPermuted")
        if current < previous:
            batch_swap = True
        previous = current
    if transposed or batch_swap:
        return MatrixBatchLayout("MildlyPermuted", transposed, batch_swap)
    return MatrixBatchLayout("Contiguous")
