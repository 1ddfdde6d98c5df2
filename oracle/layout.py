"""CPU restatement (numpy index arithmetic) of the reference's strided copies: copy_into / into_contiguous and the
packed re-pack.

TEST INFRASTRUCTURE ONLY: imported by tests/ and bench.py's checks -- never by cubecl_amd (the product path).

Pinned against the known answers and CPU formulas the reference's own tests hold for this path
(crates/cubecl-std/src/tests/tensor/into_contiguous.rs): see tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np


def contiguous_strides(shape):
    """tests/tensor/into_contiguous.rs:5-14 / contiguous/base.rs:503-511 (compact_strides)."""
    strides = [1] * len(shape)
    cur = 1
    for d in range(len(shape) - 1, -1, -1):
        strides[d] = cur
        cur *= int(shape[d])
    return strides


def linear_offsets(n: int, shape, strides) -> np.ndarray:
    """Offset (in elements) of linear element q = 0..n-1 of a view: q is decomposed from the innermost axis outwards
    (contiguous/base.rs:43-64 index_offset_contiguous; the outermost axis keeps the modulo, so a q beyond the view
    wraps exactly like the reference's chain)."""
    q = np.arange(n, dtype=np.int64)
    off = np.zeros(n, dtype=np.int64)
    rem = q
    for d in range(len(shape) - 1, -1, -1):
        ext = max(int(shape[d]), 1)
        off += (rem % ext) * int(strides[d])
        rem = rem // ext
    return off


def copy_into(src: np.ndarray, in_shape, in_strides, dst: np.ndarray, out_shape, out_strides) -> np.ndarray:
    """copy_gpu_ref (contiguous/base.rs:295-389): element q of the input's linear view is written to position q of the
    output's linear layout.  src / dst are the flat buffers (1-D arrays of the element type); dst is updated in place
    and returned.  The two views may differ in rank (launch.rs:40-56, the rank-mismatch test :139-185)."""
    n = int(np.prod([int(s) for s in in_shape], dtype=np.int64)) if len(in_shape) else 1
    assert n == (int(np.prod([int(s) for s in out_shape], dtype=np.int64)) if len(out_shape) else 1)
    if n == 0:
        return dst
    dst[linear_offsets(n, out_shape, out_strides)] = src[linear_offsets(n, in_shape, in_strides)]
    return dst


def into_contiguous(src: np.ndarray, shape, strides) -> np.ndarray:
    """launch.rs:5-20: a fresh contiguous tensor holding the view's elements in row-major order."""
    n = int(np.prod([int(s) for s in shape], dtype=np.int64))
    out = np.zeros(n, dtype=src.dtype)
    return copy_into(src, shape, strides, out, shape, contiguous_strides(shape))


def pack_along(unpacked: np.ndarray, shape, pack_dim: int, packing: int, bits: int) -> np.ndarray:
    """The reference TEST's CPU packer (tests/tensor/into_contiguous.rs:16-58): row-major `unpacked` values are packed
    along axis pack_dim; consecutive values along that axis take increasing bit slots of one word."""
    shape = [int(s) for s in shape]
    storage_shape = list(shape)
    storage_shape[pack_dim] = -(-storage_shape[pack_dim] // packing)
    sstr = contiguous_strides(storage_shape)
    out = np.zeros(int(np.prod(storage_shape)), dtype=np.uint64)
    n = int(np.prod(shape))
    q = np.arange(n, dtype=np.int64)
    rem = q
    coords = [None] * len(shape)
    for d in range(len(shape) - 1, -1, -1):
        coords[d] = rem % shape[d]
        rem = rem // shape[d]
    slot = coords[pack_dim] % packing
    off = np.zeros(n, dtype=np.int64)
    for d in range(len(shape)):
        c = coords[d] // packing if d == pack_dim else coords[d]
        off += c * sstr[d]
    mask = (1 << bits) - 1
    np.bitwise_or.at(out, off, (unpacked.astype(np.uint64) & mask) << (slot * bits).astype(np.uint64))
    return out


def into_contiguous_packed(storage: np.ndarray, in_strides, shape, packed_dim: int, packing: int, word_bits: int = 32) -> np.ndarray:
    """into_contiguous_packed (contiguous/base.rs:254-293) through index_packed (:170-207): output word `pos` collects,
    in bit slot n, logical element pos * packing + n; that element's coordinates come from the div_mod chain over the
    LOGICAL shape, its storage word from the input strides with the packed axis' coordinate divided by `packing`.
    packed_dim counts from the innermost axis (in_packed_dim = rank - packed_dim - 1, :407)."""
    shape = [int(s) for s in shape]
    rank = len(shape)
    axis = rank - 1 - packed_dim
    out_shape = list(shape)
    out_shape[-1] = -(-out_shape[-1] // packing)
    words = int(np.prod(out_shape))
    bits = word_bits // packing
    mask = (1 << bits) - 1
    pos = np.arange(words, dtype=np.int64)
    acc = np.zeros(words, dtype=np.uint64)
    src = storage.astype(np.uint64)
    for n in range(packing):
        rem = pos * packing + n
        off = np.zeros(words, dtype=np.int64)
        slot = np.zeros(words, dtype=np.int64)
        for d in range(rank - 1, -1, -1):
            local = rem % max(shape[d], 1)
            rem = rem // max(shape[d], 1)
            if d == axis:
                slot = local % packing
                local = local // packing
            off += local * int(in_strides[d])
        acc |= ((src[off] >> (slot * bits).astype(np.uint64)) & np.uint64(mask)) << np.uint64(n * bits)
    return acc.astype(storage.dtype)
