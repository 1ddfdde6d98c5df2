"""Launch entry points of the hot path: tiled matmul and reductions over TensorHandles.

The reference keeps these launchers in the out-of-tree `cubek` crates (README.md:161-165); what
the snapshot pins is the operand contract -- TensorHandle {handle, shape, strides, dtype}
(crates/cubecl-std/src/tensor/handle.rs:13-23), strides in elements, transposition / broadcast
expressed through strides and classified by matrix_batch_layout
(crates/cubecl-std/src/tensor/matrix_batch_layout.rs:21-79) -- and the arithmetic
(runtime_tests/cmma.rs:695-722 for matmul; examples/sum_things/src/lib.rs:6-19 and the book's
reduce_matrix for sums).  These functions take exactly that contract and forward to the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import _native as N
from .runtime import ComputeClient, ElemType, Handle, ServerError
from .tensor import TensorHandle, matrix_batch_layout


def _matrix_operand(t: TensorHandle, name: str):
    """-> (transposed, leading dimension, batch count, batch stride)."""
    if t.rank() < 2:
        raise ServerError(N.E_INVALID_ARGUMENT, f"matmul: {name} must have rank >= 2")
    if t.shape[-2] == 0 or t.shape[-1] == 0:   # empty operand (K == 0 or empty output): nothing is read
        batch = 1
        for d in t.shape[:-2]:
            batch *= d
        return False, max(t.shape[-1], 1), batch, 0
    layout = matrix_batch_layout(t.strides)
    if layout.kind == "HighlyPermuted":
        raise ServerError(N.E_UNSUPPORTED_STRIDES, f"matmul: {name} is HighlyPermuted; make it contiguous first")
    rows, cols = t.shape[-2], t.shape[-1]
    rs, cs = t.strides[-2], t.strides[-1]
    if cs == 1:
        transposed, ld = False, rs
    elif rs == 1:
        transposed, ld = True, cs
    else:
        raise ServerError(N.E_UNSUPPORTED_STRIDES, f"matmul: {name} has no unit stride in its last two dims")
    if (rows == 1 or cols == 1) and ld < (rows if transposed else cols):
        ld = rows if transposed else cols
    batch_dims, batch_strides = t.shape[:-2], t.strides[:-2]
    batch = 1
    for d in batch_dims:
        batch *= d
    nontrivial = [(d, s) for d, s in zip(batch_dims, batch_strides) if d != 1]
    if not nontrivial:
        bstride = 0
    else:
        # collapsible iff each outer stride = inner stride * inner dim (or everything broadcast)
        bstride = nontrivial[-1][1]
        for (d_out, s_out), (d_in, s_in) in zip(nontrivial[:-1], nontrivial[1:]):
            if s_out != s_in * d_in:
                raise ServerError(N.E_UNSUPPORTED_STRIDES, f"matmul: {name} batch dims are not collapsible")
    return transposed, ld, batch, bstride


def _kernel_ready(client: ComputeClient, t: TensorHandle, name: str):
    """What the reference's matmul launchers do in front of the kernel: an operand whose strides the kernel cannot
    consume (matrix_batch_layout says HighlyPermuted, or the batch axes do not collapse) goes through
    into_contiguous first (crates/cubecl-std/src/tensor/matrix_batch_layout.rs:9-19, contiguous/launch.rs:5-20)."""
    try:
        return t, _matrix_operand(t, name)
    except ServerError as e:
        if e.code != N.E_UNSUPPORTED_STRIDES:
            raise
    t = into_contiguous(client, t)
    return t, _matrix_operand(t, name)


def matmul(client: ComputeClient, lhs: TensorHandle, rhs: TensorHandle, out: TensorHandle,
           algo: int = N.GEMM_ALGO_AUTO, acc: Optional[TensorHandle] = None) -> None:
    """out[.., m, n] = sum_k lhs[.., m, k] * rhs[.., k, n]   (f32 accumulate)   [+ acc[.., m, n]].

    `acc` is the C operand of cmma::execute(a, b, c, d) (frontend/cmma.rs:1066-1110): the product plus acc is formed
    in f32 and rounded once to out's dtype; acc has out's dtype and may be `out` itself (in-place accumulate).

    Layouts come from strides alone: a `rhs` of logical shape [k, n] with strides [1, k] is the
    reference tests' "ColMajor B" / Out = Lhs * Rhs^T form (cmma.rs:23) and takes the fast
    K-contiguous MFMA path; strides [n, 1] is row-major B.
    """
    m, k = lhs.shape[-2], lhs.shape[-1]
    k2, n = rhs.shape[-2], rhs.shape[-1]
    if k != k2 or out.shape[-2] != m or out.shape[-1] != n:
        raise ServerError(N.E_INVALID_ARGUMENT, f"matmul: shape mismatch {lhs.shape} x {rhs.shape} -> {out.shape}")
    if lhs.dtype != rhs.dtype:
        raise ServerError(N.E_INVALID_ARGUMENT, "matmul: lhs/rhs dtypes differ")
    lhs, (ta, lda, ba, sa) = _kernel_ready(client, lhs, "lhs")
    rhs, (tb, ldb, bb, sb) = _kernel_ready(client, rhs, "rhs")
    tc, ldc, bc, sc = _matrix_operand(out, "out")
    if tc:
        raise ServerError(N.E_UNSUPPORTED_STRIDES, "matmul: out must be row-major")
    batch = bc
    for b, name in ((ba, "lhs"), (bb, "rhs")):
        if b not in (1, batch):
            raise ServerError(N.E_INVALID_ARGUMENT, f"matmul: {name} batch {b} does not broadcast to {batch}")
    desc = N.GemmDesc(m=m, n=n, k=k, batch=batch, lda=lda, ldb=ldb, ldc=ldc,
                      stride_a=sa if ba == batch else 0, stride_b=sb if bb == batch else 0, stride_c=sc,
                      dtype_ab=int(lhs.dtype), dtype_c=int(out.dtype), trans_a=int(ta), trans_b=int(tb), algo=algo)
    if acc is None:
        client._s.check(client.lib.mi355_gemm(client.ctx, client.on(lhs, rhs, out), C.byref(desc), C.c_void_p(lhs.device_ptr()),
                                              C.c_void_p(rhs.device_ptr()), C.c_void_p(out.device_ptr())))
        return
    if acc.dtype != out.dtype or tuple(acc.shape) != tuple(out.shape):
        raise ServerError(N.E_INVALID_ARGUMENT, "matmul: acc must have out's shape and dtype")
    if tuple(acc.strides) != tuple(out.strides):
        acc = _like(client, acc, out)                       # bring C into D's layout (copy_into)
    client._s.check(client.lib.mi355_gemm_add(client.ctx, client.on(lhs, rhs, acc, out), C.byref(desc), C.c_void_p(lhs.device_ptr()),
                                              C.c_void_p(rhs.device_ptr()), C.c_void_p(acc.device_ptr()),
                                              C.c_void_p(out.device_ptr())))


def _like(client: ComputeClient, t: TensorHandle, model: TensorHandle) -> TensorHandle:
    """A copy of `t` laid out with `model`'s strides."""
    span = sum((d - 1) * s for d, s in zip(model.shape, model.strides)) + 1 if model.num_elems() else 0
    out = TensorHandle.new(client.empty(span * t.dtype.size()), model.shape, model.strides, t.dtype)
    copy_into(client, t, out)
    return out


def matmul_scaled(client: ComputeClient, lhs: TensorHandle, lhs_scales: TensorHandle, rhs: TensorHandle,
                  rhs_scales: TensorHandle, out: TensorHandle, *, block: int = 32, algo: int = N.GEMM_ALGO_AUTO) -> None:
    """Block-scaled matmul, out[.., m, n] = sum_k lhs[.., m, k] s_l[.., m, k/block] rhs[.., n, k] s_r[.., n, k/block].

    The layouts are the reference test's (runtime_tests/cmma.rs:1549-1560): `lhs` [.., m, k] row-major, `rhs` stored
    [.., n, k], scales [.., rows, k / block] (ue8m0).  F4E2M1X2 handles have a BYTE shape: [.., rows, k / 2]."""
    packed = 2 if lhs.dtype == ElemType.F4E2M1X2 else 1
    m, k = lhs.shape[-2], lhs.shape[-1] * packed
    n, k2 = rhs.shape[-2], rhs.shape[-1] * packed
    if k != k2 or out.shape[-2] != m or out.shape[-1] != n or lhs_scales.shape[-2] != m or rhs_scales.shape[-2] != n:
        raise ServerError(N.E_INVALID_ARGUMENT, f"matmul_scaled: shape mismatch {lhs.shape} x {rhs.shape} -> {out.shape}")
    if lhs_scales.shape[-1] * block != k or rhs_scales.shape[-1] * block != k:
        raise ServerError(N.E_INVALID_ARGUMENT, f"matmul_scaled: {lhs_scales.shape[-1]} scales per row do not cover k = {k} in blocks of {block}")
    for t, name in ((lhs, "lhs"), (rhs, "rhs"), (out, "out"), (lhs_scales, "lhs_scales"), (rhs_scales, "rhs_scales")):
        if t.strides[-1] != 1:
            raise ServerError(N.E_UNSUPPORTED_STRIDES, f"matmul_scaled: {name} must be contiguous along its last axis")

    def bstride(t, name):
        if t.rank() == 2:
            return 1, 0
        if t.rank() != 3:
            raise ServerError(N.E_UNSUPPORTED_STRIDES, f"matmul_scaled: {name} must have rank 2 or 3")
        return t.shape[0], t.strides[0]
    (ba, sa), (bb, sb), (bc, sc) = bstride(lhs, "lhs"), bstride(rhs, "rhs"), bstride(out, "out")
    (bsa, ssa), (bsb, ssb) = bstride(lhs_scales, "lhs_scales"), bstride(rhs_scales, "rhs_scales")
    batch = bc
    for b_, name in ((ba, "lhs"), (bb, "rhs"), (bsa, "lhs_scales"), (bsb, "rhs_scales")):
        if b_ not in (1, batch):
            raise ServerError(N.E_INVALID_ARGUMENT, f"matmul_scaled: {name} batch {b_} does not broadcast to {batch}")
    if (ba == batch) != (bsa == batch) or (bb == batch) != (bsb == batch):
        raise ServerError(N.E_INVALID_ARGUMENT, "matmul_scaled: an operand and its scales must be batched alike")
    desc = N.GemmScaledDesc(m=m, n=n, k=k, batch=batch, lda=lhs.strides[-2] * packed, ldb=rhs.strides[-2] * packed,
                            ldc=out.strides[-2], ld_sa=lhs_scales.strides[-2], ld_sb=rhs_scales.strides[-2],
                            stride_a=(sa if ba == batch else 0) * packed, stride_b=(sb if bb == batch else 0) * packed,
                            stride_c=sc, stride_sa=ssa if bsa == batch else 0, stride_sb=ssb if bsb == batch else 0,
                            dtype_a=int(lhs.dtype), dtype_b=int(rhs.dtype), dtype_c=int(out.dtype), block=block, algo=algo)
    client._s.check(client.lib.mi355_gemm_scaled(client.ctx, client.on(lhs, lhs_scales, rhs, rhs_scales, out), C.byref(desc), C.c_void_p(lhs.device_ptr()),
                                                 C.c_void_p(lhs_scales.device_ptr()), C.c_void_p(rhs.device_ptr()),
                                                 C.c_void_p(rhs_scales.device_ptr()), C.c_void_p(out.device_ptr())))


def gemm_scaled_select(client: ComputeClient, desc: N.GemmScaledDesc) -> int:
    algo = C.c_int32()
    client._s.check(client.lib.mi355_gemm_scaled_select(client.ctx, C.byref(desc), C.byref(algo)))
    return algo.value


def gemm_select(client: ComputeClient, desc: N.GemmDesc) -> int:
    algo = C.c_int32()
    client._s.check(client.lib.mi355_gemm_select(client.ctx, C.byref(desc), C.byref(algo)))
    return algo.value


def gemm_relayout_plan(client: ComputeClient, desc: N.GemmDesc):
    """(relayout_a, relayout_b): which operands mi355_gemm would copy into library scratch before the MFMA kernel."""
    ra, rb = C.c_int32(), C.c_int32()
    client._s.check(client.lib.mi355_gemm_relayout_plan(C.byref(desc), C.byref(ra), C.byref(rb)))
    return bool(ra.value), bool(rb.value)


def gemm_split_plan(client: ComputeClient, desc: N.GemmDesc) -> int:
    """K slices the 128x128 kernel's launcher cuts `desc` into on this device (1 = one plain launch)."""
    out = C.c_int32()
    client._s.check(client.lib.mi355_gemm_split_plan(C.byref(desc), int(client.properties().num_streaming_multiprocessors or 0), C.byref(out)))
    return out.value


_WORKSPACES: dict = {}


def _workspace(client: ComputeClient, n: int) -> Handle:
    key = (id(client._s), client.stream_id())      # one per logical stream: two lanes may reduce at the same time
    need = C.c_uint64()
    client._s.check(client.lib.mi355_reduce_workspace_bytes(client.ctx, n, C.byref(need)))
    ws = _WORKSPACES.get(key)
    if ws is None or ws.size < need.value or ws.memory.server is not client._s:
        ws = client.empty(need.value)
        _WORKSPACES[key] = ws
    return ws


def _require_flat_f32(t: TensorHandle, what: str) -> int:
    """Array-wide reductions take f32, bf16 or f16 input (16-bit elements are widened to f32 on load)."""
    if t.dtype not in (ElemType.F32, ElemType.BF16, ElemType.F16):
        raise ServerError(N.E_UNSUPPORTED, f"{what}: input must be f32, bf16 or f16")
    if not t.is_contiguous():
        raise ServerError(N.E_UNSUPPORTED_STRIDES, f"{what}: input must be contiguous")
    return t.num_elems()


def _consumable(client: ComputeClient, t: TensorHandle, view, what: str):
    """-> (tensor, view(tensor)): a permuted / sliced input the reduction kernels cannot walk goes through into_contiguous
    first, the same step the matmul launcher takes (contiguous/launch.rs:5-20)."""
    try:
        return t, view(t, what)
    except ServerError as e:
        if e.code != N.E_UNSUPPORTED_STRIDES:
            raise
    t = into_contiguous(client, t)
    return t, view(t, what)


def reduce_sum(client: ComputeClient, input: TensorHandle, output: TensorHandle) -> None:
    """Array-wide sum into output[0] (f32)."""
    input, n = _consumable(client, input, _require_flat_f32, "reduce_sum")
    ws = _workspace(client, n)
    client._s.check(client.lib.mi355_reduce_sum(client.ctx, client.on(input, output, ws), C.c_void_p(input.device_ptr()), int(input.dtype), n,
                                                C.c_void_p(output.device_ptr()), C.c_void_p(ws.device_ptr()), ws.size))


def argmax(client: ComputeClient, input: TensorHandle, out_index: TensorHandle,
           out_value: Optional[TensorHandle] = None) -> None:
    """Array-wide argmax: out_index[0] (u64) = lowest index of the maximum; NaN ranks highest."""
    input, n = _consumable(client, input, _require_flat_f32, "argmax")
    ws = _workspace(client, n)
    client._s.check(client.lib.mi355_argmax(
        client.ctx, client.on(input, out_value, out_index, ws), C.c_void_p(input.device_ptr()), int(input.dtype), n,
        C.c_void_p(out_value.device_ptr()) if out_value is not None else None,
        C.c_void_p(out_index.device_ptr()), C.c_void_p(ws.device_ptr()), ws.size))


def sum_argmax(client: ComputeClient, input: TensorHandle, out_sum: TensorHandle, out_index: TensorHandle,
               out_value: Optional[TensorHandle] = None) -> None:
    """Sum and argmax in one pass over the data."""
    input, n = _consumable(client, input, _require_flat_f32, "sum_argmax")
    ws = _workspace(client, n)
    client._s.check(client.lib.mi355_sum_argmax(
        client.ctx, client.on(input, out_sum, out_value, out_index, ws), C.c_void_p(input.device_ptr()), int(input.dtype), n, C.c_void_p(out_sum.device_ptr()),
        C.c_void_p(out_value.device_ptr()) if out_value is not None else None,
        C.c_void_p(out_index.device_ptr()), C.c_void_p(ws.device_ptr()), ws.size))


_VALUE_OPS = {"sum": N.REDUCE_SUM, "mean": N.REDUCE_MEAN, "max": N.REDUCE_MAX, "min": N.REDUCE_MIN, "prod": N.REDUCE_PROD}
_INDEX_OPS = {"argmax": N.REDUCE_ARGMAX, "argmin": N.REDUCE_ARGMIN}


def _op_code(op, table, what: str) -> int:
    if isinstance(op, str):
        if op not in table:
            raise ServerError(N.E_UNSUPPORTED, f"{what}: unknown operation '{op}' (one of {sorted(table)})")
        return table[op]
    return int(op)


def reduce(client: ComputeClient, input: TensorHandle, output: TensorHandle, op="sum") -> None:
    """Array-wide value reduction into output[0] (f32): op = "sum" | "mean" | "max" | "min" | "prod" (or a MI355_REDUCE_* code).
    max / min: NaN if any element is NaN, -0 < +0 (mi355_reduce, include/mi355cube.h)."""
    input, n = _consumable(client, input, _require_flat_f32, "reduce")
    ws = _workspace(client, n)
    client._s.check(client.lib.mi355_reduce(client.ctx, client.on(input, output, ws), C.c_void_p(input.device_ptr()), int(input.dtype), n,
                                            _op_code(op, _VALUE_OPS, "reduce"), C.c_void_p(output.device_ptr()), C.c_void_p(ws.device_ptr()), ws.size))


def argreduce(client: ComputeClient, input: TensorHandle, out_index: TensorHandle, out_value: Optional[TensorHandle] = None,
              op="argmax") -> None:
    """Array-wide index reduction: op = "argmax" | "argmin"; out_index[0] (u64) = lowest index of the extremum, NaN wins, -0 == +0."""
    input, n = _consumable(client, input, _require_flat_f32, "argreduce")
    ws = _workspace(client, n)
    client._s.check(client.lib.mi355_argreduce(
        client.ctx, client.on(input, out_value, out_index, ws), C.c_void_p(input.device_ptr()), int(input.dtype), n, _op_code(op, _INDEX_OPS, "argreduce"),
        C.c_void_p(out_value.device_ptr()) if out_value is not None else None,
        C.c_void_p(out_index.device_ptr()), C.c_void_p(ws.device_ptr()), ws.size))


def argmin(client: ComputeClient, input: TensorHandle, out_index: TensorHandle, out_value: Optional[TensorHandle] = None) -> None:
    argreduce(client, input, out_index, out_value, "argmin")


def argmax_combine(client: ComputeClient, records: Handle, count: int, index_base, out_value: Optional[Handle],
                   out_index: Optional[Handle]) -> None:
    """The combine step of the multi-GPU argmax on the device (mi355_argmax_combine_f32): folds `count` gathered records
    {f32 value, u32 unused, u64 local index} -- rank order, as all_gather delivers them -- into (value, GLOBAL index) with
    the single-GPU rule; `index_base[r]` is the first element of shard r.  Queued on the client's stream."""
    base = (C.c_uint64 * max(count, 1))(*[int(b) for b in (index_base or [0] * count)])
    client._s.check(client.lib.mi355_argmax_combine_f32(
        client.ctx, client.on(records, out_value, out_index), C.c_void_p(records.device_ptr()), count, base,
        C.c_void_p(out_value.device_ptr()) if out_value is not None else None,
        C.c_void_p(out_index.device_ptr()) if out_index is not None else None))


def sum_argmax_combine(client: ComputeClient, records: Handle, count: int, index_base, out_sum: Optional[Handle],
                       out_value: Optional[Handle], out_index: Optional[Handle]) -> None:
    """The whole combine step of the sharded sum + argmax on the device (mi355_sum_argmax_combine_f32): folds `count` gathered
    records {f32 value, f32 partial sum, u64 local index} -- rank order, as ONE all_gather delivers them -- into the global
    sum (partial sums added in rank order: the same bits on every rank) and (value, GLOBAL index) with the single-GPU rule."""
    base = (C.c_uint64 * max(count, 1))(*[int(b) for b in (index_base or [0] * count)])
    client._s.check(client.lib.mi355_sum_argmax_combine_f32(
        client.ctx, client.on(records, out_sum, out_value, out_index), C.c_void_p(records.device_ptr()), count, base,
        C.c_void_p(out_sum.device_ptr()) if out_sum is not None else None,
        C.c_void_p(out_value.device_ptr()) if out_value is not None else None,
        C.c_void_p(out_index.device_ptr()) if out_index is not None else None))


def _rows_view(t: TensorHandle, what: str):
    if t.dtype not in (ElemType.F32, ElemType.BF16, ElemType.F16):
        raise ServerError(N.E_UNSUPPORTED, f"{what}: input must be f32, bf16 or f16")
    if t.rank() == 0:
        raise ServerError(N.E_INVALID_ARGUMENT, f"{what}: rank-0 input")
    cols = t.shape[-1]
    rows = t.num_elems() // cols if cols else 0
    if t.strides[-1] != 1 and cols > 1:
        raise ServerError(N.E_UNSUPPORTED_STRIDES, f"{what}: last axis must have unit stride")
    row_stride = t.strides[-2] if t.rank() >= 2 else cols
    # leading dims must collapse onto one row stride (pitched row-major is fine)
    acc = row_stride
    for i in range(t.rank() - 3, -1, -1):
        acc *= t.shape[i + 1]
        if t.strides[i] != acc and t.shape[i] != 1:
            raise ServerError(N.E_UNSUPPORTED_STRIDES, f"{what}: leading dims are not collapsible")
    return rows, cols, row_stride


def reduce_sum_last_axis(client: ComputeClient, input: TensorHandle, output: TensorHandle) -> None:
    """The book's reduce_matrix (cubecl-book/src/getting-started/src/bin/v1-cpu.rs:7-15):
    output shape = input shape minus the last axis."""
    input, (rows, cols, stride) = _consumable(client, input, _rows_view, "reduce_sum_last_axis")
    client._s.check(client.lib.mi355_reduce_last_axis_sum(
        client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()), int(input.dtype), C.c_void_p(output.device_ptr()), rows, cols, stride))


def argmax_last_axis(client: ComputeClient, input: TensorHandle, output: TensorHandle) -> None:
    input, (rows, cols, stride) = _consumable(client, input, _rows_view, "argmax_last_axis")
    client._s.check(client.lib.mi355_reduce_last_axis_argmax(
        client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()), int(input.dtype), C.c_void_p(output.device_ptr()), rows, cols, stride))


def _axis_view(t: TensorHandle, axis: int, what: str):
    if t.dtype not in (ElemType.F32, ElemType.BF16, ElemType.F16):
        raise ServerError(N.E_UNSUPPORTED, f"{what}: input must be f32, bf16 or f16")
    if not t.is_contiguous():
        raise ServerError(N.E_UNSUPPORTED_STRIDES, f"{what}: input must be contiguous")
    rank = t.rank()
    if not -rank <= axis < rank:
        raise ServerError(N.E_INVALID_ARGUMENT, f"{what}: axis {axis} out of range for rank {rank}")
    axis %= rank
    outer = inner = 1
    for d in t.shape[:axis]:
        outer *= d
    for d in t.shape[axis + 1:]:
        inner *= d
    return outer, t.shape[axis], inner


def reduce_sum_axis(client: ComputeClient, input: TensorHandle, output: TensorHandle, axis: int) -> None:
    """Sum over one axis: output shape = input shape minus that axis (any axis; a strided view is made contiguous first)."""
    input, (outer, red, inner) = _consumable(client, input, lambda t, w: _axis_view(t, axis, w), "reduce_sum_axis")
    client._s.check(client.lib.mi355_reduce_axis_sum(client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()), int(input.dtype),
                                                     C.c_void_p(output.device_ptr()), outer, red, inner))


def argmax_axis(client: ComputeClient, input: TensorHandle, output: TensorHandle, axis: int) -> None:
    """Argmax over one axis (u32 indices along that axis; lowest index wins ties, NaN ranks highest)."""
    input, (outer, red, inner) = _consumable(client, input, lambda t, w: _axis_view(t, axis, w), "argmax_axis")
    client._s.check(client.lib.mi355_reduce_axis_argmax(client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()), int(input.dtype),
                                                        C.c_void_p(output.device_ptr()), outer, red, inner))


def reduce_axis(client: ComputeClient, input: TensorHandle, output: TensorHandle, axis: int, op="sum") -> None:
    """Value reduction over one axis (f32 out): op = "sum" | "mean" | "max" | "min" | "prod"; output shape = input shape minus that axis."""
    input, (outer, red, inner) = _consumable(client, input, lambda t, w: _axis_view(t, axis, w), "reduce_axis")
    client._s.check(client.lib.mi355_reduce_axis(client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()), int(input.dtype),
                                                 _op_code(op, _VALUE_OPS, "reduce_axis"), C.c_void_p(output.device_ptr()), outer, red, inner))


def argreduce_axis(client: ComputeClient, input: TensorHandle, output: TensorHandle, axis: int, op="argmax") -> None:
    """Index reduction over one axis (u32 indices along it): op = "argmax" | "argmin"; lowest index wins ties, NaN wins."""
    input, (outer, red, inner) = _consumable(client, input, lambda t, w: _axis_view(t, axis, w), "argreduce_axis")
    client._s.check(client.lib.mi355_argreduce_axis(client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()), int(input.dtype),
                                                    _op_code(op, _INDEX_OPS, "argreduce_axis"), C.c_void_p(output.device_ptr()), outer, red, inner))


def plane_reduce(client: ComputeClient, input: TensorHandle, output: TensorHandle, op: int, active: int = 64) -> None:
    """plane_sum / plane_prod / plane_max / plane_min / inclusive / exclusive sum and product over 64-lane planes."""
    n = input.num_elems()
    client._s.check(client.lib.mi355_plane_reduce_f32(client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()),
                                                      C.c_void_p(output.device_ptr()), n, active, op))


def plane_op(client: ComputeClient, input: TensorHandle, output: TensorHandle, op: int, plane: int = 64, arg: int = 0) -> None:
    """plane_all / plane_any / plane_elect / plane_broadcast / plane_shuffle / _xor / _up / _down / plane_ballot
    (crates/cubecl-core/src/frontend/plane.rs:62-216, :388-440) over planes of `plane` lanes (32 or 64).  `output` holds one f32
    per input, except for PLANE_BALLOT: 4 x u32 per plane."""
    n = input.num_elems()
    client._s.check(client.lib.mi355_plane_op_f32(client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()),
                                                  C.c_void_p(output.device_ptr()), n, plane, op, arg))


def identity(client: ComputeClient, output: TensorHandle) -> None:
    """tensor::identity::launch (crates/cubecl-std/src/tensor/identity.rs:36-84): `output`, a square matrix (possibly with
    pitched rows), becomes the identity matrix of its dtype."""
    if output.rank() != 2:
        raise ServerError(N.E_INVALID_ARGUMENT, "identity: input should be a matrix")
    if output.shape[0] != output.shape[1]:
        raise ServerError(N.E_INVALID_ARGUMENT, "identity: input should be a square matrix")
    if output.strides[1] != 1 and output.shape[1] > 1:
        raise ServerError(N.E_UNSUPPORTED_STRIDES, "identity: the matrix must be row-major")
    client._s.check(client.lib.mi355_fill_identity(client.ctx, client.on(output), C.c_void_p(output.device_ptr()), int(output.dtype),
                                                   output.shape[0], max(output.strides[0], output.shape[1])))


# ---- strided copies (crates/cubecl-std/src/tensor/contiguous/) -------------------------------------------------------
def copy_into(client: ComputeClient, input: TensorHandle, output: TensorHandle) -> None:
    """copy_into (contiguous/launch.rs:40-56): element q of `input`'s linear view goes to element q of `output`'s linear
    layout.  Both may be strided, and they may differ in rank as long as they hold the same number of elements."""
    if input.dtype.size() != output.dtype.size():
        raise ServerError(N.E_INVALID_ARGUMENT, "copy_into: element sizes differ")
    li, lo = N.TensorLayout.of(input.shape, input.strides), N.TensorLayout.of(output.shape, output.strides)
    client._s.check(client.lib.mi355_copy_strided(client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()), C.byref(li),
                                                  C.c_void_p(output.device_ptr()), C.byref(lo), input.dtype.size()))


def copy_plan(client: ComputeClient, input: TensorHandle, output: TensorHandle):
    """-> (path, bytes per access) copy_into takes for these two views (host-side only)."""
    li, lo = N.TensorLayout.of(input.shape, input.strides), N.TensorLayout.of(output.shape, output.strides)
    path, access = C.c_int32(), C.c_int32()
    rc = client.lib.mi355_copy_strided_plan(C.c_void_p(input.device_ptr()), C.byref(li), C.c_void_p(output.device_ptr()), C.byref(lo),
                                            input.dtype.size(), C.byref(path), C.byref(access))
    if rc != N.OK:
        raise ServerError(rc, "copy_plan: the two views cannot be copied into each other")
    return path.value, access.value


def into_contiguous(client: ComputeClient, input: TensorHandle) -> TensorHandle:
    """into_contiguous (contiguous/launch.rs:5-20): a new contiguous tensor with the view's elements."""
    handle = client.empty(input.num_elems() * input.dtype.size())
    output = TensorHandle.new_contiguous(input.shape, handle, input.dtype)
    copy_into(client, input, output)
    return output


def into_contiguous_pitched(client: ComputeClient, input: TensorHandle) -> TensorHandle:
    """into_contiguous_pitched (contiguous/launch.rs:22-37): like into_contiguous, rows padded to the pitch that
    ComputeClient::empty_tensor picks."""
    if input.rank() <= 1:
        return into_contiguous(client, input)
    output = TensorHandle.empty(client, input.shape, input.dtype)
    copy_into(client, input, output)
    return output


def into_contiguous_packed(client: ComputeClient, input: TensorHandle, packed_dim: int, shape, packing: int) -> TensorHandle:
    """into_contiguous_packed (contiguous/base.rs:254-293): `input` stores `packing` sub-word values per u32 / u8 word,
    packed along axis rank - 1 - packed_dim of the logical `shape`; the result stores the same values packed along the
    innermost axis."""
    rank = len(shape)
    if rank <= 1:
        return into_contiguous(client, input)
    out_shape = list(shape)
    out_shape[-1] = -(-out_shape[-1] // packing)
    output = TensorHandle.empty(client, out_shape, input.dtype)
    li, lo = N.TensorLayout.of(input.shape, input.strides), N.TensorLayout.of(output.shape, output.strides)
    logical = (C.c_int64 * rank)(*[int(d) for d in shape])
    client._s.check(client.lib.mi355_copy_packed(client.ctx, client.on(input, output), C.c_void_p(input.device_ptr()), C.byref(li),
                                                 C.c_void_p(output.device_ptr()), C.byref(lo), logical, packed_dim, packing,
                                                 input.dtype.size()))
    return output
