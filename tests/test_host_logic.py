"""Host-side logic of the reference surface restated in cubecl_amd (no device needed)."""
import numpy as np
import pytest

from cubecl_amd import (CubeCount, CubeDim, DeviceId, ElemType, ReduceOperation, contiguous_strides,
                        has_pitched_row_major_strides, matrix_batch_layout)
from cubecl_amd.ops import _matrix_operand, _rows_view
from cubecl_amd.runtime import Handle, ServerError, _Memory
from cubecl_amd.tensor import TensorHandle


# crates/cubecl-std/src/tensor/matrix_batch_layout.rs tests (:86-190)
@pytest.mark.parametrize("strides,kind,transposed,batch_swap", [
    ((8, 4, 2, 1), "Contiguous", False, False),
    ((1,), "Contiguous", False, False),
    ((8, 4, 1, 2), "MildlyPermuted", True, False),
    ((4, 8, 2, 1), "MildlyPermuted", False, True),
    ((4, 8, 1, 2), "MildlyPermuted", True, True),
    ((8, 2, 4, 1), "HighlyPermuted", False, False),
    ((2, 8, 4, 1), "HighlyPermuted", False, False),
    ((0, 4, 2, 1), "MildlyPermuted", False, True),       # broadcast batch dim
    ((8, 4, 0, 1), "HighlyPermuted", False, False),      # broadcast inside the matrix
])
def test_matrix_batch_layout(strides, kind, transposed, batch_swap):
    layout = matrix_batch_layout(strides)
    assert layout.kind == kind
    if kind == "MildlyPermuted":
        assert (layout.transposed, layout.batch_swap) == (transposed, batch_swap)


def test_contiguous_and_pitched_strides():
    assert contiguous_strides((512, 2048, 2048)) == (4194304, 2048, 1)
    assert contiguous_strides((8192, 8192)) == (8192, 1)
    assert has_pitched_row_major_strides((4, 6), (8, 1))
    assert has_pitched_row_major_strides((2, 4, 6), (32, 8, 1))
    assert not has_pitched_row_major_strides((2, 4, 6), (40, 8, 1))
    assert not has_pitched_row_major_strides((4, 6), (1, 4))
    assert not has_pitched_row_major_strides((4, 6), (4, 1))


def _fake(shape, strides, dtype=ElemType.F32):
    return TensorHandle(Handle(_Memory(None, 0, 0), None, None, 0), tuple(shape), tuple(strides), dtype)


def test_matmul_operand_classification():
    # row-major [M, K]
    assert _matrix_operand(_fake((64, 32), (32, 1)), "lhs") == (False, 32, 1, 0)
    # "ColMajor B": logical [K, N] stored [N][K]
    assert _matrix_operand(_fake((32, 48), (1, 32)), "rhs") == (True, 32, 1, 0)
    # padded rows (PitchedMemoryLayoutPolicy)
    assert _matrix_operand(_fake((64, 30), (32, 1)), "lhs") == (False, 32, 1, 0)
    # C5: batch 512 x 2048^2 contiguous
    assert _matrix_operand(_fake((512, 2048, 2048), (4194304, 2048, 1)), "lhs") == (False, 2048, 512, 4194304)
    # broadcast batch
    assert _matrix_operand(_fake((8, 16, 16), (0, 16, 1)), "rhs") == (False, 16, 8, 0)
    # two collapsible batch dims
    assert _matrix_operand(_fake((2, 3, 16, 16), (768, 256, 16, 1)), "lhs") == (False, 16, 6, 256)
    with pytest.raises(ServerError):
        _matrix_operand(_fake((2, 3, 16, 16), (256, 512, 16, 1)), "lhs")


def test_rows_view():
    assert _rows_view(_fake((512, 8192), (8192, 1)), "r") == (512, 8192, 8192)
    assert _rows_view(_fake((64, 256, 1024), (262144, 1024, 1)), "r") == (16384, 1024, 1024)
    assert _rows_view(_fake((4, 30), (32, 1)), "r") == (4, 30, 32)
    with pytest.raises(ServerError):
        _rows_view(_fake((4, 30), (1, 4)), "r")


def test_handle_offsets():
    # handle.rs:85-103 / :118-121 and runtime_tests/metadata.rs:200-269 (in-use window)
    h = Handle(_Memory(None, 1000, 256), None, None, 256)
    assert h.size_in_used() == 256
    h2 = h.offset_start_by(64).offset_end_by(32)
    assert h2.size_in_used() == 160 and h2.device_ptr() == 1064
    assert h2.offset_start_by(16).offset_start == 80


def test_value_types():
    assert CubeDim.new_1d(64).num_elems() == 64
    assert CubeCount.Static(1, 1, 1) == CubeCount(1, 1, 1)
    assert sorted([DeviceId(0, 3), DeviceId(0, 1)])[0].index_id == 1
    assert ElemType.BF16.size() == 2 and ElemType.F32.size() == 4
    assert int(ReduceOperation.Sum) == 0 and int(ReduceOperation.Mean) == 1


# ---- throughput curve host logic (crates/cubecl-runtime/src/throughput/curve.rs tests :186-259, ported) ------------
def test_working_set_sweep_known_answers():
    from cubecl_amd.throughput import MIN_WORKING_SET as m, working_set_sweep
    assert working_set_sweep(8 * m) == [m, 2 * m, 4 * m, 8 * m]
    assert working_set_sweep(12 * m) == [m, 2 * m, 4 * m, 8 * m]      # stops at the last power of two that fits
    assert working_set_sweep(m // 4) == [m // 4]                       # below the minimum: the cap alone


def test_memory_curve_interpolates_clamps_and_drops_unusable_points():
    from cubecl_amd.throughput import MemoryAccess, MemoryCurve, MemoryPoint, _log2
    MB = 1 << 20
    curve = MemoryCurve(MemoryAccess.Read, [MemoryPoint(4 * MB, 200.0), MemoryPoint(MB, 100.0), MemoryPoint(MB, 999.0)])
    assert [p.bytes for p in curve.points()] == [MB, 4 * MB] and curve.points()[0].bytes_per_s == 100.0
    assert abs(curve.ceiling_at(2 * MB) - 150.0) < 1e-6                # geometric midpoint -> half the rate span
    assert abs(curve.ceiling_at(MB) - 100.0) < 1e-6 and abs(curve.ceiling_at(4 * MB) - 200.0) < 1e-6
    assert abs(curve.ceiling_at(1) - 100.0) < 1e-6 and abs(curve.ceiling_at(2 ** 64 - 1) - 200.0) < 1e-6
    empty = MemoryCurve(MemoryAccess.Read, [MemoryPoint(MB, float("nan")), MemoryPoint(0, 5.0), MemoryPoint(MB, 0.0)])
    assert empty.points() == () and empty.ceiling_at(MB) is None
    assert _log2(1) == 0.0 and _log2(MB) == 20.0 and _log2(1 << 63) == 63.0 and _log2(0) == 0.0
    assert _log2(2 * MB) < _log2(3 * MB) < _log2(4 * MB)


def test_roofline_time_limit_is_the_slower_bound_plus_a_launch():
    from cubecl_amd.throughput import Bounds, Thresholds, Work
    w = Work(compute_ops=2 * 8192 ** 3, bytes=3 * 8192 * 8192 * 2)
    b = Bounds(2.5e15, 8e12, 2e-6, w, Thresholds.uniform(0.5))
    assert abs(b.time_limit() - (w.compute_ops / 2.5e15 / 0.5 + 2e-6)) < 1e-12      # compute bound
    b = Bounds(2.5e15, 8e12, 2e-6, Work(10, 1 << 30), Thresholds(0.5, 0.8))
    assert abs(b.time_limit() - ((1 << 30) / 8e12 / 0.8 + 2e-6)) < 1e-12             # memory bound
    assert Bounds(1.0, 1.0, 0.0, w, Thresholds(0.0, float("nan"))).time_limit() is None


# ---- the GEMM dispatcher's last-round planner (pure host logic inside libmi355cube.so; no device needed) -----------------
def _tail_plan(m, n, k, dtype=None, batch=1, **kw):
    import ctypes as C
    from cubecl_amd import _native as N
    lib = N.load()
    fields = dict(m=m, n=n, k=k, batch=batch, lda=k, ldb=k, ldc=n, dtype_ab=N.DTYPE_BF16 if dtype is None else dtype,
                  dtype_c=N.DTYPE_BF16, trans_b=1)
    fields.update(kw)
    d = N.GemmDesc(**fields)
    along, extent, splits = C.c_int32(), C.c_int64(), C.c_int32()
    assert lib.mi355_gemm_tail_plan(C.byref(d), C.byref(along), C.byref(extent), C.byref(splits)) == N.OK
    return along.value, extent.value, splits.value


def test_tail_plan_splits_only_small_leftover_rounds():
    from cubecl_amd import _native as N
    # whole rounds, or a leftover of half a round and more: one plain launch
    for shape in ((8192, 8192, 8192), (4096, 4096, 4096), (4096, 6144, 4096), (5120, 5120, 5120), (9216, 8192, 4096)):
        assert _tail_plan(*shape)[2] == 1, shape
    # 576 tiles = 2.25 rounds: a strip of tile rows, K split so that strip tiles x splits <= 256
    along, extent, splits = _tail_plan(6144, 6144, 6144)
    strip_tiles = (6144 - extent) // 256 * 24
    assert splits > 1 and along == 1 and extent % 256 == 0 and 0 < strip_tiles <= 96 and strip_tiles * splits <= 256
    assert (6144 // 64) % splits == 0                                  # whole K-tiles per slice
    # 20 x 13 tiles (ragged M, 4 tiles more than one round): the plain launch keeps at most one round, the strip the rest
    along, extent, splits = _tail_plan(5000, 3328, 2048)
    tiles_main = (extent // 256) * (13 if along else 20)
    assert splits > 1 and extent % 256 == 0 and 0 < tiles_main <= 256 and (20 * 13 - tiles_main) * splits <= 256
    assert (2048 // 64) % splits == 0
    # fp8 counts K-tiles of 128, f32 of 32; batches and non-K-contiguous operands are never split
    assert _tail_plan(4608, 4096, 8192, dtype=N.DTYPE_F8E4M3)[2] > 1
    assert _tail_plan(4608, 4096, 8192, batch=2)[2] == 1
    assert _tail_plan(6144, 6144, 6144, trans_b=0)[2] == 1
    assert _tail_plan(6144, 6144, 256)[2] == 1                          # too few K-tiles to split


def _copy_plan(shape, strides, es, out_shape=None, out_strides=None, in_ptr=0x1000, out_ptr=0x2000):
    import ctypes as C
    from cubecl_amd import _native as N
    from oracle.layout import contiguous_strides as cs
    lib = N.load()
    out_shape = list(shape if out_shape is None else out_shape)
    out_strides = cs(out_shape) if out_strides is None else out_strides
    li, lo = N.TensorLayout.of(shape, strides), N.TensorLayout.of(out_shape, out_strides)
    path, access = C.c_int32(-1), C.c_int32(-1)
    rc = lib.mi355_copy_strided_plan(C.c_void_p(in_ptr), C.byref(li), C.c_void_p(out_ptr), C.byref(lo), es, C.byref(path), C.byref(access))
    return rc, path.value, access.value


def test_copy_strided_plan_picks_the_cheapest_mover():
    """Host half of copy_into (cubecl_amd/csrc/copy_strided.hip): the two views are refined to a joint space, unit axes
    dropped, adjacent axes merged, and the mover follows from what is left.  No device involved."""
    from cubecl_amd import _native as N
    OK = N.OK
    # contiguous views of any rank collapse to one run; so does a reshape
    assert _copy_plan([4, 6, 64], [384, 64, 1], 4) == (OK, N.COPY_PATH_FLAT, 16)
    assert _copy_plan([4, 6, 64], [384, 64, 1], 4, out_shape=[24, 64]) == (OK, N.COPY_PATH_FLAT, 16)
    assert _copy_plan([1, 2, 4, 1], [8, 4, 1, 1], 4, out_shape=[1, 2, 4]) == (OK, N.COPY_PATH_FLAT, 16)
    assert _copy_plan([7], [1], 2) == (OK, N.COPY_PATH_FLAT, 2)                       # 14 bytes: 2-byte accesses
    assert _copy_plan([8], [1], 2, in_ptr=0x1004) == (OK, N.COPY_PATH_FLAT, 4)        # the pointer caps the width
    # padded rows on either side: rows
    assert _copy_plan([30, 64], [80, 1], 4) == (OK, N.COPY_PATH_ROWS, 16)
    assert _copy_plan([30, 64], [64, 1], 4, out_strides=[66, 1]) == (OK, N.COPY_PATH_ROWS, 8)
    assert _copy_plan([30, 63], [63, 1], 4, out_strides=[64, 1]) == (OK, N.COPY_PATH_ROWS, 4)
    assert _copy_plan([5, 7, 64], [64, 320, 1], 2) == (OK, N.COPY_PATH_ROWS, 16)     # outer axes swapped
    # the input contiguous along another axis than the output: tiles, vectorised when every stride allows
    assert _copy_plan([64, 128], [1, 64], 2) == (OK, N.COPY_PATH_TRANSPOSE, 16)
    assert _copy_plan([64, 104], [1, 64], 2)[1:] == (N.COPY_PATH_TRANSPOSE, 16)
    assert _copy_plan([64, 100], [1, 64], 2)[1:] == (N.COPY_PATH_TRANSPOSE, 2)          # output rows of 200 bytes
    assert _copy_plan([3, 64, 128], [8192, 1, 64], 1)[1:] == (N.COPY_PATH_TRANSPOSE, 16)
    assert _copy_plan([64, 128], [128, 1], 4, out_strides=[1, 64])[1:] == (N.COPY_PATH_TRANSPOSE, 16)
    # NCHW -> NHWC: H and W merge into one axis of 40 * 56
    assert _copy_plan([4, 40, 56, 48], [107520, 56, 1, 2240], 4)[1:] == (N.COPY_PATH_TRANSPOSE, 16)
    # too short to tile, 8-byte elements, strided gathers: generic; when one side is contiguous along the innermost joint
    # axis a thread packs as many elements as divide it into one 16-byte access on that side
    # (8 bytes when a row holds several such groups -- wider scatters the lanes of the element-wise side --, 16 when the
    # row is one group and the lanes walk the other side's contiguous axis)
    assert _copy_plan([8, 128], [1, 8], 4)[1:] == (N.COPY_PATH_GENERIC, 8)
    assert _copy_plan([64, 128], [1, 64], 8)[1:] == (N.COPY_PATH_TRANSPOSE, 16)
    assert _copy_plan([8, 128], [1, 8], 8)[1:] == (N.COPY_PATH_GENERIC, 8)
    assert _copy_plan([100], [3], 4)[1:] == (N.COPY_PATH_GENERIC, 8)
    assert _copy_plan([100], [3], 1)[1:] == (N.COPY_PATH_GENERIC, 4)
    assert _copy_plan([102], [3], 4)[1:] == (N.COPY_PATH_GENERIC, 8)
    assert _copy_plan([101], [3], 4)[1:] == (N.COPY_PATH_GENERIC, 4)
    assert _copy_plan([1000, 8, 8], [64, 1, 8], 2)[1:] == (N.COPY_PATH_GENERIC, 16)           # 8 x 8 transposes: K = the row
    assert _copy_plan([100], [1], 2, out_strides=[5])[1:] == (N.COPY_PATH_GENERIC, 8)       # scatter: vector loads
    assert _copy_plan([100], [3], 4, out_strides=[5])[1:] == (N.COPY_PATH_GENERIC, 4)       # neither side contiguous
    # the reference's rank-mismatch case (tests/tensor/into_contiguous.rs:143-146) refines to [2, 4]: generic
    assert _copy_plan([1, 2, 4, 1], [8, 1, 2, 2], 4, out_shape=[1, 2, 4])[1:] == (N.COPY_PATH_GENERIC, 16)
    # a contiguous output refines against anything; two strided views whose axis boundaries do not nest have no common
    # refinement
    assert _copy_plan([2, 3], [1, 2], 4, out_shape=[3, 2])[1:] == (N.COPY_PATH_GENERIC, 4)
    assert _copy_plan([2, 3], [1, 2], 4, out_shape=[3, 2], out_strides=[4, 1])[1:] == (N.COPY_PATH_TWO_SIDED, 4)
    # contiguous [2, 3] -> [3, 2] needs none: both collapse
    assert _copy_plan([2, 3], [3, 1], 4, out_shape=[3, 2])[1:] == (N.COPY_PATH_FLAT, 8)
    # rejected: element counts differ, broadcasting output, bad element size, rank 9, negative stride
    E = N.E_INVALID_ARGUMENT
    assert _copy_plan([4, 8], [8, 1], 4, out_shape=[4, 7])[0] == E
    assert _copy_plan([4, 8], [8, 1], 4, out_strides=[0, 1])[0] == E
    assert _copy_plan([4, 8], [8, 1], 3)[0] == E
    assert _copy_plan([4, 8], [8, -1], 4)[0] == E
    lay = N.TensorLayout.of([4], [1])
    lay.rank = 9
    import ctypes as C
    assert N.load().mi355_copy_strided_plan(None, C.byref(lay), None, C.byref(lay), 4, None, None) == E


def test_contiguous_pitched_predicate_and_permute():
    from cubecl_amd.runtime import Handle, _Memory
    h = Handle(_Memory(None, 0, 0, None), None, None, 0)
    t = TensorHandle.new(h, (2, 3, 4), (16, 4, 1), ElemType.F32)            # pitched rows? no: 3 * 4 != 16
    assert not t.is_contiguous() and not t.is_contiguous_pitched()
    assert TensorHandle.new(h, (2, 3, 4), (24, 8, 1), ElemType.F32).is_contiguous_pitched()
    assert TensorHandle.new(h, (2, 3, 4), (12, 4, 1), ElemType.F32).is_contiguous_pitched()
    assert not TensorHandle.new(h, (2, 3, 4), (4, 8, 1), ElemType.F32).is_contiguous_pitched()
    assert not TensorHandle.new(h, (3, 4), (1, 3), ElemType.F32).is_contiguous_pitched()
    p = TensorHandle.new(h, (2, 3, 4), (12, 4, 1), ElemType.F32).permute([2, 0, 1])
    assert p.shape == (4, 2, 3) and p.strides == (1, 12, 4)
    with pytest.raises(ValueError):
        p.permute([0, 0, 1])
