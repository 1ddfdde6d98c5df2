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
