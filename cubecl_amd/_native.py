"""ctypes binding of libmi355cube.so (include/mi355cube.h).

The product path has NO fallback: if the shared library is missing or does not load, every
entry point raises -- nothing here ever routes through the CPU oracle or PyTorch eager.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = _CSRC / "libmi355cube.so"

OK = 0
E_INVALID_ARGUMENT = 1
E_OUT_OF_MEMORY = 2
E_BUFFER_TOO_BIG = 3
E_UNSUPPORTED_STRIDES = 4
E_NOT_FOUND = 5
E_SHARED_MEMORY = 6
E_UNITS = 7
E_CUBE_DIM = 8
E_MAX_UNITS_PER_CUBE = 9
E_COMPILATION = 10
E_LAUNCH = 11
E_EXECUTION = 12
E_UNSUPPORTED = 13
E_SERVER_UNHEALTHY = 14
E_COMM = 15
E_NO_DEVICE = 16
E_PROFILE = 17

ERROR_NAMES = {
    0: "Ok", 1: "Validation", 2: "OutOfMemory", 3: "BufferTooBig", 4: "UnsupportedStrides", 5: "NotFound",
    6: "TooManyResources(SharedMemory)", 7: "TooManyResources(Units)", 8: "TooManyResources(CubeDim)",
    9: "TooManyResources(MaxUnitPerCube)", 10: "CompilationError", 11: "LaunchError", 12: "Execution",
    13: "Unsupported", 14: "ServerUnhealthy", 15: "Communication", 16: "NoDevice", 17: "ProfileError",
}

DTYPE_F32, DTYPE_BF16, DTYPE_F16, DTYPE_F64, DTYPE_I32, DTYPE_U32, DTYPE_I64, DTYPE_U64, DTYPE_U8, DTYPE_I8 = range(10)
DTYPE_SIZE = {DTYPE_F32: 4, DTYPE_BF16: 2, DTYPE_F16: 2, DTYPE_F64: 8, DTYPE_I32: 4, DTYPE_U32: 4, DTYPE_I64: 8,
              DTYPE_U64: 8, DTYPE_U8: 1, DTYPE_I8: 1}
DTYPE_F8E4M3, DTYPE_F8E5M2 = 10, 11            # OCP FP8 (fp8_e4m3.rs / fp8_e5m2.rs of the reference)
DTYPE_F4E2M1X2, DTYPE_UE8M0 = 12, 13            # packed e2m1 pairs (one byte per pair), MX block scales
DTYPE_SIZE.update({DTYPE_F8E4M3: 1, DTYPE_F8E5M2: 1, DTYPE_F4E2M1X2: 1, DTYPE_UE8M0: 1})
# advertised for generated kernels only (register_supported_types, crates/cubecl-cpp/src/shared/base.rs:322-375)
DTYPE_I16, DTYPE_U16, DTYPE_BOOL, DTYPE_FLEX32, DTYPE_INDEX = 14, 15, 16, 17, 18
DTYPE_SIZE.update({DTYPE_I16: 2, DTYPE_U16: 2, DTYPE_BOOL: 1, DTYPE_FLEX32: 4})
TYPE_USAGE = {"Conversion": 1, "Arithmetic": 2, "DotProduct": 4, "Buffer": 8}                     # features.rs:79-88
ATOMIC_USAGE = {"LoadStore": 1, "Exchange": 2, "Add": 4, "MinMax": 8, "Bitwise": 16, "CompareExchange": 32}   # :110-123
ADDRESS_TYPE_U32, ADDRESS_TYPE_U64 = 1, 2
LAYOUT_ROW_MAJOR, LAYOUT_COL_MAJOR = 0, 1

REDUCE_SUM, REDUCE_MEAN, REDUCE_MAX, REDUCE_MIN, REDUCE_PROD, REDUCE_ARGMAX, REDUCE_ARGMIN = 0, 1, 2, 3, 4, 5, 6
PLANE_PROD, PLANE_INCLUSIVE_SUM, PLANE_EXCLUSIVE_SUM, PLANE_INCLUSIVE_PROD, PLANE_EXCLUSIVE_PROD = 100, 101, 102, 103, 104
(PLANE_ALL, PLANE_ANY, PLANE_ELECT, PLANE_BROADCAST, PLANE_SHUFFLE, PLANE_SHUFFLE_XOR, PLANE_SHUFFLE_UP, PLANE_SHUFFLE_DOWN,
 PLANE_BALLOT) = range(200, 209)

GEMM_ALGO_AUTO, GEMM_ALGO_GENERIC, GEMM_ALGO_F32_MFMA, GEMM_ALGO_LP_128, GEMM_ALGO_LP_256, GEMM_ALGO_LP_256W4, GEMM_ALGO_LP_256P = 0, 1, 2, 3, 4, 5, 6
GEMM_ALGO_LP_256Q = 7
GEMM_ALGO_SKINNY = 8
GEMM_ALGO_STREAM64 = 9
GEMM_ALGO_LP_256X128 = 10
GEMM_ALGO_NNROWS = 11
GEMM_ALGO_LP_256X192 = 12
GEMM_ALGO_LP_192X192 = 13
GEMM_ALGO_LP_256M16 = 14
GEMM_ALGO_LP_256QM = 15
UNIQUE_ID_BYTES = 128


class MmaConfig(C.Structure):
    _fields_ = [("m", C.c_uint32), ("n", C.c_uint32), ("k", C.c_uint32),
                ("a_type", C.c_int32), ("b_type", C.c_int32), ("cd_type", C.c_int32)]


ABI_VERSION = 9     # MI355_ABI_VERSION of include/mi355cube.h this table was written against


class MemoryUsage(C.Structure):
    """mi355_memory_usage (MemoryUsage of memory_management/base.rs:8-28 + driver-call counters)."""
    _fields_ = [(n, C.c_uint64) for n in ("number_allocs", "bytes_in_use", "bytes_padding", "bytes_reserved",
                                          "driver_allocs", "driver_frees", "cache_hits", "reserved")]


ALLOC_MODE_AUTO, ALLOC_MODE_PERSISTENT = 0, 1


class ScaledMmaConfig(C.Structure):
    _fields_ = [("m", C.c_uint32), ("n", C.c_uint32), ("k", C.c_uint32), ("a_type", C.c_int32), ("b_type", C.c_int32),
                ("cd_type", C.c_int32), ("scales_type", C.c_int32), ("scales_factor", C.c_uint32)]


class TypeUsageEntry(C.Structure):
    """mi355_type_usage: one ElemType with its TypeUsage (or AtomicUsage) bit set."""
    _fields_ = [("dtype", C.c_int32), ("usage", C.c_uint32)]


class MmaProperties(C.Structure):
    """mi355_mma_properties: TargetProperties.mma for MFMA (crates/cubecl-ir/src/runtime_properties.rs:19-39)."""
    _fields_ = [(n, C.c_uint32) for n in ("register_size_bits", "const_plane_size", "register_layout_a", "register_layout_b",
                                          "register_layout_acc", "register_duplication_a", "register_duplication_b",
                                          "register_duplication_acc", "contiguous_elements_ab_bits", "contiguous_elements_acc")]


class DeviceProps(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device_index", C.c_int32),
        ("name", C.c_char * 64), ("gcn_arch_name", C.c_char * 64), ("fingerprint", C.c_char * 96),
        ("load_width_bits", C.c_uint32), ("plane_size_min", C.c_uint32), ("plane_size_max", C.c_uint32),
        ("max_bindings", C.c_uint32), ("max_shared_memory_size", C.c_uint64),
        ("max_cube_count", C.c_uint32 * 3), ("max_units_per_cube", C.c_uint32), ("max_cube_dim", C.c_uint32 * 3),
        ("num_streaming_multiprocessors", C.c_uint32), ("num_tensor_cores", C.c_uint32),
        ("min_tensor_cores_dim", C.c_uint32), ("num_xcd", C.c_uint32),
        ("total_memory", C.c_uint64), ("max_page_size", C.c_uint64), ("mem_alignment", C.c_uint64),
        ("clock_khz", C.c_uint32), ("memory_clock_khz", C.c_uint32), ("memory_bus_width_bits", C.c_uint32),
        ("l2_cache_bytes", C.c_uint32), ("plane_ops", C.c_uint32), ("plane_non_uniform", C.c_uint32),
        ("timing_method_device", C.c_uint32), ("server_comm_enabled", C.c_uint32),
        ("num_mma_configs", C.c_uint32), ("mma_configs", MmaConfig * 16),
        ("num_scaled_mma_configs", C.c_uint32), ("scaled_mma_configs", ScaledMmaConfig * 8),
        ("address_types", C.c_uint32), ("num_type_usage", C.c_uint32), ("type_usage", TypeUsageEntry * 24),
        ("num_atomic_usage", C.c_uint32), ("atomic_usage", TypeUsageEntry * 8), ("mma_properties", MmaProperties),
    ]


class GemmDesc(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("k", C.c_int64), ("batch", C.c_int64),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
        ("stride_a", C.c_int64), ("stride_b", C.c_int64), ("stride_c", C.c_int64),
        ("dtype_ab", C.c_int32), ("dtype_c", C.c_int32), ("trans_a", C.c_int32), ("trans_b", C.c_int32),
        ("algo", C.c_int32), ("reserved", C.c_int32),
    ]


class GemmScaledDesc(C.Structure):
    """mi355_gemm_scaled_desc."""
    _fields_ = [(n, C.c_int64) for n in ("m", "n", "k", "batch", "lda", "ldb", "ldc", "ld_sa", "ld_sb", "stride_a", "stride_b",
                                         "stride_c", "stride_sa", "stride_sb")] + \
               [(n, C.c_int32) for n in ("dtype_a", "dtype_b", "dtype_c", "block", "algo", "reserved")]


MAX_RANK = 8
COPY_PATH_FLAT, COPY_PATH_ROWS, COPY_PATH_TRANSPOSE, COPY_PATH_GENERIC, COPY_PATH_TWO_SIDED = range(5)


class TensorLayout(C.Structure):
    """mi355_tensor_layout: shape / strides in elements, outermost axis first."""
    _fields_ = [("rank", C.c_int32), ("reserved", C.c_int32), ("shape", C.c_int64 * MAX_RANK), ("strides", C.c_int64 * MAX_RANK)]

    @staticmethod
    def of(shape, strides) -> "TensorLayout":
        shape, strides = list(shape) or [1], list(strides) or [1]
        if len(shape) != len(strides) or len(shape) > MAX_RANK:
            raise ValueError(f"layout of rank {len(shape)} / {len(strides)} (at most {MAX_RANK} axes)")
        lay = TensorLayout()
        lay.rank = len(shape)
        for i, (a, b) in enumerate(zip(shape, strides)):
            lay.shape[i], lay.strides[i] = int(a), int(b)
        return lay


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_U64P = C.POINTER(C.c_uint64)
_I32P = C.POINTER(C.c_int32)
_U32x3 = C.POINTER(C.c_uint32)

# name -> (restype, argtypes); every symbol include/mi355cube.h declares
PROTOTYPES = {
    "mi355_abi_version": (C.c_int32, []),
    "mi355_device_count": (C.c_int32, [_I32P]),
    "mi355_ctx_create": (C.c_int32, [C.c_int32, _PP]),
    "mi355_ctx_destroy": (C.c_int32, [_P]),
    "mi355_device_props": (C.c_int32, [_P, C.POINTER(DeviceProps)]),
    "mi355_last_error": (C.c_char_p, [_P]),
    "mi355_last_global_error": (C.c_char_p, []),
    "mi355_error_count": (C.c_int32, [_P, _I32P]),
    "mi355_error_pop": (C.c_int32, [_P, _I32P, _U64P, _U64P, C.c_char_p, C.c_size_t]),
    "mi355_alloc": (C.c_int32, [_P, C.c_uint64, _PP]),
    "mi355_free": (C.c_int32, [_P, _P]),
    "mi355_mem_info": (C.c_int32, [_P, _U64P, _U64P]),
    "mi355_copy_to_ctx": (C.c_int32, [_P, _P, _P, _P, _P, _P, C.c_uint64]),
    "mi355_pool_alloc": (C.c_int32, [_P, _P, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mi355_pool_free": (C.c_int32, [_P, _P, _P]),
    "mi355_pool_cleanup": (C.c_int32, [_P, C.c_int32]),
    "mi355_pool_mode": (C.c_int32, [_P, C.c_int32]),
    "mi355_pool_usage": (C.c_int32, [_P, C.POINTER(MemoryUsage)]),
    "mi355_pitched_row_bytes": (C.c_int32, [_P, C.c_uint64, _U64P]),
    "mi355_pinned_alloc": (C.c_int32, [_P, C.c_uint64, _PP]),
    "mi355_pinned_free": (C.c_int32, [_P, _P]),
    "mi355_stream_create": (C.c_int32, [_P, _PP]),
    "mi355_stream_destroy": (C.c_int32, [_P, _P]),
    "mi355_default_stream": (C.c_int32, [_P, _PP]),
    "mi355_comm_stream": (C.c_int32, [_P, _PP]),
    "mi355_event_create": (C.c_int32, [_P, _PP]),
    "mi355_event_destroy": (C.c_int32, [_P, _P]),
    "mi355_event_record": (C.c_int32, [_P, _P, _P]),
    "mi355_stream_wait_event": (C.c_int32, [_P, _P, _P]),
    "mi355_event_sync": (C.c_int32, [_P, _P]),
    "mi355_event_elapsed_ms": (C.c_int32, [_P, _P, _P, C.POINTER(C.c_float)]),
    "mi355_write": (C.c_int32, [_P, _P, _P, _P, C.c_uint64]),
    "mi355_read": (C.c_int32, [_P, _P, _P, _P, C.c_uint64]),
    "mi355_read_async": (C.c_int32, [_P, _P, _P, _P, C.c_uint64]),
    "mi355_write_2d": (C.c_int32, [_P, _P, _P, C.c_uint64, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_read_2d": (C.c_int32, [_P, _P, _P, C.c_uint64, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_copy_d2d": (C.c_int32, [_P, _P, _P, _P, C.c_uint64]),
    "mi355_memset": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_uint64]),
    "mi355_sync": (C.c_int32, [_P, _P]),
    "mi355_flush": (C.c_int32, [_P]),
    "mi355_module_load": (C.c_int32, [_P, _P, C.c_size_t, _PP]),
    "mi355_module_unload": (C.c_int32, [_P, _P]),
    "mi355_module_get_function": (C.c_int32, [_P, _P, C.c_char_p, _PP]),
    "mi355_launch": (C.c_int32, [_P, _P, _P, _U32x3, _U32x3, C.c_uint32, _PP, C.c_uint32]),
    "mi355_fill_uniform": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float]),
    "mi355_cast": (C.c_int32, [_P, _P, _P, C.c_int32, _P, C.c_int32, C.c_uint64]),
    "mi355_gemm": (C.c_int32, [_P, _P, C.POINTER(GemmDesc), _P, _P, _P]),
    "mi355_gemm_add": (C.c_int32, [_P, _P, C.POINTER(GemmDesc), _P, _P, _P, _P]),
    "mi355_gemm_select": (C.c_int32, [_P, C.POINTER(GemmDesc), _I32P]),
    "mi355_gemm_tail_plan": (C.c_int32, [C.POINTER(GemmDesc), _I32P, C.POINTER(C.c_int64), _I32P]),
    "mi355_gemm_split_plan": (C.c_int32, [C.POINTER(GemmDesc), C.c_int32, _I32P]),
    "mi355_gemm_strip_plan": (C.c_int32, [C.POINTER(GemmDesc), C.c_int32, _I32P, _I32P, _I32P]),
    "mi355_gemm_relayout_plan": (C.c_int32, [C.POINTER(GemmDesc), _I32P, _I32P]),
    "mi355_gemm_scaled": (C.c_int32, [_P, _P, C.POINTER(GemmScaledDesc), _P, _P, _P, _P, _P]),
    "mi355_gemm_scaled_select": (C.c_int32, [_P, C.POINTER(GemmScaledDesc), _I32P]),
    "mi355_fill_identity": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_uint64, C.c_uint64]),
    "mi355_copy_strided": (C.c_int32, [_P, _P, _P, C.POINTER(TensorLayout), _P, C.POINTER(TensorLayout), C.c_int32]),
    "mi355_copy_strided_plan": (C.c_int32, [_P, C.POINTER(TensorLayout), _P, C.POINTER(TensorLayout), C.c_int32, _I32P, _I32P]),
    "mi355_copy_packed": (C.c_int32, [_P, _P, _P, C.POINTER(TensorLayout), _P, C.POINTER(TensorLayout), C.POINTER(C.c_int64),
                                      C.c_int32, C.c_int32, C.c_int32]),
    "mi355_reduce_workspace_bytes": (C.c_int32, [_P, C.c_uint64, _U64P]),
    "mi355_reduce_sum_f32": (C.c_int32, [_P, _P, _P, C.c_uint64, _P, _P, C.c_uint64]),
    "mi355_argmax_f32": (C.c_int32, [_P, _P, _P, C.c_uint64, _P, _P, _P, C.c_uint64]),
    "mi355_sum_argmax_f32": (C.c_int32, [_P, _P, _P, C.c_uint64, _P, _P, _P, _P, C.c_uint64]),
    "mi355_argmax_combine_f32": (C.c_int32, [_P, _P, _P, C.c_uint32, C.POINTER(C.c_uint64), _P, _P]),
    "mi355_sum_argmax_combine_f32": (C.c_int32, [_P, _P, _P, C.c_uint32, C.POINTER(C.c_uint64), _P, _P, _P]),
    "mi355_sum_argmax_exchange": (C.c_int32, [_P, _P, _P, _P, _P, C.POINTER(C.c_uint64), _P, _P, _P]),
    "mi355_reduce": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_uint64, C.c_int32, _P, _P, C.c_uint64]),
    "mi355_argreduce": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_uint64, C.c_int32, _P, _P, _P, C.c_uint64]),
    "mi355_reduce_axis": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_argreduce_axis": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_reduce_axis_sum": (C.c_int32, [_P, _P, _P, C.c_int32, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_reduce_axis_argmax": (C.c_int32, [_P, _P, _P, C.c_int32, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_reduce_last_axis_sum": (C.c_int32, [_P, _P, _P, C.c_int32, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_reduce_last_axis_argmax": (C.c_int32, [_P, _P, _P, C.c_int32, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_reduce_sum": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_uint64, _P, _P, C.c_uint64]),
    "mi355_argmax": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_uint64, _P, _P, _P, C.c_uint64]),
    "mi355_sum_argmax": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_uint64, _P, _P, _P, _P, C.c_uint64]),
    "mi355_reduce_last_axis_sum_f32": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_reduce_last_axis_argmax_f32": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_reduce_axis_sum_f32": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_reduce_axis_argmax_f32": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "mi355_plane_reduce_f32": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, C.c_uint32, C.c_int32]),
    "mi355_plane_op_f32": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, C.c_uint32, C.c_int32, C.c_uint32]),
    "mi355_probe_memory_read": (C.c_int32, [_P, _P, _P, C.c_uint64, C.c_uint32, _P]),
    "mi355_probe_mfma": (C.c_int32, [_P, _P, C.c_int32, C.c_uint32, _P, _U64P]),
    "mi355_probe_mfma_data": (C.c_int32, [_P, _P, C.c_int32, C.c_uint32, _P, _U64P]),
    "mi355_probe_clock": (C.c_int32, [_P, _P, _P]),
    "mi355_probe_memory_copy": (C.c_int32, [_P, _P, _P, _P, C.c_uint64]),
    "mi355_probe_memory_write": (C.c_int32, [_P, _P, _P, C.c_uint64]),
    "mi355_probe_compute_direct": (C.c_int32, [_P, _P, C.c_uint32, _P, _U64P]),
    "mi355_probe_launch_overhead": (C.c_int32, [_P, _P, C.c_uint32, _P]),
    "mi355_comm_unique_id": (C.c_int32, [C.POINTER(C.c_uint8)]),
    "mi355_comm_init": (C.c_int32, [_P, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, _PP]),
    "mi355_comm_destroy": (C.c_int32, [_P, _P]),
    "mi355_all_reduce": (C.c_int32, [_P, _P, _P, _P, _P, C.c_uint64, C.c_int32, C.c_int32]),
    "mi355_all_gather": (C.c_int32, [_P, _P, _P, _P, _P, C.c_uint64, C.c_int32]),
    "mi355_send": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, C.c_int32, C.c_int32]),
    "mi355_recv": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, C.c_int32, C.c_int32]),
    "mi355_sync_collective": (C.c_int32, [_P, _P]),
    "mi355_graph_begin_capture": (C.c_int32, [_P, _P]),
    "mi355_graph_end_capture": (C.c_int32, [_P, _P, _PP]),
    "mi355_graph_replay": (C.c_int32, [_P, _P, _P]),
    "mi355_graph_destroy": (C.c_int32, [_P, _P]),
    "mi355_profile_start": (C.c_int32, [_P, _P, _U64P]),
    "mi355_profile_stop": (C.c_int32, [_P, _P, C.c_uint64, _U64P]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    """libmi355cube.so is missing / unloadable: the product path refuses to run."""


def load() -> C.CDLL:
    """Load libmi355cube.so (once) and type every entry point.  Raises NativeLibraryError loudly."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("MI355CUBE_LIB", str(LIB_PATH)))
    if not path.exists():
        raise NativeLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cubecl_amd/csrc`). There is no CPU/PyTorch fallback for this path.")
    try:
        lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise NativeLibraryError(f"cannot load {path}: {exc}") from exc
    missing = []
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = restype
        fn.argtypes = argtypes
    if missing:
        raise NativeLibraryError(f"{path} does not export: {', '.join(missing)}")
    if lib.mi355_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"{path}: ABI version {lib.mi355_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def header_symbols() -> list[str]:
    """Function names declared in include/mi355cube.h (used by the CPU-side export test)."""
    import re
    text = (Path(__file__).resolve().parents[1] / "include" / "mi355cube.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))
