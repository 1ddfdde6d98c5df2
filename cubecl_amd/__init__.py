"""cubecl_amd -- MI355X (gfx950) native backend for the tiled-matmul / reduction hot path of
tracel-ai/cubecl: hand-written HIP kernels behind a C ABI (include/mi355cube.h) and a host-side
mirror of the reference's Runtime / ComputeClient / TensorHandle surface."""
from . import _native
from .runtime import (ComputeClient, CopyDescriptor, CubeCount, CubeCountSelection, CubeDim, cube_count_spread, DeviceId, ElemType, Handle, MemoryLayout,
                      Mi355Runtime, ReduceOperation, ServerError, contiguous_strides, has_pitched_row_major_strides)
from .tensor import MatrixBatchLayout, TensorHandle, matrix_batch_layout
from . import ops
from .info import AddressType, InfoBuilder, KernelArguments, MetadataBindingInfo

__all__ = ["ComputeClient", "CopyDescriptor", "CubeCount", "CubeCountSelection", "CubeDim", "cube_count_spread", "DeviceId", "ElemType", "Handle",
           "MemoryLayout", "Mi355Runtime", "ReduceOperation", "ServerError", "TensorHandle", "MatrixBatchLayout",
           "matrix_batch_layout", "AddressType", "InfoBuilder", "KernelArguments", "MetadataBindingInfo", "contiguous_strides", "has_pitched_row_major_strides", "ops", "_native"]
