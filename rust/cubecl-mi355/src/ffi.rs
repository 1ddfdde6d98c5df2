//! Raw bindings of `include/mi355cube.h` (ABI version `MI355_ABI_VERSION`).  One declaration per C entry point; the
//! trait each one serves is named in the header next to its prototype.
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_void};

/// `MI355_ABI_VERSION` of the header these declarations were written against; `Mi355Runtime` compares it with what the
/// loaded library reports (`mi355_device_props_t::abi_version`).
pub const MI355_ABI_VERSION: u32 = 9;

pub const MI355_OK: i32 = 0;
pub const MI355_E_INVALID_ARGUMENT: i32 = 1;
pub const MI355_E_OUT_OF_MEMORY: i32 = 2;
pub const MI355_E_BUFFER_TOO_BIG: i32 = 3;
pub const MI355_E_UNSUPPORTED_STRIDES: i32 = 4;
pub const MI355_E_NOT_FOUND: i32 = 5;
pub const MI355_E_SHARED_MEMORY: i32 = 6;
pub const MI355_E_UNITS: i32 = 7;
pub const MI355_E_CUBE_DIM: i32 = 8;
pub const MI355_E_MAX_UNITS_PER_CUBE: i32 = 9;
pub const MI355_E_COMPILATION: i32 = 10;
pub const MI355_E_LAUNCH: i32 = 11;
pub const MI355_E_EXECUTION: i32 = 12;
pub const MI355_E_UNSUPPORTED: i32 = 13;
pub const MI355_E_SERVER_UNHEALTHY: i32 = 14;
pub const MI355_E_COMM: i32 = 15;
pub const MI355_E_NO_DEVICE: i32 = 16;
pub const MI355_E_PROFILE: i32 = 17;

pub const MI355_DTYPE_F32: i32 = 0;
pub const MI355_DTYPE_BF16: i32 = 1;
pub const MI355_DTYPE_F16: i32 = 2;
pub const MI355_DTYPE_F64: i32 = 3;
pub const MI355_DTYPE_I32: i32 = 4;
pub const MI355_DTYPE_U32: i32 = 5;
pub const MI355_DTYPE_I64: i32 = 6;
pub const MI355_DTYPE_U64: i32 = 7;
pub const MI355_DTYPE_U8: i32 = 8;
pub const MI355_DTYPE_I8: i32 = 9;
pub const MI355_DTYPE_F8E4M3: i32 = 10;
pub const MI355_DTYPE_F8E5M2: i32 = 11;
pub const MI355_DTYPE_F4E2M1X2: i32 = 12;
pub const MI355_DTYPE_UE8M0: i32 = 13;
pub const MI355_DTYPE_I16: i32 = 14;
pub const MI355_DTYPE_U16: i32 = 15;
pub const MI355_DTYPE_BOOL: i32 = 16;
pub const MI355_DTYPE_FLEX32: i32 = 17;
pub const MI355_DTYPE_INDEX: i32 = 18;
pub const MI355_TYPE_USAGE_CONVERSION: u32 = 1;
pub const MI355_TYPE_USAGE_ARITHMETIC: u32 = 2;
pub const MI355_TYPE_USAGE_DOT_PRODUCT: u32 = 4;
pub const MI355_TYPE_USAGE_BUFFER: u32 = 8;
pub const MI355_ATOMIC_LOAD_STORE: u32 = 1;
pub const MI355_ATOMIC_EXCHANGE: u32 = 2;
pub const MI355_ATOMIC_ADD: u32 = 4;
pub const MI355_ATOMIC_MIN_MAX: u32 = 8;
pub const MI355_ATOMIC_BITWISE: u32 = 16;
pub const MI355_ATOMIC_COMPARE_EXCHANGE: u32 = 32;
pub const MI355_ADDRESS_TYPE_U32: u32 = 1;
pub const MI355_ADDRESS_TYPE_U64: u32 = 2;
pub const MI355_LAYOUT_ROW_MAJOR: u32 = 0;
pub const MI355_LAYOUT_COL_MAJOR: u32 = 1;
pub const MI355_ALLOC_MODE_AUTO: i32 = 0;
pub const MI355_ALLOC_MODE_PERSISTENT: i32 = 1;

pub const MI355_REDUCE_SUM: i32 = 0;
pub const MI355_REDUCE_MEAN: i32 = 1;
pub const MI355_REDUCE_MAX: i32 = 2;
pub const MI355_REDUCE_MIN: i32 = 3;
pub const MI355_REDUCE_PROD: i32 = 4;
pub const MI355_REDUCE_ARGMAX: i32 = 5;
pub const MI355_REDUCE_ARGMIN: i32 = 6;
pub const MI355_UNIQUE_ID_BYTES: usize = 128;

#[repr(C)]
pub struct mi355_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mi355_comm {
    _private: [u8; 0],
}
pub type mi355_stream = *mut c_void;
pub type mi355_event = *mut c_void;
pub type mi355_module = *mut c_void;
pub type mi355_function = *mut c_void;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355_mma_config {
    pub m: u32,
    pub n: u32,
    pub k: u32,
    pub a_type: i32,
    pub b_type: i32,
    pub cd_type: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_device_props_t {
    pub abi_version: u32,
    pub device_index: i32,
    pub name: [c_char; 64],
    pub gcn_arch_name: [c_char; 64],
    pub fingerprint: [c_char; 96],
    pub load_width_bits: u32,
    pub plane_size_min: u32,
    pub plane_size_max: u32,
    pub max_bindings: u32,
    pub max_shared_memory_size: u64,
    pub max_cube_count: [u32; 3],
    pub max_units_per_cube: u32,
    pub max_cube_dim: [u32; 3],
    pub num_streaming_multiprocessors: u32,
    pub num_tensor_cores: u32,
    pub min_tensor_cores_dim: u32,
    pub num_xcd: u32,
    pub total_memory: u64,
    pub max_page_size: u64,
    pub mem_alignment: u64,
    pub clock_khz: u32,
    pub memory_clock_khz: u32,
    pub memory_bus_width_bits: u32,
    pub l2_cache_bytes: u32,
    pub plane_ops: u32,
    pub plane_non_uniform: u32,
    pub timing_method_device: u32,
    pub server_comm_enabled: u32,
    pub num_mma_configs: u32,
    pub mma_configs: [mi355_mma_config; 16],
    pub num_scaled_mma_configs: u32,
    pub scaled_mma_configs: [mi355_scaled_mma_config; 8],
    pub address_types: u32,
    pub num_type_usage: u32,
    pub type_usage: [mi355_type_usage; 24],
    pub num_atomic_usage: u32,
    pub atomic_usage: [mi355_type_usage; 8],
    pub mma_properties: mi355_mma_properties,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355_type_usage {
    pub dtype: i32,
    pub usage: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355_mma_properties {
    pub register_size_bits: u32,
    pub const_plane_size: u32,
    pub register_layout_a: u32,
    pub register_layout_b: u32,
    pub register_layout_acc: u32,
    pub register_duplication_a: u32,
    pub register_duplication_b: u32,
    pub register_duplication_acc: u32,
    pub contiguous_elements_ab_bits: u32,
    pub contiguous_elements_acc: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355_scaled_mma_config {
    pub m: u32,
    pub n: u32,
    pub k: u32,
    pub a_type: i32,
    pub b_type: i32,
    pub cd_type: i32,
    pub scales_type: i32,
    pub scales_factor: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355_gemm_desc {
    pub m: i64,
    pub n: i64,
    pub k: i64,
    pub batch: i64,
    pub lda: i64,
    pub ldb: i64,
    pub ldc: i64,
    pub stride_a: i64,
    pub stride_b: i64,
    pub stride_c: i64,
    pub dtype_ab: i32,
    pub dtype_c: i32,
    pub trans_a: i32,
    pub trans_b: i32,
    pub algo: i32,
    pub reserved: i32,
}

/// mi355_tensor_layout: TensorBinding's shape / strides (elements, outermost axis first).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355_tensor_layout {
    pub rank: i32,
    pub reserved: i32,
    pub shape: [i64; 8],
    pub strides: [i64; 8],
}

unsafe extern "C" {
    // Runtime
    pub fn mi355_abi_version() -> i32;
    pub fn mi355_device_count(out_count: *mut i32) -> i32;
    pub fn mi355_ctx_create(device_index: i32, out_ctx: *mut *mut mi355_ctx) -> i32;
    pub fn mi355_ctx_destroy(ctx: *mut mi355_ctx) -> i32;
    pub fn mi355_device_props(ctx: *mut mi355_ctx, out: *mut mi355_device_props_t) -> i32;
    pub fn mi355_last_error(ctx: *mut mi355_ctx) -> *const c_char;
    pub fn mi355_last_global_error() -> *const c_char;
    pub fn mi355_error_count(ctx: *mut mi355_ctx, out_count: *mut i32) -> i32;
    pub fn mi355_error_pop(ctx: *mut mi355_ctx, out_code: *mut i32, out_requested: *mut u64, out_max: *mut u64,
                           msg: *mut c_char, msg_capacity: usize) -> i32;
    // Storage
    pub fn mi355_alloc(ctx: *mut mi355_ctx, bytes: u64, out_dptr: *mut *mut c_void) -> i32;
    pub fn mi355_free(ctx: *mut mi355_ctx, dptr: *mut c_void) -> i32;
    pub fn mi355_mem_info(ctx: *mut mi355_ctx, out_free: *mut u64, out_total: *mut u64) -> i32;
    pub fn mi355_pitched_row_bytes(ctx: *mut mi355_ctx, width_bytes: u64, out_pitch: *mut u64) -> i32;
    pub fn mi355_pinned_alloc(ctx: *mut mi355_ctx, bytes: u64, out_hptr: *mut *mut c_void) -> i32;
    pub fn mi355_pinned_free(ctx: *mut mi355_ctx, hptr: *mut c_void) -> i32;
    // Streams / events
    pub fn mi355_stream_create(ctx: *mut mi355_ctx, out: *mut mi355_stream) -> i32;
    pub fn mi355_stream_destroy(ctx: *mut mi355_ctx, stream: mi355_stream) -> i32;
    pub fn mi355_event_create(ctx: *mut mi355_ctx, out: *mut mi355_event) -> i32;
    pub fn mi355_event_destroy(ctx: *mut mi355_ctx, event: mi355_event) -> i32;
    pub fn mi355_event_record(ctx: *mut mi355_ctx, event: mi355_event, stream: mi355_stream) -> i32;
    pub fn mi355_stream_wait_event(ctx: *mut mi355_ctx, stream: mi355_stream, event: mi355_event) -> i32;
    pub fn mi355_event_sync(ctx: *mut mi355_ctx, event: mi355_event) -> i32;
    // IO
    pub fn mi355_write(ctx: *mut mi355_ctx, stream: mi355_stream, dst: *mut c_void, src: *const c_void, bytes: u64) -> i32;
    pub fn mi355_read(ctx: *mut mi355_ctx, stream: mi355_stream, dst: *mut c_void, src: *const c_void, bytes: u64) -> i32;
    pub fn mi355_write_2d(ctx: *mut mi355_ctx, stream: mi355_stream, dst: *mut c_void, dst_pitch: u64, src: *const c_void,
                          src_pitch: u64, width_bytes: u64, rows: u64) -> i32;
    pub fn mi355_read_2d(ctx: *mut mi355_ctx, stream: mi355_stream, dst: *mut c_void, dst_pitch: u64, src: *const c_void,
                         src_pitch: u64, width_bytes: u64, rows: u64) -> i32;
    pub fn mi355_memset(ctx: *mut mi355_ctx, stream: mi355_stream, dptr: *mut c_void, byte_value: i32, bytes: u64) -> i32;
    pub fn mi355_sync(ctx: *mut mi355_ctx, stream: mi355_stream) -> i32;
    pub fn mi355_flush(ctx: *mut mi355_ctx) -> i32;
    // Generic launch (the reference device ABI: one pointer per binding + the info pointer)
    pub fn mi355_module_load(ctx: *mut mi355_ctx, image: *const c_void, image_bytes: usize, out: *mut mi355_module) -> i32;
    pub fn mi355_module_get_function(ctx: *mut mi355_ctx, module: mi355_module, name: *const c_char,
                                     out: *mut mi355_function) -> i32;
    pub fn mi355_launch(ctx: *mut mi355_ctx, stream: mi355_stream, function: mi355_function, grid: *const u32,
                        block: *const u32, shared_mem_bytes: u32, buffer_ptrs: *const *mut c_void, num_ptrs: u32) -> i32;
    // Hot-path operations
    pub fn mi355_gemm(ctx: *mut mi355_ctx, stream: mi355_stream, desc: *const mi355_gemm_desc, a: *const c_void,
                      b: *const c_void, c: *mut c_void) -> i32;
    // cmma::execute(a, b, c, d): D = A * B + C (frontend/cmma.rs:1066-1110)
    pub fn mi355_gemm_add(ctx: *mut mi355_ctx, stream: mi355_stream, desc: *const mi355_gemm_desc, a: *const c_void, b: *const c_void,
                          c: *const c_void, d: *mut c_void) -> i32;
    pub fn mi355_reduce_workspace_bytes(ctx: *mut mi355_ctx, n: u64, out_bytes: *mut u64) -> i32;
    pub fn mi355_reduce_sum_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, n: u64, out: *mut f32,
                                workspace: *mut c_void, workspace_bytes: u64) -> i32;
    pub fn mi355_argmax_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, n: u64, out_val: *mut f32,
                            out_idx: *mut u64, workspace: *mut c_void, workspace_bytes: u64) -> i32;
    pub fn mi355_sum_argmax_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, n: u64, out_sum: *mut f32,
                                out_val: *mut f32, out_idx: *mut u64, workspace: *mut c_void, workspace_bytes: u64) -> i32;
    pub fn mi355_argmax_combine_f32(ctx: *mut mi355_ctx, stream: mi355_stream, records: *const c_void, count: u32,
                                    index_base: *const u64, out_val: *mut f32, out_idx: *mut u64) -> i32;
    pub fn mi355_sum_argmax_combine_f32(ctx: *mut mi355_ctx, stream: mi355_stream, records: *const c_void, count: u32,
                                        index_base: *const u64, out_sum: *mut f32, out_val: *mut f32, out_idx: *mut u64) -> i32;
    pub fn mi355_sum_argmax_exchange(ctx: *mut mi355_ctx, comm: *mut mi355_comm, stream: mi355_stream, record: *const c_void,
                                     gathered: *mut c_void, index_base: *const u64, out_sum: *mut f32, out_val: *mut f32,
                                     out_idx: *mut u64) -> i32;
    pub fn mi355_reduce_last_axis_sum_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, out: *mut f32,
                                          rows: u64, cols: u64, row_stride: u64) -> i32;
    // Collectives (RCCL over xGMI)
    pub fn mi355_comm_unique_id(id: *mut u8) -> i32;
    pub fn mi355_comm_init(ctx: *mut mi355_ctx, id: *const u8, rank: i32, world_size: i32, out: *mut *mut mi355_comm) -> i32;
    pub fn mi355_comm_destroy(ctx: *mut mi355_ctx, comm: *mut mi355_comm) -> i32;
    pub fn mi355_all_reduce(ctx: *mut mi355_ctx, comm: *mut mi355_comm, compute_stream: mi355_stream, src: *const c_void,
                            dst: *mut c_void, count: u64, dtype: i32, op: i32) -> i32;
    pub fn mi355_send(ctx: *mut mi355_ctx, comm: *mut mi355_comm, compute_stream: mi355_stream, src: *const c_void,
                      count: u64, dtype: i32, peer: i32) -> i32;
    pub fn mi355_recv(ctx: *mut mi355_ctx, comm: *mut mi355_comm, compute_stream: mi355_stream, dst: *mut c_void,
                      count: u64, dtype: i32, peer: i32) -> i32;
    pub fn mi355_sync_collective(ctx: *mut mi355_ctx, compute_stream: mi355_stream) -> i32;
    // Profiling
    pub fn mi355_profile_start(ctx: *mut mi355_ctx, stream: mi355_stream, out_token: *mut u64) -> i32;
    pub fn mi355_profile_stop(ctx: *mut mi355_ctx, stream: mi355_stream, token: u64, out_nanos: *mut u64) -> i32;
}

// ---- entry points added after the first cut of this crate (same header; not yet used by server.rs) -----------------
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355_gemm_scaled_desc {
    pub m: i64, pub n: i64, pub k: i64, pub batch: i64,
    pub lda: i64, pub ldb: i64, pub ldc: i64,
    pub ld_sa: i64, pub ld_sb: i64,
    pub stride_a: i64, pub stride_b: i64, pub stride_c: i64, pub stride_sa: i64, pub stride_sb: i64,
    pub dtype_a: i32, pub dtype_b: i32, pub dtype_c: i32,
    pub block: i32, pub algo: i32, pub reserved: i32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mi355_memory_usage {
    pub number_allocs: u64, pub bytes_in_use: u64, pub bytes_padding: u64, pub bytes_reserved: u64,
    pub driver_allocs: u64, pub driver_frees: u64, pub cache_hits: u64, pub reserved: u64,
}
#[repr(C)]
pub struct mi355_graph {
    _private: [u8; 0],
}
unsafe extern "C" {
    // MmaDefinition::execute_scaled at matmul level (cmma.rs:795-840)
    pub fn mi355_gemm_scaled(ctx: *mut mi355_ctx, stream: mi355_stream, desc: *const mi355_gemm_scaled_desc, a: *const c_void,
                             a_scales: *const c_void, b: *const c_void, b_scales: *const c_void, c: *mut c_void) -> i32;
    pub fn mi355_gemm_scaled_select(ctx: *mut mi355_ctx, desc: *const mi355_gemm_scaled_desc, out_algo: *mut i32) -> i32;
    pub fn mi355_gemm_select(ctx: *mut mi355_ctx, desc: *const mi355_gemm_desc, out_algo: *mut i32) -> i32;
    pub fn mi355_gemm_tail_plan(desc: *const mi355_gemm_desc, out_along_m: *mut i32, out_main_extent: *mut i64, out_splits: *mut i32) -> i32;
    pub fn mi355_gemm_split_plan(desc: *const mi355_gemm_desc, compute_units: i32, out_slices: *mut i32) -> i32;
    pub fn mi355_gemm_strip_plan(desc: *const mi355_gemm_desc, compute_units: i32, out_strip_bytes: *mut i32, out_strips: *mut i32, out_slices: *mut i32) -> i32;
    pub fn mi355_gemm_relayout_plan(desc: *const mi355_gemm_desc, out_relayout_a: *mut i32, out_relayout_b: *mut i32) -> i32;
    // an alternative to MemoryManagement-over-Mi355Storage for hosts without the reference's pool (memory_manage.rs)
    pub fn mi355_pool_alloc(ctx: *mut mi355_ctx, stream: mi355_stream, bytes: u64, out_dptr: *mut *mut c_void) -> i32;
    pub fn mi355_pool_free(ctx: *mut mi355_ctx, stream: mi355_stream, dptr: *mut c_void) -> i32;
    pub fn mi355_pool_cleanup(ctx: *mut mi355_ctx, explicit_cleanup: i32) -> i32;
    pub fn mi355_pool_mode(ctx: *mut mi355_ctx, mode: i32) -> i32;
    pub fn mi355_pool_usage(ctx: *mut mi355_ctx, out: *mut mi355_memory_usage) -> i32;
    // ComputeClient::to_client (client.rs:733-751)
    pub fn mi355_copy_to_ctx(src_ctx: *mut mi355_ctx, src_stream: mi355_stream, src_dptr: *const c_void, dst_ctx: *mut mi355_ctx,
                             dst_stream: mi355_stream, dst_dptr: *mut c_void, bytes: u64) -> i32;
    // tensor::identity::launch (crates/cubecl-std/src/tensor/identity.rs:36-84)
    pub fn mi355_fill_identity(ctx: *mut mi355_ctx, stream: mi355_stream, out: *mut c_void, dtype: i32, dim: u64, ld: u64) -> i32;
    // copy_into / into_contiguous / into_contiguous_packed (crates/cubecl-std/src/tensor/contiguous/launch.rs:5-56, base.rs:254-293)
    pub fn mi355_copy_strided(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, in_layout: *const mi355_tensor_layout,
                              out: *mut c_void, out_layout: *const mi355_tensor_layout, elem_size: i32) -> i32;
    pub fn mi355_copy_strided_plan(input: *const c_void, in_layout: *const mi355_tensor_layout, out: *const c_void,
                                   out_layout: *const mi355_tensor_layout, elem_size: i32, path: *mut i32, access_bytes: *mut i32) -> i32;
    pub fn mi355_copy_packed(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, in_storage: *const mi355_tensor_layout,
                             out: *mut c_void, out_storage: *const mi355_tensor_layout, shape: *const i64, packed_dim: i32,
                             packing: i32, word_size: i32) -> i32;
    // ComputeServer::{begin_capture, end_capture, replay} (server/base.rs:453-532)
    pub fn mi355_graph_begin_capture(ctx: *mut mi355_ctx, stream: mi355_stream) -> i32;
    pub fn mi355_graph_end_capture(ctx: *mut mi355_ctx, stream: mi355_stream, out_graph: *mut *mut mi355_graph) -> i32;
    pub fn mi355_graph_replay(ctx: *mut mi355_ctx, stream: mi355_stream, graph: *mut mi355_graph) -> i32;
    pub fn mi355_graph_destroy(ctx: *mut mi355_ctx, graph: *mut mi355_graph) -> i32;
    // array-wide reductions of f32 / bf16 / f16 input (f32 arithmetic)
    pub fn mi355_reduce_sum(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, n: u64, out: *mut f32,
                            workspace: *mut c_void, workspace_bytes: u64) -> i32;
    pub fn mi355_argmax(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, n: u64, out_val: *mut f32,
                        out_idx: *mut u64, workspace: *mut c_void, workspace_bytes: u64) -> i32;
    pub fn mi355_sum_argmax(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, n: u64, out_sum: *mut f32,
                            out_val: *mut f32, out_idx: *mut u64, workspace: *mut c_void, workspace_bytes: u64) -> i32;
    pub fn mi355_reduce_last_axis_sum(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, out: *mut f32,
                                      rows: u64, cols: u64, row_stride: u64) -> i32;
    pub fn mi355_reduce_last_axis_argmax(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, out_idx: *mut u32,
                                         rows: u64, cols: u64, row_stride: u64) -> i32;
    pub fn mi355_reduce_axis_sum(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, out: *mut f32,
                                 outer: u64, reduce: u64, inner: u64) -> i32;
    pub fn mi355_reduce_axis_argmax(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, out_idx: *mut u32,
                                    outer: u64, reduce: u64, inner: u64) -> i32;
    // every reduce operation (sum / mean / max / min / prod values, argmax / argmin indices), array-wide and over one axis
    pub fn mi355_reduce(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, n: u64, op: i32, out: *mut f32,
                        workspace: *mut c_void, workspace_bytes: u64) -> i32;
    pub fn mi355_argreduce(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, n: u64, op: i32, out_val: *mut f32,
                           out_idx: *mut u64, workspace: *mut c_void, workspace_bytes: u64) -> i32;
    pub fn mi355_reduce_axis(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, op: i32, out: *mut f32,
                             outer: u64, reduce: u64, inner: u64) -> i32;
    pub fn mi355_argreduce_axis(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const c_void, dtype: i32, op: i32, out_idx: *mut u32,
                                outer: u64, reduce: u64, inner: u64) -> i32;
    // reductions over any axis, plane ops
    pub fn mi355_reduce_axis_sum_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, out: *mut f32, outer: u64,
                                     reduce: u64, inner: u64) -> i32;
    pub fn mi355_reduce_axis_argmax_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, out_idx: *mut u32, outer: u64,
                                        reduce: u64, inner: u64) -> i32;
    pub fn mi355_reduce_last_axis_argmax_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, out_idx: *mut u32,
                                             rows: u64, cols: u64, row_stride: u64) -> i32;
    pub fn mi355_plane_reduce_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, out: *mut f32, n: u64, active: u32, op: i32) -> i32;
    pub fn mi355_plane_op_f32(ctx: *mut mi355_ctx, stream: mi355_stream, input: *const f32, out: *mut c_void, n: u64, plane: u32, op: i32, arg: u32) -> i32;
    // synthetic data, casts, copies
    pub fn mi355_fill_uniform(ctx: *mut mi355_ctx, stream: mi355_stream, dst: *mut c_void, dtype: i32, n: u64, seed: u64,
                              tensor: u64, lo: f32, hi: f32) -> i32;
    pub fn mi355_cast(ctx: *mut mi355_ctx, stream: mi355_stream, src: *const c_void, src_dtype: i32, dst: *mut c_void,
                      dst_dtype: i32, n: u64) -> i32;
    pub fn mi355_copy_d2d(ctx: *mut mi355_ctx, stream: mi355_stream, dst: *mut c_void, src: *const c_void, bytes: u64) -> i32;
    pub fn mi355_read_async(ctx: *mut mi355_ctx, stream: mi355_stream, dst_host: *mut c_void, src_dptr: *const c_void, bytes: u64) -> i32;
    // throughput probes (cubecl-std throughput runners)
    pub fn mi355_probe_memory_read(ctx: *mut mi355_ctx, stream: mi355_stream, buf: *const c_void, bytes: u64, iters: u32, sink: *mut c_void) -> i32;
    pub fn mi355_probe_memory_copy(ctx: *mut mi355_ctx, stream: mi355_stream, src: *const c_void, dst: *mut c_void, bytes: u64) -> i32;
    pub fn mi355_probe_memory_write(ctx: *mut mi355_ctx, stream: mi355_stream, dst: *mut c_void, bytes: u64) -> i32;
    pub fn mi355_probe_mfma(ctx: *mut mi355_ctx, stream: mi355_stream, dtype_ab: i32, iters: u32, sink: *mut c_void, out_ops: *mut u64) -> i32;
    pub fn mi355_probe_mfma_data(ctx: *mut mi355_ctx, stream: mi355_stream, mode: i32, iters: u32, sink: *mut c_void, out_ops: *mut u64) -> i32;
    pub fn mi355_probe_compute_direct(ctx: *mut mi355_ctx, stream: mi355_stream, iters: u32, sink: *mut c_void, out_ops: *mut u64) -> i32;
    pub fn mi355_probe_launch_overhead(ctx: *mut mi355_ctx, stream: mi355_stream, launches: u32, sink: *mut c_void) -> i32;
    pub fn mi355_probe_clock(ctx: *mut mi355_ctx, stream: mi355_stream, dev_out: *mut u64) -> i32;
    // streams / events / modules / collectives the server does not need on its main path (kept so that the table is the header's)
    pub fn mi355_default_stream(ctx: *mut mi355_ctx, out_stream: *mut mi355_stream) -> i32;
    pub fn mi355_comm_stream(ctx: *mut mi355_ctx, out_stream: *mut mi355_stream) -> i32;
    pub fn mi355_event_elapsed_ms(ctx: *mut mi355_ctx, start: mi355_event, stop: mi355_event, out_ms: *mut f32) -> i32;
    pub fn mi355_module_unload(ctx: *mut mi355_ctx, module: mi355_module) -> i32;
    pub fn mi355_all_gather(ctx: *mut mi355_ctx, comm: *mut mi355_comm, compute_stream: mi355_stream, src: *const c_void,
                            dst: *mut c_void, count: u64, dtype: i32) -> i32;
}
