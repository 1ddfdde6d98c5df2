/*
 * mi355cube.h -- C ABI of libmi355cube.so: an MI355X (gfx950 / CDNA4) native backend for the
 * tiled-matmul + reduction hot path of tracel-ai/cubecl.
 *
 * This header IS the drop-in boundary.  Each entry point replaces one operation of the
 * reference's backend trait surface (citations are file:line under the reference checkout):
 *
 *   Runtime                crates/cubecl-runtime/src/runtime.rs:14-52
 *   ComputeServer          crates/cubecl-runtime/src/server/base.rs:372-601
 *   ServerCommunication    crates/cubecl-runtime/src/server/base.rs:632-737
 *   ComputeStorage         crates/cubecl-runtime/src/storage/base.rs:74-101
 *   HIP launch leaf        crates/cubecl-hip/src/compute/context.rs:390-440
 *
 * A Rust `cubecl-mi355` crate (rust/cubecl-mi355) implements those traits by
 * forwarding to these functions; tests and benches drive the same functions from Python
 * (ctypes) and C++.
 *
 * Conventions
 *  - Plain C: opaque pointers, integers, caller-owned buffers.  No exceptions or panics cross
 *    the boundary.  Every function returns an int32 status (MI355_OK == 0).
 *  - Threading: one `mi355_ctx` per device, used by one thread at a time (the reference runs
 *    one runner thread per DeviceId: crates/cubecl-common/src/device/handle/channel.rs:24-37).
 *    Different contexts may be used concurrently.  Each call re-selects its device
 *    (hipSetDevice), so a context may migrate between threads.
 *  - Asynchrony and errors: `launch`, copies and the op entry points are stream-ordered and
 *    fire-and-forget like ComputeServer::launch/write.  Failures discovered at submission
 *    (resource limits, bad launch) are QUEUED on the context and reported by the next
 *    mi355_flush / mi355_sync / mi355_read as MI355_E_SERVER_UNHEALTHY (the reference's
 *    ServerError::ServerUnhealthy{errors}, server/base.rs:286-332); drain the queue with
 *    mi355_error_pop.  Argument errors (NULL pointers, unsupported dtype/shape) are returned
 *    immediately and also recorded for mi355_last_error.
 *  - Ownership: the caller owns every device pointer it gets from mi355_alloc and every
 *    workspace it passes in; the library never frees caller memory.  mi355_free defers the
 *    hipFree to the next mi355_flush, as GpuStorage does
 *    (crates/cubecl-hip/src/compute/storage/gpu.rs:136-169).
 *  - Strides are in ELEMENTS, row-major (TensorHandle, crates/cubecl-std/src/tensor/handle.rs:13-23).
 */
#ifndef MI355CUBE_H
#define MI355CUBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 9

/* ---- status codes (map onto LaunchError / IoError / ServerError, server/base.rs:177-332,
 *      :884-1019; the Rust shim performs the conversion) ------------------------------- */
enum {
    MI355_OK = 0,
    MI355_E_INVALID_ARGUMENT = 1,      /* ServerError::Validation                              */
    MI355_E_OUT_OF_MEMORY = 2,         /* IoError::OutOfMemory / LaunchError::OutOfMemory      */
    MI355_E_BUFFER_TOO_BIG = 3,        /* IoError::BufferTooBig (> memory.max_page_size)       */
    MI355_E_UNSUPPORTED_STRIDES = 4,   /* IoError::UnsupportedStrides                          */
    MI355_E_NOT_FOUND = 5,             /* IoError::NotFound (unknown symbol / handle)          */
    MI355_E_SHARED_MEMORY = 6,         /* ResourceLimitError::SharedMemory{requested,max}      */
    MI355_E_UNITS = 7,                 /* ResourceLimitError::Units                            */
    MI355_E_CUBE_DIM = 8,              /* ResourceLimitError::CubeDim                          */
    MI355_E_MAX_UNITS_PER_CUBE = 9,    /* ResourceLimitError::MaxUnitPerCube                   */
    MI355_E_COMPILATION = 10,          /* LaunchError::CompilationError (module load failed)   */
    MI355_E_LAUNCH = 11,               /* LaunchError::Unknown                                 */
    MI355_E_EXECUTION = 12,            /* IoError::Execution / ServerError::Generic            */
    MI355_E_UNSUPPORTED = 13,          /* IoError::UnsupportedIoOperation / unsupported op     */
    MI355_E_SERVER_UNHEALTHY = 14,     /* ServerError::ServerUnhealthy{errors}: pop the queue  */
    MI355_E_COMM = 15,                 /* collective failure (RCCL)                            */
    MI355_E_NO_DEVICE = 16,            /* no HIP device / HIP runtime unusable                 */
    MI355_E_PROFILE = 17               /* ProfileError                                         */
};

/* ---- element types (subset of cubecl_ir::ElemType this path moves) ---------------------- */
enum {
    MI355_DTYPE_F32 = 0,
    MI355_DTYPE_BF16 = 1,
    MI355_DTYPE_F16 = 2,
    MI355_DTYPE_F64 = 3,
    MI355_DTYPE_I32 = 4,
    MI355_DTYPE_U32 = 5,
    MI355_DTYPE_I64 = 6,
    MI355_DTYPE_U64 = 7,
    MI355_DTYPE_U8 = 8,
    MI355_DTYPE_I8 = 9,
    MI355_DTYPE_F8E4M3 = 10, /* OCP e4m3fn: no infinities, S.1111.111 = NaN, max 448 (fp8_e4m3.rs:12-37) */
    MI355_DTYPE_F8E5M2 = 11, /* OCP e5m2: IEEE-style inf / NaN, max 57344 (fp8_e5m2.rs:12-38) */
    MI355_DTYPE_F4E2M1X2 = 12, /* two e2m1 per byte, the first in the low nibble (fp4.rs:204-224); sizes are in BYTES */
    MI355_DTYPE_UE8M0 = 13,  /* block scale 2^(bits - 127), 0xFF = NaN (fp8/fp8_e8m0.rs) */
    /* types a backend ADVERTISES for generated kernels (register_supported_types, crates/cubecl-cpp/src/shared/base.rs:
     * 322-375); this library's own entry points do not take them, kernels launched through mi355_launch may */
    MI355_DTYPE_I16 = 14,
    MI355_DTYPE_U16 = 15,
    MI355_DTYPE_BOOL = 16,
    MI355_DTYPE_FLEX32 = 17, /* FloatKind::Flex32: f32 storage, relaxed compute precision */
    MI355_DTYPE_INDEX = 18   /* ElemType::Index: the kernel's address type (u32 or u64) */
};

/* TypeUsage / AtomicUsage bit sets (crates/cubecl-ir/src/features.rs:79-88, :110-123), bit = 1 << variant order */
enum { MI355_TYPE_USAGE_CONVERSION = 1, MI355_TYPE_USAGE_ARITHMETIC = 2, MI355_TYPE_USAGE_DOT_PRODUCT = 4, MI355_TYPE_USAGE_BUFFER = 8,
       MI355_TYPE_USAGE_ALL = 15 };
enum { MI355_ATOMIC_LOAD_STORE = 1, MI355_ATOMIC_EXCHANGE = 2, MI355_ATOMIC_ADD = 4, MI355_ATOMIC_MIN_MAX = 8, MI355_ATOMIC_BITWISE = 16,
       MI355_ATOMIC_COMPARE_EXCHANGE = 32, MI355_ATOMIC_ALL = 63 };
enum { MI355_ADDRESS_TYPE_U32 = 1, MI355_ADDRESS_TYPE_U64 = 2 };   /* register_address_type (base.rs:323-324) */
enum { MI355_LAYOUT_ROW_MAJOR = 0, MI355_LAYOUT_COL_MAJOR = 1 };   /* cmma::MatrixLayout */

/* ReduceOperation (server/base.rs:623-628) + the operations the reductions and the argmax exchange need beside it.
 * Sum and Mean are the reference's; Max / Min / Prod are an API delta for the collectives (RCCL has all three)
 * and the value reductions of mi355_reduce / mi355_reduce_axis; ArgMax / ArgMin exist only as local reductions
 * (mi355_argreduce / mi355_argreduce_axis): mi355_all_reduce refuses them. */
enum { MI355_REDUCE_SUM = 0, MI355_REDUCE_MEAN = 1, MI355_REDUCE_MAX = 2, MI355_REDUCE_MIN = 3, MI355_REDUCE_PROD = 4,
       MI355_REDUCE_ARGMAX = 5, MI355_REDUCE_ARGMIN = 6 };
/* scans of mi355_plane_reduce_f32 (plane_inclusive_sum / _exclusive_sum / _inclusive_prod / _exclusive_prod,
 * crates/cubecl-core/src/frontend/plane.rs:242-283, :309, :334) */
enum { MI355_PLANE_INCLUSIVE_SUM = 101, MI355_PLANE_EXCLUSIVE_SUM = 102, MI355_PLANE_INCLUSIVE_PROD = 103, MI355_PLANE_EXCLUSIVE_PROD = 104 };

typedef struct mi355_ctx mi355_ctx;
typedef void *mi355_stream;   /* hipStream_t; NULL = the context's compute stream */
typedef void *mi355_event;    /* hipEvent_t */
typedef void *mi355_module;   /* hipModule_t */
typedef void *mi355_function; /* hipFunction_t */
typedef struct mi355_comm mi355_comm;

/* ---- device description: HardwareProperties / MemoryDeviceProperties / DeviceIdentity /
 *      Features (crates/cubecl-ir/src/properties.rs:26-108, features.rs:10-32); values for
 *      gfx950 follow SURVEY.md Appendix C ------------------------------------------------ */
typedef struct {
    uint32_t m, n, k;
    int32_t a_type, b_type, cd_type; /* MI355_DTYPE_* */
} mi355_mma_config;                  /* cubecl_ir::features::MmaConfig (features.rs:145) */

typedef struct {
    uint32_t m, n, k;
    int32_t a_type, b_type, cd_type, scales_type; /* MI355_DTYPE_*; scales_type = MI355_DTYPE_UE8M0 */
    uint32_t scales_factor;                       /* scales per k of one instruction (k / 32)       */
} mi355_scaled_mma_config; /* ScaledMmaConfig (what test_cmma_scaled looks up, runtime_tests/cmma.rs:1493-1505) */

typedef struct {
    int32_t dtype;       /* MI355_DTYPE_* */
    uint32_t usage;      /* MI355_TYPE_USAGE_* bits, or MI355_ATOMIC_* bits in the atomic table */
} mi355_type_usage;      /* one entry of DeviceProperties' type_usage / atomic_type_usage maps */

/* TargetProperties.mma for the matrix cores of this device (cubecl_ir::MmaProperties, crates/cubecl-ir/src/
 * runtime_properties.rs:19-39; the reference's HIP values describe RDNA WMMA only, crates/cubecl-hip/src/runtime.rs:282-304).
 * MFMA on gfx950 (cdna_hip_programming.md section 3): a lane holds k-CONTIGUOUS elements of one A row and of one B
 * column (8 x 16-bit, 1 x f32, 16 x 8-bit per 128-bit register group) and, of the accumulator, 4 consecutive ROWS of one
 * column per register quad (32x32: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), col = lane & 31) -- nothing is held twice. */
typedef struct {
    uint32_t register_size_bits;        /* 32 */
    uint32_t const_plane_size;          /* 64 */
    uint32_t register_layout_a;         /* MI355_LAYOUT_ROW_MAJOR: k runs along a register group */
    uint32_t register_layout_b;         /* MI355_LAYOUT_COL_MAJOR */
    uint32_t register_layout_acc;       /* MI355_LAYOUT_COL_MAJOR: consecutive registers walk down a column */
    uint32_t register_duplication_a, register_duplication_b, register_duplication_acc;   /* 1, 1, 1 */
    uint32_t contiguous_elements_ab_bits;   /* contiguous operand elements per lane = this / element bits: 128 (f32: one element) */
    uint32_t contiguous_elements_acc;       /* 4 */
} mi355_mma_properties;

typedef struct {
    uint32_t abi_version;
    int32_t device_index;
    char name[64];                /* hipDeviceProp_t.name                                   */
    char gcn_arch_name[64];       /* "gfx950:sramecc+:xnack-"                               */
    char fingerprint[96];         /* DeviceIdentity.fingerprint: "mi355-aot_<gcnArchName>"  */
    uint32_t load_width_bits;     /* 128                                                    */
    uint32_t plane_size_min;      /* 64                                                     */
    uint32_t plane_size_max;      /* 64                                                     */
    uint32_t max_bindings;        /* 1024                                                   */
    uint64_t max_shared_memory_size;   /* bytes of LDS one cube may use                    */
    uint32_t max_cube_count[3];
    uint32_t max_units_per_cube;
    uint32_t max_cube_dim[3];
    uint32_t num_streaming_multiprocessors; /* CUs (256 on MI355X)                         */
    uint32_t num_tensor_cores;              /* matrix pipes per CU (4)                     */
    uint32_t min_tensor_cores_dim;
    uint32_t num_xcd;                       /* 8: private-L2 chiplets, for XCD-aware grids  */
    uint64_t total_memory;
    uint64_t max_page_size;       /* total/4 (crates/cubecl-hip/src/runtime.rs:155-158)     */
    uint64_t mem_alignment;       /* max(32, texture/surface alignment) (:82,115-116)       */
    uint32_t clock_khz;
    uint32_t memory_clock_khz;
    uint32_t memory_bus_width_bits;
    uint32_t l2_cache_bytes;
    uint32_t plane_ops;               /* 1: features.plane has Ops                          */
    uint32_t plane_non_uniform;       /* 1: NonUniformControlFlow                           */
    uint32_t timing_method_device;    /* 1: profile() reports GPU time from hipEvents       */
    uint32_t server_comm_enabled;     /* ServerCommunication::SERVER_COMM_ENABLED (RCCL found) */
    uint32_t num_mma_configs;
    mi355_mma_config mma_configs[16]; /* features.matmul.cmma / .mma for gfx950 MFMA        */
    uint32_t num_scaled_mma_configs;  /* features.matmul.scaled_mma (register_scaled_mma_features,
                                         crates/cubecl-cpp/src/shared/mma.rs:21-46); since ABI 2   */
    mi355_scaled_mma_config scaled_mma_configs[8];
    /* since ABI 4: what register_supported_types advertises (crates/cubecl-cpp/src/shared/base.rs:322-375) */
    uint32_t address_types;           /* MI355_ADDRESS_TYPE_* bits                              */
    uint32_t num_type_usage;
    mi355_type_usage type_usage[24];  /* every supported ElemType with its TypeUsage set        */
    uint32_t num_atomic_usage;
    mi355_type_usage atomic_usage[8]; /* atomic element types with their AtomicUsage set        */
    mi355_mma_properties mma_properties;   /* TargetProperties.mma for MFMA                     */
} mi355_device_props_t;

/* =================================== Runtime ============================================= */

int32_t mi355_abi_version(void);
/* Runtime::enumerate_devices (crates/cubecl-hip/src/runtime.rs:306-327). */
int32_t mi355_device_count(int32_t *out_count);
/* DeviceService::init for one DeviceId (crates/cubecl-hip/src/runtime.rs:61-247): selects the
 * device, refuses anything that is not wave64 gfx950, creates the non-blocking compute stream
 * and the communication stream, fills the property block. */
int32_t mi355_ctx_create(int32_t device_index, mi355_ctx **out_ctx);
int32_t mi355_ctx_destroy(mi355_ctx *ctx);
int32_t mi355_device_props(mi355_ctx *ctx, mi355_device_props_t *out_props);
/* message of the most recent failing call on this context (never NULL) */
const char *mi355_last_error(mi355_ctx *ctx);
/* ctx == NULL variant for failures of mi355_ctx_create itself */
const char *mi355_last_global_error(void);

/* ---- queued (asynchronous) errors ------------------------------------------------------- */
int32_t mi355_error_count(mi355_ctx *ctx, int32_t *out_count);
/* pops the oldest queued error: code, two numeric details (e.g. requested / max) and text */
int32_t mi355_error_pop(mi355_ctx *ctx, int32_t *out_code, uint64_t *out_requested,
                        uint64_t *out_max, char *msg, size_t msg_capacity);

/* =================================== Storage ============================================= */

/* ComputeStorage::alloc (storage/gpu.rs:136-169): plain hipMalloc.  A driver OOM maps to
 * MI355_E_OUT_OF_MEMORY, a request above max_page_size to MI355_E_BUFFER_TOO_BIG
 * (server/base.rs:895-911).  bytes == 0 returns a NULL pointer and MI355_OK. */
int32_t mi355_alloc(mi355_ctx *ctx, uint64_t bytes, void **out_dptr);
/* ComputeStorage::dealloc: deferred until mi355_flush. */
int32_t mi355_free(mi355_ctx *ctx, void *dptr);
int32_t mi355_mem_info(mi355_ctx *ctx, uint64_t *out_free, uint64_t *out_total);

/* ----- memory pool: what ComputeClient::empty / create reserve from ---------------------------
 * MemoryManagement of the reference (crates/cubecl-runtime/src/memory_management/memory_manage.rs):
 * `reserve` (:1084), `cleanup` (:938), `memory_usage` (:1238), `mode` (:900).  A stream-ordered
 * caching allocator: requests up to 32 MiB are slices of quarter-octave size classes carved out of
 * slab pages, larger ones exclusive pages (2 MiB granules) cached by size when freed.  A freed block
 * is reused at once by the stream it was freed on and by other streams once that stream has passed
 * the free point (an event, never a device synchronisation).  Cached exclusive pages go back to the
 * driver after 5000 x (1 + size / 1 GiB) reservations without reuse (:242, :641-651) or on an
 * explicit cleanup; nothing is released and no driver allocation is made inside a graph-capture
 * window (:948-952) -- a request that would need one fails with MI355_E_UNSUPPORTED.
 * mi355_alloc / mi355_free above stay the raw ComputeStorage page calls. */
enum {
    MI355_ALLOC_MODE_AUTO = 0,       /* MemoryAllocationMode::Auto (memory_manage.rs:155-162) */
    MI355_ALLOC_MODE_PERSISTENT = 1  /* ::Persistent: exact-size pages that periodic cleanup never releases */
};
typedef struct {
    uint64_t number_allocs;  /* MemoryUsage (memory_management/base.rs:8-28): live reservations     */
    uint64_t bytes_in_use;   /* bytes requested by them                                              */
    uint64_t bytes_padding;  /* rounding on top of that                                              */
    uint64_t bytes_reserved; /* device memory the pool holds (in use + cached)                       */
    uint64_t driver_allocs;  /* additions: hipMalloc / hipFree calls made and requests served from   */
    uint64_t driver_frees;   /* the cache since the context was created                              */
    uint64_t cache_hits;
    uint64_t reserved;
} mi355_memory_usage;
int32_t mi355_pool_alloc(mi355_ctx *ctx, mi355_stream stream, uint64_t bytes, void **out_dptr);
int32_t mi355_pool_free(mi355_ctx *ctx, mi355_stream stream, void *dptr);
int32_t mi355_pool_cleanup(mi355_ctx *ctx, int32_t explicit_cleanup);
int32_t mi355_pool_mode(mi355_ctx *ctx, int32_t mode);
int32_t mi355_pool_usage(mi355_ctx *ctx, mi355_memory_usage *out);
/* MemoryLayoutPolicy::apply for a rank>=2 tensor (crates/cubecl-runtime/src/allocator.rs:21-72):
 * returns the row pitch in bytes for rows of `width_bytes`. */
int32_t mi355_pitched_row_bytes(mi355_ctx *ctx, uint64_t width_bytes, uint64_t *out_pitch);
/* pinned host staging (crates/cubecl-hip/src/compute/storage/cpu.rs:96-140) */
int32_t mi355_pinned_alloc(mi355_ctx *ctx, uint64_t bytes, void **out_hptr);
int32_t mi355_pinned_free(mi355_ctx *ctx, void *hptr);

/* =================================== Streams / events ==================================== */

int32_t mi355_stream_create(mi355_ctx *ctx, mi355_stream *out_stream); /* hipStreamNonBlocking */
int32_t mi355_stream_destroy(mi355_ctx *ctx, mi355_stream stream);
int32_t mi355_default_stream(mi355_ctx *ctx, mi355_stream *out_stream);
int32_t mi355_comm_stream(mi355_ctx *ctx, mi355_stream *out_stream);
int32_t mi355_event_create(mi355_ctx *ctx, mi355_event *out_event);
int32_t mi355_event_destroy(mi355_ctx *ctx, mi355_event event);
int32_t mi355_event_record(mi355_ctx *ctx, mi355_event event, mi355_stream stream);
/* Fence::wait_async (crates/cubecl-hip/src/compute/fence.rs): hipStreamWaitEvent */
int32_t mi355_stream_wait_event(mi355_ctx *ctx, mi355_stream stream, mi355_event event);
/* Fence::wait_sync: hipEventSynchronize */
int32_t mi355_event_sync(mi355_ctx *ctx, mi355_event event);
int32_t mi355_event_elapsed_ms(mi355_ctx *ctx, mi355_event start, mi355_event stop,
                               float *out_ms);

/* =================================== IO =================================================== */

/* ComputeServer::write (command.rs:347-411).  Stream-ordered; `src` must stay valid until the
 * stream passed the copy (mi355_sync / an event).  bytes == 0 is a no-op. */
int32_t mi355_write(mi355_ctx *ctx, mi355_stream stream, void *dst_dptr, const void *src_host,
                    uint64_t bytes);
/* ComputeServer::read (command.rs:244-265, :538-614): enqueues the D2H copy, waits for it and
 * reports queued errors (MI355_E_SERVER_UNHEALTHY). */
int32_t mi355_read(mi355_ctx *ctx, mi355_stream stream, void *dst_host, const void *src_dptr,
                   uint64_t bytes);
int32_t mi355_read_async(mi355_ctx *ctx, mi355_stream stream, void *dst_host,
                         const void *src_dptr, uint64_t bytes);
/* pitched (2-D) forms used when PitchedMemoryLayoutPolicy padded the rows (command.rs:318-322);
 * pitches in bytes. */
int32_t mi355_write_2d(mi355_ctx *ctx, mi355_stream stream, void *dst_dptr, uint64_t dst_pitch,
                       const void *src_host, uint64_t src_pitch, uint64_t width_bytes,
                       uint64_t rows);
int32_t mi355_read_2d(mi355_ctx *ctx, mi355_stream stream, void *dst_host, uint64_t dst_pitch,
                      const void *src_dptr, uint64_t src_pitch, uint64_t width_bytes,
                      uint64_t rows);
int32_t mi355_copy_d2d(mi355_ctx *ctx, mi355_stream stream, void *dst_dptr,
                       const void *src_dptr, uint64_t bytes);
/* ComputeClient::to_client (client.rs:733-751; runtime_tests/to_client.rs): copies `bytes` from a buffer of `src_ctx`
 * into a buffer of `dst_ctx` (another device of the same process, or the same one) as a peer copy over xGMI, ordered
 * after the work queued on the source stream; the source stream waits for the copy in turn. */
int32_t mi355_copy_to_ctx(mi355_ctx *src_ctx, mi355_stream src_stream, const void *src_dptr,
                          mi355_ctx *dst_ctx, mi355_stream dst_stream, void *dst_dptr, uint64_t bytes);
int32_t mi355_memset(mi355_ctx *ctx, mi355_stream stream, void *dptr, int32_t byte_value,
                     uint64_t bytes); /* zeros_array, tensor/handle.rs:199-207 */
/* ComputeServer::sync: waits for the stream, then reports queued errors. */
int32_t mi355_sync(mi355_ctx *ctx, mi355_stream stream);
/* ComputeServer::flush: releases deferred frees whose work has completed, reports queued
 * errors. Does not block on the device unless frees are pending. */
int32_t mi355_flush(mi355_ctx *ctx);

/* =================================== Generic launch ====================================== */

/* Loads a gfx950 code object (what hiprtcGetCode would have produced,
 * crates/cubecl-hip/src/compute/context.rs:334-387).  Failure -> MI355_E_COMPILATION. */
int32_t mi355_module_load(mi355_ctx *ctx, const void *image, size_t image_bytes,
                          mi355_module *out_module);
int32_t mi355_module_unload(mi355_ctx *ctx, mi355_module module);
int32_t mi355_module_get_function(mi355_ctx *ctx, mi355_module module, const char *name,
                                  mi355_function *out_function);
/* ComputeServer::launch -> HipContext::execute_task (context.rs:390-440): the reference device
 * ABI -- one pointer per buffer binding followed by the `info` pointer
 * (crates/cubecl-cpp/src/hip/signature.rs:28-62), passed as an array of pointers.
 * A zero in `grid` makes the launch a no-op (client.rs:880-884).  Resource-limit violations are
 * queued, not returned (runtime_tests/launch.rs:226-348). */
int32_t mi355_launch(mi355_ctx *ctx, mi355_stream stream, mi355_function function,
                     const uint32_t grid[3], const uint32_t block[3], uint32_t shared_mem_bytes,
                     void *const *buffer_ptrs, uint32_t num_ptrs);

/* =================================== Data generation / casts ============================== */

/* Counter-based uniform fill, bit-identical to oracle_fill_uniform_f32 (oracle/oracle.c):
 * dst[i] = lo + (hi-lo) * u(seed, tensor, i); output dtype f32 / bf16 / f16 (RNE cast). */
int32_t mi355_fill_uniform(mi355_ctx *ctx, mi355_stream stream, void *dst, int32_t dtype,
                           uint64_t n, uint64_t seed, uint64_t tensor, float lo, float hi);
/* cmma::cast (frontend/cmma.rs:1113-1213) widened to whole buffers: f32 <-> bf16 / f16. */
int32_t mi355_cast(mi355_ctx *ctx, mi355_stream stream, const void *src, int32_t src_dtype,
                   void *dst, int32_t dst_dtype, uint64_t n);

/* =================================== GEMM ================================================ */

/* Tiled matmul launch: C[b] = A[b] * B[b] (f32 accumulate), the operation cubek-matmul's
 * launcher hands to a backend and `cmma::execute` defines per fragment
 * (crates/cubecl-core/src/frontend/cmma.rs:1066-1110; semantics pinned by
 * runtime_tests/cmma.rs:695-722).
 *   trans_a == 0: A is row-major [M][K] (lda >= K)      trans_a == 1: A stored [K][M] (lda >= M)
 *   trans_b == 0: B is row-major [K][N] (ldb >= N)      trans_b == 1: B stored [N][K] (ldb >= K)
 *                                                        i.e. Out = Lhs * Rhs^T, the ColMajor-B
 *                                                        form of the cmma tests (cmma.rs:23)
 * Batch strides are in elements; 0 broadcasts an operand
 * (crates/cubecl-std/src/tensor/matrix_batch_layout.rs:21-79).
 * dtype_ab: F32 (MFMA f32, exact-f32 products), BF16 or F16 (MFMA, f32 accumulate), F8E4M3 or F8E5M2
 *           (OCP FP8 on v_mfma_f32_32x32x64_f8f6f4, unscaled, f32 accumulate; both operands the same format).
 * dtype_c: F32, or the same 16-bit type as the inputs (RNE on store); fp8 inputs write F32, BF16 or F16.
 * Row-major B (what TensorHandle::new_contiguous gives a rhs) is staged natively by the tile kernels for F32, BF16 and F16
 * (N a multiple of 16 bytes' worth of columns), and so is A stored [K][M] together with a row-major B for BF16 / F16 on the
 * 128x128 and the 256x256 kernel (lhs^T . grad_out; M a multiple of 8).  What the MFMA kernels do not stage directly -- trans_a otherwise, fp8 row-major B, a
 * row-major B of at most 64 columns, K not a multiple of the K-tile, rows or bases not 16-byte aligned -- is first re-laid
 * out K-contiguous (zero-padded) into library-owned per-stream scratch, as the reference's launchers do with into_contiguous
 * (mi355_gemm_relayout_plan says which operands); tiny shapes with K <= 64 run on the bounds-checked generic kernel (it walks K serially).  The library owns:
 * that scratch, the split-K slabs of skinny shapes, and the reductions' arrival tickets -- never caller memory.
 * fp8: v_mfma_f32_32x32x64_f8f6f4 adds its products in groups of 8, each cut 13 bits below the group's largest product
 * (measured) -- MI355_GEMM_ALGO_GENERIC gives the exact-product f32 chain. */
typedef struct {
    int64_t m, n, k, batch;
    int64_t lda, ldb, ldc;
    int64_t stride_a, stride_b, stride_c;
    int32_t dtype_ab, dtype_c;
    int32_t trans_a, trans_b;
    int32_t algo;   /* MI355_GEMM_ALGO_*; 0 = pick the fastest kernel that supports the shape */
    int32_t reserved;
} mi355_gemm_desc;

enum {
    MI355_GEMM_ALGO_AUTO = 0,
    MI355_GEMM_ALGO_GENERIC = 1,  /* bounds-checked scalar-FMA kernel, any shape / layout     */
    MI355_GEMM_ALGO_F32_MFMA = 2, /* 128x128 LDS-tiled v_mfma_f32_32x32x2_f32                 */
    MI355_GEMM_ALGO_LP_128 = 3,   /* bf16/f16 128x128x64 LDS-DMA tile, v_mfma_f32_32x32x16     */
    MI355_GEMM_ALGO_LP_256 = 4,   /* retired (ABI 8): was the 8-wave 256x256 kernel; now an alias of _LP_256W4  */
    MI355_GEMM_ALGO_LP_256W4 = 5, /* fp8/bf16/f16/f32 256x256 tile x 128-byte K line, 4 waves x 128x128 */
    MI355_GEMM_ALGO_LP_256P = 6,  /* the same tile as a persistent kernel: one workgroup per CU walks
                                     several output tiles with a continuous K-tile stream          */
    MI355_GEMM_ALGO_LP_256Q = 7,  /* the persistent kernel with the finished tile held in registers (16-bit C) and its
                                     stores dripped into the next tile's K loop                      */
    MI355_GEMM_ALGO_SKINNY = 8,   /* bf16/f16, M <= 16 or N <= 16: the large operand streamed once from HBM,
                                     v_dot2c_f32 accumulation, no matrix core (gemm_skinny.hip)      */
    MI355_GEMM_ALGO_STREAM64 = 9, /* bf16/f16, M <= 64 or N <= 64: 32 streamed rows x the whole K per workgroup,
                                     loader waves + MFMA, no split-K (gemm_stream64.hip); f32 operands: the same shapes
                                     on v_mfma_f32_16x16x4_f32 with per-wave LDS rings (gemm_stream64_f32.hip)         */
    MI355_GEMM_ALGO_LP_256X128 = 10, /* bf16/f16 256x128x64 tile (gemm_lp128.hip, MI = 4): three-stage LDS ring + loader
                                     waves, one workgroup per CU; mid-size shapes of at most one round of such tiles */
    MI355_GEMM_ALGO_NNROWS = 11,  /* bf16/f16, M <= 16 against a row-major [K][N] weight (the rhs TensorHandle::new_contiguous
                                     gives): wide row strips streamed once, transposed in registers into 4x4x4 MFMA
                                     operands, K slices folded by the last workgroup to arrive (gemm_nnrows.hip) */
    MI355_GEMM_ALGO_LP_256X192 = 12, /* bf16/f16, [N][K] or row-major rhs: the 4-wave kernel of _LP_256W4 on a 256 x 192 tile (each wave 128 x 96):
                                     grids on which the square tile leaves CUs idle (ABI 8; gemm_lp256w4.hip NJ = 3) */
    MI355_GEMM_ALGO_LP_192X192 = 13, /* ... and on a 192 x 192 tile (each wave 96 x 96; NJ = NI = 3) */
    MI355_GEMM_ALGO_LP_256M16 = 14,  /* bf16/f16, [N][K] rhs: the 256 x 256 tile on v_mfma_f32_16x16x32 (eight MFMAs per A fragment: the
                                     order that issues at 16 cycles and holds a higher clock on random operands; gemm_lp256m16.hip) */
    MI355_GEMM_ALGO_LP_256QM = 15    /* bf16/f16, [N][K] rhs, 16-bit C, full tiles: the persistent dripped-store kernel (_LP_256Q) on the 16x16x32
                                     MFMA order of _LP_256M16; bit-identical to _LP_256M16 (ABI 9; gemm_lp256qm.hip, config C5) */
};

int32_t mi355_gemm(mi355_ctx *ctx, mi355_stream stream, const mi355_gemm_desc *desc,
                   const void *a, const void *b, void *c);
/* cmma::execute(a, b, c, d) at tensor level (frontend/cmma.rs:1066-1110): D[b] = A[b] * B[b] + C[b].
 * C and D share the descriptor's ldc / stride_c / dtype_c; `c` may be the same buffer as `d` (in-place accumulate),
 * otherwise the two must not overlap.  A * B + C is formed in f32 -- the reference's accumulator fragment -- and rounded
 * once to dtype_c.  The f32 product goes through library-owned scratch (batch x M x N x 4 bytes per stream). */
int32_t mi355_gemm_add(mi355_ctx *ctx, mi355_stream stream, const mi355_gemm_desc *desc,
                       const void *a, const void *b, const void *c, void *d);
/* which kernel AUTO resolves to for a descriptor (for tests / logs).  Host-side only: the answer depends on the descriptor alone
 * (operands taken as 16-byte aligned), so `ctx` may be NULL -- the dispatcher's decisions are testable without a device. */
int32_t mi355_gemm_select(mi355_ctx *ctx, const mi355_gemm_desc *desc, int32_t *out_algo);
/* How a descriptor that resolves to the 256x256 kernel is cut when its last round of tiles is only partly filled (pure
 * function, no device): *out_splits == 1 -> one plain launch; otherwise rows (out_along_m = 1) or columns
 * [0, *out_main_extent) of C go to a plain launch and the remaining strip is computed with K split *out_splits ways
 * (one batched launch + the deterministic slab fold). */
int32_t mi355_gemm_tail_plan(const mi355_gemm_desc *desc, int32_t *out_along_m, int64_t *out_main_extent,
                             int32_t *out_splits);
/* (since ABI 6) How many K slices the 128x128 kernel's launcher cuts a descriptor into on a device with `compute_units` CUs
 * (0 = 256, the MI355X): 1 = one plain launch, otherwise that many f32 partial slabs + the deterministic fold.  A pure
 * function like the one above -- what it answers is only acted on when mi355_gemm_select says the 128x128 kernel runs. */
int32_t mi355_gemm_split_plan(const mi355_gemm_desc *desc, int32_t compute_units, int32_t *out_slices);
/* (round 4) How the few-rows x row-major-weight kernel (MI355_GEMM_ALGO_NNROWS, gemm_nnrows.hip) would cut a descriptor on a
 * device with `compute_units` CUs (0 = 256): bytes of a k-row per strip (1024 / 512 / 256), strips along N and K slices per
 * strip.  Its f32 summation order -- and with it the bits of the result -- is a function of these three alone.  All three 0:
 * the kernel does not take the descriptor.  Pure function, no device. */
int32_t mi355_gemm_strip_plan(const mi355_gemm_desc *desc, int32_t compute_units, int32_t *out_strip_bytes, int32_t *out_strips,
                              int32_t *out_slices);
/* Which operands mi355_gemm (AUTO) would first copy into library scratch, re-laid out K-contiguous -- the role of the
 * reference launchers' into_contiguous after matrix_batch_layout (crates/cubecl-std/src/tensor/matrix_batch_layout.rs:21-79)
 * -- before an MFMA kernel runs (pure function, no device; operands taken as 16-byte aligned).  Both 0: the kernels stage
 * the caller's layout directly and no scratch is touched -- true for row-major A with B either [N][K] or, for f32 and
 * (since ABI 5) bf16 / f16 tile-kernel shapes, row-major [K][N] as TensorHandle::new_contiguous lays a rhs out
 * (crates/cubecl-std/src/tensor/handle.rs:89). */
int32_t mi355_gemm_relayout_plan(const mi355_gemm_desc *desc, int32_t *out_relayout_a, int32_t *out_relayout_b);

/* Block-scaled matmul (MX formats): C[b] = (A[b] .* SA[b]) * (B[b] .* SB[b])^T, f32 accumulate -- the operation
 * `MmaDefinition::execute_scaled` defines per fragment (crates/cubecl-core/src/frontend/cmma.rs:795-840), semantics
 * pinned by test_cmma_scaled / test_cmma_scaled_fp4 (runtime_tests/cmma.rs:1476-1704):
 *     C[i][j] = sum_l A[i][l] * SA[i][l / block] * B[j][l] * SB[j][l / block]
 *   A is row-major [M][K] (lda >= K), B is stored [N][K] (ldb >= K) -- the tests' layouts (:1549-1551);
 *   SA [M][K/block], SB [N][K/block] are ue8m0 (ld_sa, ld_sb >= K/block, in scales);
 *   dtype_a / dtype_b: F8E4M3, F8E5M2 (may be mixed) or F4E2M1X2 (both; K, lda, ldb, strides still count ELEMENTS,
 *   a row of K elements is K/2 bytes, K and lda / ldb even); dtype_c: F32, BF16 or F16.
 * block == 32 with K a multiple of 128 (fp8) / 256 (fp4) and 16-byte aligned rows runs on
 * v_mfma_scale_f32_32x32x64_f8f6f4 inside the 256x256 kernel (the scales are first re-arranged per K-tile into
 * library-owned scratch); every other block size / shape runs on a bounds-checked scalar kernel that follows the
 * reference loop literally. */
typedef struct {
    int64_t m, n, k, batch;
    int64_t lda, ldb, ldc;
    int64_t ld_sa, ld_sb;
    int64_t stride_a, stride_b, stride_c, stride_sa, stride_sb; /* elements / scales between batch entries */
    int32_t dtype_a, dtype_b, dtype_c;
    int32_t block;  /* k-values per scale (scales_factor of the reference = k / block) */
    int32_t algo;   /* MI355_GEMM_ALGO_AUTO, _GENERIC or _LP_256W4 */
    int32_t reserved;
} mi355_gemm_scaled_desc;
int32_t mi355_gemm_scaled(mi355_ctx *ctx, mi355_stream stream, const mi355_gemm_scaled_desc *desc, const void *a,
                          const void *a_scales, const void *b, const void *b_scales, void *c);
int32_t mi355_gemm_scaled_select(mi355_ctx *ctx, const mi355_gemm_scaled_desc *desc, int32_t *out_algo);

/* =================================== Reductions ========================================== */

/* Bytes of device workspace the array-wide reductions need for `n` elements (one partial
 * record per workgroup + a ticket word); caller allocates, library never does.  `ctx` may be NULL:
 * the size is a property of the kernels (a client can plan the allocation before it reaches a server). */
int32_t mi355_reduce_workspace_bytes(mi355_ctx *ctx, uint64_t n, uint64_t *out_bytes);

/* Array-wide f32 sum: out[0] = sum_i in[i].  Semantics: examples/sum_things/src/lib.rs:6-19
 * (acc from 0.0f32) with the summation re-associated as a fixed tree (per-lane chunks ->
 * wave64 xor butterfly in the order of crates/cubecl-cpp/src/shared/plane.rs:60-70 -> waves ->
 * workgroups in index order); deterministic run to run.  n == 0 writes 0.0. */
int32_t mi355_reduce_sum_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, uint64_t n,
                             float *out, void *workspace, uint64_t workspace_bytes);
/* Array-wide argmax: out_idx[0] = lowest index of the maximum, out_val[0] = in[out_idx]
 * (bit copy).  NaN ranks above every number, first NaN wins; -0.0 == +0.0.  n == 0 writes
 * index 0 and -inf.  Indices are u64 (arrays may exceed 2^32 elements in 288 GB). */
int32_t mi355_argmax_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, uint64_t n,
                         float *out_val, uint64_t *out_idx, void *workspace,
                         uint64_t workspace_bytes);
/* Both in ONE pass over the data (1x the HBM bytes). Any of out_sum/out_val/out_idx non-NULL. */
int32_t mi355_sum_argmax_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, uint64_t n,
                             float *out_sum, float *out_val, uint64_t *out_idx, void *workspace,
                             uint64_t workspace_bytes);
/* The combine step of the multi-GPU argmax (the reference's ReduceOperation has only Sum / Mean,
 * crates/cubecl-runtime/src/server/base.rs:623-628 -- an API delta like MI355_REDUCE_MAX / MIN).  `records` (device) holds
 * `count` <= 64 records of 16 bytes, one per shard in rank order, as mi355_all_gather delivers them: {f32 value, u32
 * unused, u64 LOCAL index} = the bytes mi355_sum_argmax_f32 writes at out_val / out_idx when those are 8 bytes apart.
 * index_base (HOST array of `count` u64, NULL = zeros) is the first element of each shard; a record whose index is
 * 2^64-1 is an empty shard and is ignored.  Same rule as inside one device: larger value wins, NaN above every number,
 * -0.0 == +0.0, equal values keep the LOWEST global index; no record at all gives -inf / 0.  One 64-lane launch on
 * `stream`: queue it after mi355_sync_collective and the result is in device memory on every rank, stream-ordered. */
int32_t mi355_argmax_combine_f32(mi355_ctx *ctx, mi355_stream stream, const void *records, uint32_t count,
                                 const uint64_t *index_base, float *out_val, uint64_t *out_idx);
/* The whole exchange of the sharded sum + argmax (config C4) behind ONE collective (ABI 8): the
 * record's second word carries the shard's partial sum -- {f32 value, f32 partial sum, u64 LOCAL index} = the bytes
 * mi355_sum_argmax_f32 writes with out_val = rec, out_sum = rec + 4, out_idx = rec + 8 -- so mi355_all_gather moves sums
 * and argmax candidates together, and this launch folds both: out_sum = the partial sums added in RANK ORDER starting from
 * +0.0 (a fixed tree: the same bits on every rank and every run, where the reference's all_reduce(Sum),
 * crates/cubecl-cuda/src/compute/server.rs:705-780, leaves the order to the collective's algorithm), out_val / out_idx as
 * mi355_argmax_combine_f32.  An empty shard passes index 2^64-1 and sum +0.0.  Any of the outputs may be NULL. */
int32_t mi355_sum_argmax_combine_f32(mi355_ctx *ctx, mi355_stream stream, const void *records, uint32_t count,
                                     const uint64_t *index_base, float *out_sum, float *out_val, uint64_t *out_idx);
/* The same three reductions for f32, bf16 or f16 inputs (`dtype`): elements are widened to f32 on load (exact), sums and
 * comparisons run in f32, the outputs stay {f32 sum, f32 value of the winning element, u64 index}.  Same rules (lowest
 * index wins ties, NaN ranks highest, -0 == +0), same single-launch deterministic tree; 16-bit inputs move half the
 * bytes per element.  Input aligned to its element size. */
int32_t mi355_reduce_sum(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n,
                         float *out, void *workspace, uint64_t workspace_bytes);
int32_t mi355_argmax(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n,
                     float *out_val, uint64_t *out_idx, void *workspace, uint64_t workspace_bytes);
int32_t mi355_sum_argmax(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n,
                         float *out_sum, float *out_val, uint64_t *out_idx, void *workspace,
                         uint64_t workspace_bytes);
/* Sum over the last axis of a [rows, cols] view with row stride `row_stride` elements: the
 * book's reduce_matrix (cubecl-book/src/getting-started/src/bin/v1-cpu.rs:7-15). */
int32_t mi355_reduce_last_axis_sum_f32(mi355_ctx *ctx, mi355_stream stream, const float *in,
                                       float *out, uint64_t rows, uint64_t cols,
                                       uint64_t row_stride);
int32_t mi355_reduce_last_axis_argmax_f32(mi355_ctx *ctx, mi355_stream stream, const float *in,
                                          uint32_t *out_idx, uint64_t rows, uint64_t cols,
                                          uint64_t row_stride);
/* the last-axis reductions for f32 / bf16 / f16 rows (`dtype`; widened to f32 on load, f32 sums / u32 indices out;
 * row_stride in elements) */
int32_t mi355_reduce_last_axis_sum(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype,
                                   float *out, uint64_t rows, uint64_t cols, uint64_t row_stride);
int32_t mi355_reduce_last_axis_argmax(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype,
                                      uint32_t *out_idx, uint64_t rows, uint64_t cols, uint64_t row_stride);
/* Sum / argmax over ANY one axis of a contiguous tensor viewed as [outer][reduce][inner] (inner == 1 is the
 * last-axis case above): out is [outer][inner].  The book's reduce_dim generalisation
 * (cubecl-book/src/getting-started/src/bin/v7-gpu.rs:59-77 reduces the last axis of a 3-D tensor; cubek's
 * reduce takes the axis as a parameter).  argmax: lowest index of the maximum along the axis, same NaN / -0
 * rules as mi355_argmax_f32. */
int32_t mi355_reduce_axis_sum_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, float *out,
                                  uint64_t outer, uint64_t reduce, uint64_t inner);
int32_t mi355_reduce_axis_argmax_f32(mi355_ctx *ctx, mi355_stream stream, const float *in,
                                     uint32_t *out_idx, uint64_t outer, uint64_t reduce, uint64_t inner);
/* ... and for f32 / bf16 / f16 input (`dtype`) */
int32_t mi355_reduce_axis_sum(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, float *out,
                              uint64_t outer, uint64_t reduce, uint64_t inner);
int32_t mi355_reduce_axis_argmax(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype,
                                 uint32_t *out_idx, uint64_t outer, uint64_t reduce, uint64_t inner);
/* Every reduce operation over f32 / bf16 / f16 input (widened to f32 on load; f32 arithmetic, f32 values / integer indices out),
 * array-wide and over any one axis of a contiguous [outer][reduce][inner] view -- the surface of cubek-reduce that the book's
 * reduce_dim stands for (cubecl-book/src/getting-started/src/bin/v7-gpu.rs:59-77; plane-level forms
 * crates/cubecl-core/src/frontend/plane.rs:218 sum, :285 prod, :352 max, :370 min; ReduceOperation::Mean
 * crates/cubecl-runtime/src/server/base.rs:623-628).
 *   SUM   as mi355_reduce_sum (fixed tree, deterministic)       MEAN  that sum / (f32)count  (count 0: 0 array-wide, NaN per axis)
 *   PROD  f32 product in the same tree shape (identity 1)
 *   MAX / MIN  IEEE maximum / minimum with -0 < +0; NaN if ANY element is NaN (numpy's max / min); empty: -inf / +inf
 *   ARGMAX     as mi355_argmax: lowest index of the maximum, -0 == +0, NaN ranks above every number, first NaN wins
 *   ARGMIN     its mirror image: lowest index of the minimum, -0 == +0, NaN ranks first, first NaN wins (numpy's argmin)
 * mi355_reduce / mi355_argreduce take the workspace of mi355_reduce_workspace_bytes (one launch, records folded by the last
 * workgroup); the axis forms need none.  argreduce: out_val = bit copy of the winning element; empty: index 0, -inf / +inf. */
int32_t mi355_reduce(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n, int32_t op, float *out,
                     void *workspace, uint64_t workspace_bytes);
int32_t mi355_argreduce(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, uint64_t n, int32_t op, float *out_val,
                        uint64_t *out_idx, void *workspace, uint64_t workspace_bytes);
int32_t mi355_reduce_axis(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, int32_t op, float *out,
                          uint64_t outer, uint64_t reduce, uint64_t inner);
int32_t mi355_argreduce_axis(mi355_ctx *ctx, mi355_stream stream, const void *in, int32_t dtype, int32_t op, uint32_t *out_idx,
                             uint64_t outer, uint64_t reduce, uint64_t inner);
/* plane_sum & friends for one 64-lane plane per 64 inputs (frontend/plane.rs:218-240): every
 * lane receives the butterfly result over the first `active` lanes (power of two <= 64),
 * matching plane_dim_checked = min(PLANE_DIM, CUBE_DIM) (shared/plane.rs:55-58).
 * op = MI355_REDUCE_SUM / MAX / MIN / PROD (100 = PROD's round-1 code), or a scan: MI355_PLANE_INCLUSIVE_SUM / _EXCLUSIVE_SUM /
 * _INCLUSIVE_PROD / _EXCLUSIVE_PROD (Hillis-Steele over shuffle_up, shared/plane.rs:72-97; lane 0 of an exclusive scan
 * receives 0 / 1). */
int32_t mi355_plane_reduce_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, float *out,
                               uint64_t n, uint32_t active, int32_t op);
/* The remaining plane intrinsics at tensor level (crates/cubecl-core/src/frontend/plane.rs:62-216, :388-440; HIP lowering
 * crates/cubecl-cpp/src/hip/plane.rs:19-64, elect shared/plane.rs:170-174; known answers runtime_tests/plane.rs:527-850): one
 * plane of `plane` lanes (32 or 64: CubeDim::new_1d(plane) on this wave64 device) per `plane` consecutive inputs.
 *   ALL / ANY        out[i] = 1.0 if every / any lane of the plane holds a non-zero input, else 0.0
 *   ELECT            out[i] = 1.0 for the lowest active lane of the plane, 0.0 for the others
 *   BROADCAST / SHUFFLE (arg = source lane), SHUFFLE_XOR (arg = mask), SHUFFLE_UP / _DOWN (arg = delta): out[i] = the value
 *                    of the source lane; a source outside the plane leaves the lane's own value (HIP's __shfl_* rule)
 *   BALLOT           out = 4 x u32 per plane (16 bytes each): the mask of lanes with a non-zero input, words 2 and 3 zero */
enum { MI355_PLANE_ALL = 200, MI355_PLANE_ANY = 201, MI355_PLANE_ELECT = 202, MI355_PLANE_BROADCAST = 203, MI355_PLANE_SHUFFLE = 204,
       MI355_PLANE_SHUFFLE_XOR = 205, MI355_PLANE_SHUFFLE_UP = 206, MI355_PLANE_SHUFFLE_DOWN = 207, MI355_PLANE_BALLOT = 208 };
int32_t mi355_plane_op_f32(mi355_ctx *ctx, mi355_stream stream, const float *in, void *out, uint64_t n, uint32_t plane,
                           int32_t op, uint32_t arg);

/* tensor::identity::launch (crates/cubecl-std/src/tensor/identity.rs:36-84): writes the dim x dim identity matrix
 * (1 on the diagonal in `dtype`, 0 elsewhere) into rows `ld` elements apart (ld >= dim; the padding of a pitched
 * allocation is left alone).  dtype: any float, integer or fp8 type of this header. */
int32_t mi355_fill_identity(mi355_ctx *ctx, mi355_stream stream, void *out, int32_t dtype, uint64_t dim, uint64_t ld);

/* =================================== Strided copies (into_contiguous) =================================== */

/* A tensor view: shape and strides in ELEMENTS, outermost axis first (TensorBinding's shape / strides,
 * crates/cubecl-std/src/tensor/handle.rs).  rank 1..MI355_MAX_RANK; a stride of 0 broadcasts (input side only). */
#define MI355_MAX_RANK 8
typedef struct mi355_tensor_layout {
    int32_t rank;
    int32_t reserved;
    int64_t shape[MI355_MAX_RANK];
    int64_t strides[MI355_MAX_RANK];
} mi355_tensor_layout;

/* copy_into / into_contiguous / into_contiguous_pitched (crates/cubecl-std/src/tensor/contiguous/launch.rs:5-56,
 * base.rs:295-389): element q of the input view in row-major (linear) order goes to element q of the output view
 * in linear order.  The two views hold the same number of elements; they may differ in rank and shape (the
 * reference's rank-mismatch case, tests/tensor/into_contiguous.rs:139-185) and the OUTPUT may be strided too
 * (a pitched allocation, or a permuted destination).  elem_size 1, 2, 4 or 8 bytes: the copy moves bits.
 * HBM-bound: 2 x elements x elem_size bytes per call.  Paths (chosen from the layouts):
 *   FLAT       both views collapse to one contiguous run                       -> 16-byte vector copy
 *   ROWS       both views are contiguous along the innermost (collapsed) axis  -> vector copy, rows located by
 *                                                                                 multiply-shift division
 *   TRANSPOSE  the input is contiguous along one axis, the output along another (both >= 16 long; 1/2/4-byte
 *              elements) -> 256-byte x 256-byte tiles through bank-swizzled LDS, 16-byte accesses on both sides
 *   GENERIC    anything else over a common refinement of the two shapes: one element per access
 *   TWO_SIDED  two strided views whose axis boundaries do not nest (a strided [2,3] into a strided [3,2]; a
 *              contiguous side refines against anything): each side decomposes the linear index by its own shape */
enum {
    MI355_COPY_PATH_FLAT = 0,
    MI355_COPY_PATH_ROWS = 1,
    MI355_COPY_PATH_TRANSPOSE = 2,
    MI355_COPY_PATH_GENERIC = 3,
    MI355_COPY_PATH_TWO_SIDED = 4
};
int32_t mi355_copy_strided(mi355_ctx *ctx, mi355_stream stream, const void *in, const mi355_tensor_layout *in_layout,
                           void *out, const mi355_tensor_layout *out_layout, int32_t elem_size);
/* Host-only: which path (and how many bytes per access) mi355_copy_strided takes for these arguments; no device
 * needed.  Returns MI355_E_INVALID_ARGUMENT for what mi355_copy_strided would reject. */
int32_t mi355_copy_strided_plan(const void *in, const mi355_tensor_layout *in_layout, const void *out,
                                const mi355_tensor_layout *out_layout, int32_t elem_size, int32_t *path,
                                int32_t *access_bytes);

/* into_contiguous_packed (base.rs:254-293, :391-472): re-packs a tensor of sub-word values -- `packing` values of
 * word_bits / packing bits in each storage word, packed along axis `rank - 1 - packed_dim` of the input (packed_dim
 * counts from the innermost axis, as the reference's argument does) -- so that the values are packed along the
 * innermost axis of the output.  `shape` is the logical (unpacked) shape, rank = in_storage->rank;
 * in_storage / out_storage describe the word tensors (strides in words); the output holds
 * prod(shape[:-1]) x ceil(shape[-1] / packing) words.  word_size 4 (u32) or 1 (u8).  Output word w takes, in bit
 * slot n, the logical element with linear index w * packing + n (base.rs:170-207). */
int32_t mi355_copy_packed(mi355_ctx *ctx, mi355_stream stream, const void *in, const mi355_tensor_layout *in_storage,
                          void *out, const mi355_tensor_layout *out_storage, const int64_t *shape,
                          int32_t packed_dim, int32_t packing, int32_t word_size);

/* =================================== Throughput probes =================================== */

/* memory_read_throughput (crates/cubecl-std/src/throughput/runners/memory_read.rs:69-154):
 * streaming read of `bytes` with a guarded store; returns after enqueueing `iters` passes. */
int32_t mi355_probe_memory_read(mi355_ctx *ctx, mi355_stream stream, const void *buf,
                                uint64_t bytes, uint32_t iters, void *sink);
/* compute_cmma_throughput (runners/compute_cmma.rs:48-91): `iters` dependent MFMAs per wave on
 * constant fragments; *out_ops receives the op count of one call (2*m*n*k per MFMA). */
int32_t mi355_probe_mfma(mi355_ctx *ctx, mi355_stream stream, int32_t dtype_ab, uint32_t iters,
                         void *sink, uint64_t *out_ops);

/* memory_direct_throughput (runners/memory_direct.rs:55-117): one streaming copy of `bytes`
 * (read + written bytes both count: 2 x bytes of traffic per call). */
int32_t mi355_probe_memory_copy(mi355_ctx *ctx, mi355_stream stream, const void *src, void *dst,
                                uint64_t bytes);
/* memory_write_throughput (runners/memory_write.rs:65-139): write-only stream of `bytes`. */
int32_t mi355_probe_memory_write(mi355_ctx *ctx, mi355_stream stream, void *dst, uint64_t bytes);
/* compute_direct_throughput (runners/compute_direct.rs:50-103): four independent f32 fma chains
 * of 4-wide vectors per lane, `iters` steps; *out_ops = flops of one call. */
int32_t mi355_probe_compute_direct(mi355_ctx *ctx, mi355_stream stream, uint32_t iters, void *sink,
                                   uint64_t *out_ops);
/* launch_overhead (runners/launch_overhead.rs:43-51): enqueues `launches` empty kernels. */
int32_t mi355_probe_launch_overhead(mi355_ctx *ctx, mi355_stream stream, uint32_t launches,
                                    void *sink);
/* The same issue loop on the GEMM kernels' 4x4 accumulator shape (bf16), register-resident, with
 * mode 0 = all-ones operands, mode 1 = uniform[-1,1) operands rotating every iteration, mode 2 = mode 1 on
 * the fp8 instruction (v_mfma_f32_32x32x64_f8f6f4, e4m3 operands), modes 3 / 4 = the block-scaled fp4 instruction
 * (v_mfma_scale_f32_32x32x64_f8f6f4) on all-ones / random nibbles and scales.  Mode 1 is
 * the matrix-pipe ceiling for the benchmark's operand distribution once DVFS has clocked the chip
 * down to its power budget; bench.py reports it beside the spec peak (not a reference probe). */
int32_t mi355_probe_mfma_data(mi355_ctx *ctx, mi355_stream stream, int32_t mode, uint32_t iters,
                              void *sink, uint64_t *out_ops);
/* Samples {shader-clock ticks (s_memtime), constant 100 MHz ticks (s_memrealtime)} per CU into
 * dev_out (device memory, MI355_CLOCK_PROBE_BYTES, zero it first): slot = XCC_ID * 64 + HW_ID[13:8], two
 * uint64 per slot.  The shader counter is local to a CU, so two samples bracketing a region give the clock
 * a CU sustained over it only when paired slot by slot (timing_method Device,
 * crates/cubecl-hip/src/runtime.rs:198 analogue). */
#define MI355_CLOCK_PROBE_BYTES 8192
int32_t mi355_probe_clock(mi355_ctx *ctx, mi355_stream stream, uint64_t *dev_out);

/* =================================== Collectives (RCCL over xGMI) ======================== */

#define MI355_UNIQUE_ID_BYTES 128
/* get_nccl_comm_id (crates/cubecl-cuda/src/compute/communication.rs:14-25): rank 0 creates the
 * id, every rank receives the same bytes out of band. */
int32_t mi355_comm_unique_id(uint8_t id[MI355_UNIQUE_ID_BYTES]);
/* ServerCommunication::comm_init (crates/cubecl-cuda/src/compute/server.rs:669-703): rank =
 * position of this device in the sorted id list. */
int32_t mi355_comm_init(mi355_ctx *ctx, const uint8_t id[MI355_UNIQUE_ID_BYTES], int32_t rank,
                        int32_t world_size, mi355_comm **out_comm);
int32_t mi355_comm_destroy(mi355_ctx *ctx, mi355_comm *comm);
/* ServerCommunication::all_reduce (server.rs:705-780): fences the compute stream into the comm
 * stream, then ncclAllReduce on the comm stream.  count in elements of `dtype`. */
int32_t mi355_all_reduce(mi355_ctx *ctx, mi355_comm *comm, mi355_stream compute_stream,
                         const void *src, void *dst, uint64_t count, int32_t dtype, int32_t op);
/* all-gather of `count` elements per rank (argmax pairs) */
int32_t mi355_all_gather(mi355_ctx *ctx, mi355_comm *comm, mi355_stream compute_stream,
                         const void *src, void *dst, uint64_t count, int32_t dtype);
/* ServerCommunication::send / recv (server.rs:799-926) */
int32_t mi355_send(mi355_ctx *ctx, mi355_comm *comm, mi355_stream compute_stream,
                   const void *src, uint64_t count, int32_t dtype, int32_t peer);
int32_t mi355_recv(mi355_ctx *ctx, mi355_comm *comm, mi355_stream compute_stream, void *dst,
                   uint64_t count, int32_t dtype, int32_t peer);
/* ServerCommunication::sync_collective (server.rs:782-797): the compute stream waits for the
 * comm stream. */
int32_t mi355_sync_collective(mi355_ctx *ctx, mi355_stream compute_stream);
/* (ABI 9) The exchange of the sharded sum + argmax (mi355_sum_argmax_combine_f32 above) as ONE call -- what the Rust server issues per step (rust/cubecl-mi355/src/comm.rs sum_argmax_exchange) and what
 * keeps two host round trips through the binding out of a 12 us exchange: mi355_all_gather of this rank's 16-byte `record` (as 2 x u64)
 * into `gathered` (16 bytes x the communicator's world size, rank order), the comm -> compute fence of mi355_sync_collective, and
 * mi355_sum_argmax_combine_f32 over `gathered`, all queued on `stream`.  `index_base`: HOST array of world-size u64 (NULL = zeros), read
 * before the call returns.  record / gathered / outputs are device memory; any output may be NULL.  Errors as the three calls'. */
int32_t mi355_sum_argmax_exchange(mi355_ctx *ctx, mi355_comm *comm, mi355_stream stream, const void *record, void *gathered,
                                  const uint64_t *index_base, float *out_sum, float *out_val, uint64_t *out_idx);

/* =================================== Graph capture ======================================= */

/* ComputeServer::begin_capture / end_capture / replay / graph_destroy (server/base.rs:472-532; HIP
 * implementation crates/cubecl-hip/src/compute/server.rs:288-521): records what is enqueued on the
 * stream between begin and end into a hipGraph and replays it with one launch -- for launch-bound
 * sequences (an empty kernel costs ~2.7 us of host time here).  Run the sequence once before capturing
 * (graph_prepare + warm-up in the reference, base.rs:453-470) so that lazily created library scratch
 * exists; nothing the library enqueues synchronises the host or allocates after that. */
typedef struct mi355_graph mi355_graph;
int32_t mi355_graph_begin_capture(mi355_ctx *ctx, mi355_stream stream);
int32_t mi355_graph_end_capture(mi355_ctx *ctx, mi355_stream stream, mi355_graph **out_graph);
int32_t mi355_graph_replay(mi355_ctx *ctx, mi355_stream stream, mi355_graph *graph);
int32_t mi355_graph_destroy(mi355_ctx *ctx, mi355_graph *graph);

/* =================================== Profiling =========================================== */

/* ComputeServer::start_profile / end_profile (server/base.rs:590-597) with device timing:
 * brackets the stream with two events; stop returns the elapsed GPU time in nanoseconds. */
int32_t mi355_profile_start(mi355_ctx *ctx, mi355_stream stream, uint64_t *out_token);
int32_t mi355_profile_stop(mi355_ctx *ctx, mi355_stream stream, uint64_t token,
                           uint64_t *out_nanos);

#ifdef __cplusplus
}
#endif
#endif /* MI355CUBE_H */
