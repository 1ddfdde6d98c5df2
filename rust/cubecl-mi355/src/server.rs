//! `Mi355Server`: `ComputeServer` (cubecl-runtime/src/server/base.rs `pub trait ComputeServer`) over the C ABI.
//!
//! Memory is the reference's: every lane owns a `MemoryManagement` pool (`reserve` -> `bind` -> `get_resource`) on top
//! of the storages in [`crate::storage`].  Ordering is the reference's too: every call starts with
//! `MultiStream::resolve(stream_id, bindings, ..)`, which inserts the cross-lane waits, and then works on the resolved
//! lane's `mi355_stream`.  What is this crate's own is the last step of each operation: a call into `libmi355cube.so`.
use crate::{
    comm,
    compiler::{Mi355Compiler, Mi355Kernel, NativeOp, ReduceKind},
    error,
    ffi::*,
    lane::LaneBackend,
    program::{self, Program},
    storage::{DeviceSlice, DeviceStorage, PinnedBytes},
};
use cubecl_common::{bytes::Bytes, device::DeviceId, profile::ProfileDuration};
use cubecl_cpp::shared::CompilationOptions;
use cubecl_environment::{backtrace::BackTrace, future::DynFut, stream::StreamId};
use cubecl_ir::MemoryDeviceProperties;
use cubecl_runtime::{
    allocator::PitchedMemoryLayoutPolicy,
    compiler::CubeTask,
    config::{CubeClRuntimeConfig, RuntimeConfig},
    dry_run::LaunchMode,
    id::{GraphId, KernelId},
    kernel::KernelMetadata,
    logging::ServerLogger,
    memory_management::{
        InstallMemoryPoolsError, ManagedMemoryHandle, MemoryAllocationMode, MemoryConfiguration, MemoryHandle, MemoryReport,
        MemoryUsage,
    },
    server::{
        BufferBinding, CommunicationId, ComputeServer, CopyDescriptor, CubeCount, Handle, IoError, KernelArguments,
        KernelResource, LaunchError, ProfileError, ProfilingToken, ServerError, ServerUtilities,
    },
    storage::{ComputeStorage, ManagedResource},
    stream::{MultiStream, ResolvedStreams},
    timestamp_profiler::TimestampProfiler,
    validation::{validate_cube_dim, validate_units},
};
use cubecl_zspace::{Shape, Strides, striding::has_pitched_row_major_strides};
use std::{collections::HashMap, sync::Arc};

/// Uploads above this size skip the pinned bounce buffer when the caller's bytes are not already pinned.
const STAGE_LIMIT: usize = 100 << 20;
/// Uploads above this size release their staging buffer at once instead of waiting for the queue to fill.
const EAGER_RELEASE: usize = 10 << 20;

/// What a `KernelId` resolved to the first time it was launched.
#[derive(Debug, Clone, Copy)]
enum Loaded {
    Native(NativeOp),
    Jit(Program),
}

/// A captured `mi355_graph` and the pool slices it pins for as long as it can be replayed.
#[derive(Debug)]
struct CapturedGraph {
    raw: *mut mi355_graph,
    _pinned: Vec<ManagedMemoryHandle>,
}

#[derive(Debug)]
pub struct Mi355Server {
    pub(crate) ctx: *mut mi355_ctx,
    pub(crate) device_id: DeviceId,
    pub(crate) lanes: MultiStream<LaneBackend>,
    compiler: Mi355Compiler,
    options: CompilationOptions,
    kernels: HashMap<KernelId, Loaded>,
    timestamps: TimestampProfiler,
    graphs: HashMap<GraphId, CapturedGraph>,
    pub(crate) communicators: HashMap<CommunicationId, *mut mi355_comm>,
    pub(crate) utilities: Arc<ServerUtilities<Self>>,
    /// Declared last on purpose: fields drop in order, and the lanes above still need the context to destroy their streams.
    _context: ContextGuard,
}
unsafe impl Send for Mi355Server {}

/// Destroys the `mi355_ctx` (its modules, its cached pages, its streams) when the server goes away.
#[derive(Debug)]
struct ContextGuard(*mut mi355_ctx);

impl Drop for ContextGuard {
    fn drop(&mut self) {
        unsafe { mi355_ctx_destroy(self.0) };
    }
}

/// How a call treats errors earlier asynchronous launches left on the lane.
#[derive(Clone, Copy, PartialEq, Eq)]
pub(crate) enum Pending {
    /// Report them now (`ServerUnhealthy`) and refuse to go on: reads, syncs, flushes.
    Surface,
    /// Leave them queued: launches and writes are fire-and-forget and must not lose an error to a later call.
    Keep,
}

/// One resolved call: the lanes `MultiStream` aligned for it.  Dropping it hands cross-lane pins to the GC thread.
pub(crate) struct Pass<'a> {
    ctx: *mut mi355_ctx,
    pub(crate) lanes: ResolvedStreams<'a, LaneBackend>,
}

impl<'a> Pass<'a> {
    pub(crate) fn sys(&mut self) -> mi355_stream {
        self.lanes.current().sys
    }

    /// Device pointer + length behind a binding, looked up in the pool of the lane that created it.
    pub(crate) fn slice(&mut self, binding: BufferBinding) -> Result<DeviceSlice, IoError> {
        self.lanes.get(&binding.stream).memory_management_gpu.get_resource(binding.memory, binding.offset_start, binding.offset_end)
    }

    fn reserve(&mut self, size: u64) -> Result<ManagedMemoryHandle, IoError> {
        match self.lanes.current().memory_management_gpu.reserve(size) {
            Err(IoError::OutOfMemory { .. }) | Err(IoError::PoolCapacityExceeded { .. }) => {
                // give cached pages back to the driver once, then ask again
                self.cleanup();
                self.lanes.current().memory_management_gpu.reserve(size)
            }
            other => other,
        }
    }

    /// Ties the freshly reserved slice to the handle the client already holds, stamped with this call's cursor so that
    /// another lane can tell whether it has waited for it.
    fn bind(&mut self, reserved: ManagedMemoryHandle, handle: ManagedMemoryHandle) -> Result<(), IoError> {
        let cursor = self.lanes.cursor;
        self.lanes.current().memory_management_gpu.bind(reserved, handle, cursor)
    }

    fn empty(&mut self, size: u64) -> Result<Handle, IoError> {
        let handle = Handle::new(self.lanes.current, size);
        let reserved = self.reserve(size)?;
        self.bind(reserved, handle.memory.clone())?;
        Ok(handle)
    }

    /// Page-locked bytes from the pinned pool of `origin` (or of the current lane); `None` when that pool is exhausted.
    fn pinned(&mut self, size: usize, origin: Option<StreamId>) -> Option<Bytes> {
        let lane = match origin {
            Some(id) => self.lanes.get(&id),
            None => self.lanes.current(),
        };
        let reserved = lane.memory_management_cpu.reserve(size as u64).ok()?;
        let binding = MemoryHandle::binding(reserved);
        let slice = lane.memory_management_cpu.get_resource(binding.clone(), None, None).ok()?;
        Some(unsafe { Bytes::from_controller(Box::new(PinnedBytes::new(binding, slice)), size) })
    }

    fn host_bytes(&mut self, size: usize, want_pinned: bool) -> Bytes {
        if !want_pinned && size > STAGE_LIMIT {
            return Bytes::from_bytes_vec(vec![0; size]);
        }
        self.pinned(size, None).unwrap_or_else(|| Bytes::from_bytes_vec(vec![0; size]))
    }

    fn cleanup(&mut self) {
        let lane = self.lanes.current();
        lane.release_staging();
        lane.release_staging();
        if !lane.capturing.is_recording() {
            lane.info_cache.clear_unpinned();
        }
        lane.memory_management_gpu.cleanup(true);
        lane.memory_management_cpu.cleanup(true);
    }

    /// Host -> device, contiguous or row-pitched (`mi355_write` / `mi355_write_2d`), asynchronous on the lane's stream.
    fn upload(&mut self, descriptor: CopyDescriptor, data: Bytes) -> Result<(), IoError> {
        let CopyDescriptor { handle, shape, strides, elem_size } = descriptor;
        if !has_pitched_row_major_strides(&shape, &strides) {
            return Err(IoError::UnsupportedStrides { backtrace: BackTrace::capture() });
        }
        let target = self.slice(handle)?;
        let size = data.len();
        if size == 0 {
            return Ok(());
        }
        let already_pinned = matches!(data.property(), cubecl_common::bytes::AllocationProperty::Pinned);
        let data = if !already_pinned && size < STAGE_LIMIT {
            match self.pinned(size, None) {
                Some(mut bounce) => {
                    data.copy_into(&mut bounce);
                    bounce
                }
                None => data,
            }
        } else {
            data
        };
        let lane = self.lanes.current();
        let rc = match pitch_of(&shape, &strides, elem_size) {
            None => unsafe { mi355_write(self.ctx, lane.sys, target.ptr, data.as_ptr() as *const core::ffi::c_void, size as u64) },
            Some((width, rows, pitch)) => unsafe {
                mi355_write_2d(self.ctx, lane.sys, target.ptr, pitch, data.as_ptr() as *const core::ffi::c_void, width, width, rows)
            },
        };
        if rc != MI355_OK {
            return Err(error::io(self.ctx, rc, size as u64));
        }
        // the copy reads `data` after this call returns: park it until a fence behind the copy has been seen
        lane.drop_queue.push(data);
        if size > EAGER_RELEASE || lane.drop_queue.should_flush() {
            lane.release_staging();
        }
        Ok(())
    }

    /// Device -> host into fresh (pinned when possible) bytes; the copy is only enqueued, see `read`.
    fn download(&mut self, descriptor: CopyDescriptor) -> Result<Bytes, IoError> {
        let CopyDescriptor { handle, shape, strides, elem_size } = descriptor;
        if !has_pitched_row_major_strides(&shape, &strides) {
            return Err(IoError::UnsupportedStrides { backtrace: BackTrace::capture() });
        }
        let size = shape.iter().product::<usize>() * elem_size;
        let mut bytes = self.host_bytes(size, true);
        if size == 0 {
            return Ok(bytes);
        }
        let source = self.slice(handle)?;
        let sys = self.sys();
        let rc = match pitch_of(&shape, &strides, elem_size) {
            None => unsafe { mi355_read_async(self.ctx, sys, bytes.as_mut_ptr() as *mut core::ffi::c_void, source.ptr, size as u64) },
            Some((width, rows, pitch)) => unsafe {
                mi355_read_2d(self.ctx, sys, bytes.as_mut_ptr() as *mut core::ffi::c_void, width, source.ptr, pitch, width, rows)
            },
        };
        match rc {
            MI355_OK => Ok(bytes),
            rc => Err(error::io(self.ctx, rc, size as u64)),
        }
    }

    /// The metadata words of a launch as a device buffer: staged through the pinned pool, copied on the lane's stream,
    /// and remembered per lane so that a steady-state loop (and a captured graph) re-uses one buffer per distinct payload.
    fn info_buffer(&mut self, words: Vec<u64>) -> Result<Handle, ServerError> {
        let size = core::mem::size_of_val(words.as_slice());
        let lane = self.lanes.current();
        let mode = lane.capturing.cache_mode();
        lane.info_cache.mode(mode);
        let cacheable = lane.info_cache.should_cache(size);
        if cacheable && let Some(hit) = lane.info_cache.get(&words) {
            return Ok(hit);
        }
        let raw: &[u8] = bytemuck::cast_slice(&words);
        let mut staging = self.pinned(raw.len(), None).ok_or_else(|| IoError::Unknown {
            description: "the pinned pool could not stage the launch metadata".into(),
            backtrace: BackTrace::capture(),
        })?;
        staging.copy_from_slice(raw);
        let handle = self.empty(raw.len() as u64)?;
        self.upload(CopyDescriptor::new(handle.clone().binding(), [raw.len()].into(), [1].into(), 1), staging)?;
        if cacheable {
            self.lanes.current().info_cache.insert(words, handle.clone());
        }
        Ok(handle)
    }
}

/// `(row bytes, rows, pitch bytes)` when the innermost rows are padded, `None` for a contiguous buffer.
fn pitch_of(shape: &Shape, strides: &Strides, elem_size: usize) -> Option<(u64, u64, u64)> {
    let rank = shape.len();
    if rank < 2 || strides[rank - 2] == shape[rank - 1] {
        return None;
    }
    let rows: usize = shape.iter().rev().skip(1).product();
    Some(((shape[rank - 1] * elem_size) as u64, rows as u64, (strides[rank - 2] * elem_size) as u64))
}

impl Mi355Server {
    pub(crate) fn new(ctx: *mut mi355_ctx, device_id: DeviceId, mem_props: MemoryDeviceProperties, mem_config: MemoryConfiguration,
                      options: CompilationOptions, utilities: ServerUtilities<Self>) -> Self {
        let max_streams = CubeClRuntimeConfig::get().streaming.max_streams;
        let logger = utilities.logger.clone();
        Self {
            ctx,
            device_id,
            lanes: MultiStream::new(logger.clone(), LaneBackend::new(ctx, mem_props, mem_config, logger), max_streams),
            compiler: Mi355Compiler::default(),
            options,
            kernels: HashMap::new(),
            timestamps: TimestampProfiler::default(),
            graphs: HashMap::new(),
            communicators: HashMap::new(),
            utilities: Arc::new(utilities),
            _context: ContextGuard(ctx),
        }
    }

    pub(crate) fn pass<'a>(&mut self, stream_id: StreamId, bindings: impl Iterator<Item = &'a BufferBinding>, pending: Pending) -> Result<Pass<'_>, ServerError> {
        if pending == Pending::Surface {
            let errors = self.take_errors(stream_id);
            if !errors.is_empty() {
                return Err(ServerError::ServerUnhealthy { errors, backtrace: BackTrace::capture() });
            }
        }
        let lanes = self.lanes.resolve(stream_id, bindings, pending == Pending::Surface)?;
        Ok(Pass { ctx: self.ctx, lanes })
    }

    pub(crate) fn pass_alone(&mut self, stream_id: StreamId, pending: Pending) -> Result<Pass<'_>, ServerError> {
        self.pass(stream_id, [].into_iter(), pending)
    }

    /// Errors queued on the lane by earlier launches, plus whatever the library queued itself (`mi355_error_pop`).
    fn take_errors(&mut self, stream_id: StreamId) -> Vec<ServerError> {
        let Ok(mut lanes) = self.lanes.resolve(stream_id, [].into_iter(), false) else {
            return Vec::new();
        };
        let lane = lanes.current();
        let mut errors = core::mem::take(&mut lane.errors);
        let mut queued = 0i32;
        if unsafe { mi355_error_count(self.ctx, &mut queued) } == MI355_OK && queued > 0 {
            if let Err(ServerError::ServerUnhealthy { errors: native, .. }) = error::check(self.ctx, MI355_E_SERVER_UNHEALTHY) {
                errors.extend(native);
            }
        }
        if !errors.is_empty() {
            self.timestamps.error(ProfileError::Unknown { reason: format!("{errors:?}"), backtrace: BackTrace::capture() });
            lane.memory_management_gpu.cleanup(false);
        }
        errors
    }

    fn record(&mut self, stream_id: StreamId, err: ServerError) {
        match self.lanes.resolve(stream_id, [].into_iter(), false) {
            Ok(mut lanes) => lanes.current().errors.push(err),
            Err(err) => unreachable!("{err}"),
        }
    }

    /// First launch of a kernel id: native descriptor, or HIP C++ -> hiprtc -> module.
    fn load(&mut self, id: &KernelId, kernel: Box<dyn CubeTask<Mi355Compiler>>) -> Result<Loaded, LaunchError> {
        if let Some(hit) = self.kernels.get(id) {
            return Ok(*hit);
        }
        let compiled = kernel.compile(kernel.define(), &mut self.compiler, &self.options)?;
        self.utilities.logger.log_compilation(&compiled);
        let loaded = match compiled.repr {
            Some(Mi355Kernel::Native(op)) => Loaded::Native(op),
            Some(Mi355Kernel::Hip(cpp)) => {
                validate_cube_dim(&self.utilities.properties, id)?;
                validate_units(&self.utilities.properties, id)?;
                let limit = self.utilities.properties.hardware.max_shared_memory_size;
                if cpp.shared_memory_size > limit {
                    return Err(cubecl_runtime::server::ResourceLimitError::SharedMemory {
                        requested: cpp.shared_memory_size,
                        max: limit,
                        backtrace: BackTrace::capture(),
                    }
                    .into());
                }
                let image = program::compile_hip(&compiled.source)?;
                Loaded::Jit(program::load(self.ctx, &image, &compiled.entrypoint_name, compiled.cube_dim, cpp.shared_memory_size)?)
            }
            None => {
                return Err(LaunchError::Unknown {
                    reason: format!("{} compiled to nothing", kernel.name()),
                    backtrace: BackTrace::capture(),
                });
            }
        };
        self.kernels.insert(id.clone(), loaded);
        Ok(loaded)
    }

    fn launch_checked(&mut self, kernel: Box<dyn CubeTask<Mi355Compiler>>, count: CubeCount, bindings: KernelArguments,
                      stream_id: StreamId, launch_mode: LaunchMode) -> Result<(), ServerError> {
        let id = kernel.id();
        let loaded = self.load(&id, kernel)?;
        if launch_mode.is_skipped() {
            return Ok(());
        }
        let ctx = self.ctx;
        let KernelArguments { resources, info } = bindings;
        let buffers: Vec<BufferBinding> = resources
            .into_iter()
            .map(|resource| match resource {
                KernelResource::Buffer(binding) => binding,
                KernelResource::TensorMap(map) => map.binding, // no TMA on CDNA4: the plain buffer is all there is
            })
            .collect();
        let mut pass = self.pass(stream_id, buffers.iter(), Pending::Keep)?;

        let grid = match count {
            CubeCount::Static(x, y, z) => [x, y, z],
            CubeCount::Dynamic(binding) => {
                // three u32 on the device: read them back (a host sync -- the reference does the same)
                let bytes = pass.download(CopyDescriptor::new(binding, [3].into(), [1].into(), 4))?;
                pass.lanes.current().fence().wait()?;
                let dims: &[u32] = bytemuck::cast_slice(&bytes);
                [dims[0], dims[1], dims[2]]
            }
        };
        if grid.contains(&0) {
            return Ok(());
        }

        let mut slices = buffers.into_iter().map(|b| pass.slice(b)).collect::<Result<Vec<_>, _>>()?;
        let sys = pass.sys();
        let rc = match loaded {
            Loaded::Native(op) => run_native(ctx, sys, op, &slices),
            Loaded::Jit(program) => {
                let info = pass.info_buffer(info.data)?;
                slices.push(pass.slice(info.binding())?);
                let pointers: Vec<*mut core::ffi::c_void> = slices.iter().map(|s| s.ptr).collect();
                let block = [program.cube_dim.x, program.cube_dim.y, program.cube_dim.z];
                unsafe {
                    mi355_launch(ctx, sys, program.function, grid.as_ptr(), block.as_ptr(), program.shared_mem_bytes as u32,
                                 pointers.as_ptr(), pointers.len() as u32)
                }
            }
        };
        let lane = pass.lanes.current();
        if lane.drop_queue.should_flush() {
            lane.release_staging();
        }
        error::check(ctx, rc)
    }

    fn replay_checked(&mut self, graph: GraphId, stream_id: StreamId) -> Result<(), ServerError> {
        let raw = self.graphs.get(&graph).map(|g| g.raw).ok_or_else(|| ServerError::graph_state("replay of an unknown or destroyed graph"))?;
        let ctx = self.ctx;
        let mut pass = self.pass_alone(stream_id, Pending::Keep)?;
        let rc = unsafe { mi355_graph_replay(ctx, pass.sys(), raw) };
        error::check(ctx, rc)
    }
}

/// The one place a native descriptor meets its buffers.  Buffer order is the one documented on [`NativeOp`].
fn run_native(ctx: *mut mi355_ctx, sys: mi355_stream, op: NativeOp, b: &[DeviceSlice]) -> i32 {
    let need = |n: usize| if b.len() < n { MI355_E_INVALID_ARGUMENT } else { MI355_OK };
    unsafe {
        match op {
            NativeOp::Gemm(key) => match need(3) {
                MI355_OK => mi355_gemm(ctx, sys, &key.desc(), b[0].ptr, b[1].ptr, b[2].ptr),
                rc => rc,
            },
            NativeOp::GemmAdd(key) => match need(4) {
                MI355_OK => mi355_gemm_add(ctx, sys, &key.desc(), b[0].ptr, b[1].ptr, b[2].ptr, b[3].ptr),
                rc => rc,
            },
            NativeOp::Reduce { kind, dtype, rows, cols, row_stride } => match kind {
                ReduceKind::Sum if b.len() >= 3 => mi355_reduce_sum(ctx, sys, b[0].ptr, dtype, cols, b[1].ptr as *mut f32, b[2].ptr, b[2].size),
                ReduceKind::Argmax if b.len() >= 4 => {
                    mi355_argmax(ctx, sys, b[0].ptr, dtype, cols, b[1].ptr as *mut f32, b[2].ptr as *mut u64, b[3].ptr, b[3].size)
                }
                ReduceKind::SumArgmax if b.len() >= 5 => mi355_sum_argmax(ctx, sys, b[0].ptr, dtype, cols, b[1].ptr as *mut f32,
                                                                          b[2].ptr as *mut f32, b[3].ptr as *mut u64, b[4].ptr, b[4].size),
                ReduceKind::RowSum if b.len() >= 2 => mi355_reduce_last_axis_sum(ctx, sys, b[0].ptr, dtype, b[1].ptr as *mut f32, rows, cols, row_stride),
                ReduceKind::RowArgmax if b.len() >= 2 => {
                    mi355_reduce_last_axis_argmax(ctx, sys, b[0].ptr, dtype, b[1].ptr as *mut u32, rows, cols, row_stride)
                }
                ReduceKind::Value(code) if b.len() >= 3 => mi355_reduce(ctx, sys, b[0].ptr, dtype, cols, code, b[1].ptr as *mut f32, b[2].ptr, b[2].size),
                ReduceKind::Index(code) if b.len() >= 4 => {
                    mi355_argreduce(ctx, sys, b[0].ptr, dtype, cols, code, b[1].ptr as *mut f32, b[2].ptr as *mut u64, b[3].ptr, b[3].size)
                }
                _ => MI355_E_INVALID_ARGUMENT,
            },
            NativeOp::ReduceAxis { op: code, dtype, outer, reduce, inner } => match need(2) {
                MI355_OK if code == MI355_REDUCE_ARGMAX || code == MI355_REDUCE_ARGMIN => {
                    mi355_argreduce_axis(ctx, sys, b[0].ptr, dtype, code, b[1].ptr as *mut u32, outer, reduce, inner)
                }
                MI355_OK => mi355_reduce_axis(ctx, sys, b[0].ptr, dtype, code, b[1].ptr as *mut f32, outer, reduce, inner),
                rc => rc,
            },
        }
    }
}

impl ComputeServer for Mi355Server {
    type Kernel = Box<dyn CubeTask<Mi355Compiler>>;
    type Info = ();
    type MemoryLayoutPolicy = PitchedMemoryLayoutPolicy;
    type Storage = DeviceStorage;

    fn initialize_memory(&mut self, memory: ManagedMemoryHandle, size: u64, stream_id: StreamId) {
        let mut pass = match self.pass_alone(stream_id, Pending::Keep) {
            Ok(pass) => pass,
            Err(err) => unreachable!("{err}"),
        };
        // the client already handed `memory` out, so a failure here has no caller to return to
        let reserved = pass.reserve(size).unwrap_or_else(|err| panic!("cannot reserve {size} bytes of HBM: {err}"));
        pass.bind(reserved, memory).unwrap_or_else(|err| panic!("cannot bind {size} bytes of HBM: {err}"));
    }

    fn staging(&mut self, sizes: &[usize], stream_id: StreamId) -> Result<Vec<Bytes>, ServerError> {
        let mut pass = self.pass_alone(stream_id, Pending::Keep)?;
        Ok(sizes.iter().map(|size| pass.host_bytes(*size, true)).collect())
    }

    fn logger(&self) -> Arc<ServerLogger> {
        self.lanes.logger.clone()
    }

    fn utilities(&self) -> Arc<ServerUtilities<Self>> {
        self.utilities.clone()
    }

    fn read(&mut self, descriptors: Vec<CopyDescriptor>, stream_id: StreamId) -> DynFut<Result<Vec<Bytes>, ServerError>> {
        let mut pass = match self.pass(stream_id, descriptors.iter().map(|d| &d.handle), Pending::Surface) {
            Ok(pass) => pass,
            Err(err) => return Box::pin(async move { Err(err) }),
        };
        if pass.lanes.current().capturing.is_recording() {
            return Box::pin(async { Err(ServerError::graph_state("a read synchronises the stream and cannot be captured")) });
        }
        // the bindings stay alive until the copies have run
        let keep: Vec<BufferBinding> = descriptors.iter().map(|d| d.handle.clone()).collect();
        let copies: Result<Vec<Bytes>, IoError> = descriptors.into_iter().map(|d| pass.download(d)).collect();
        let done = pass.lanes.current().fence();
        Box::pin(async move {
            let waited = done.wait();
            drop(keep);
            waited?;
            Ok(copies?)
        })
    }

    fn write(&mut self, descriptors: Vec<(CopyDescriptor, Bytes)>, stream_id: StreamId) {
        let mut pass = match self.pass(stream_id, descriptors.iter().map(|d| &d.0.handle), Pending::Keep) {
            Ok(pass) => pass,
            Err(err) => unreachable!("{err}"),
        };
        for (descriptor, data) in descriptors {
            if let Err(err) = pass.upload(descriptor, data) {
                pass.lanes.current().errors.push(err.into());
                return;
            }
        }
    }

    fn sync(&mut self, stream_id: StreamId) -> DynFut<Result<(), ServerError>> {
        let mut pass = match self.pass_alone(stream_id, Pending::Surface) {
            Ok(pass) => pass,
            Err(err) => return Box::pin(async move { Err(err) }),
        };
        let lane = pass.lanes.current();
        if lane.capturing.is_recording() {
            return Box::pin(async { Err(ServerError::graph_state("a sync cannot be captured")) });
        }
        let done = lane.fence();
        Box::pin(async move { done.wait() })
    }

    fn get_resource(&mut self, binding: BufferBinding, stream_id: StreamId) -> Result<ManagedResource<DeviceSlice>, ServerError> {
        let mut pass = self.pass(stream_id, [&binding].into_iter(), Pending::Keep)?;
        let keep = binding.memory.clone();
        Ok(ManagedResource::new(keep, pass.slice(binding)?))
    }

    unsafe fn launch(&mut self, kernel: Self::Kernel, count: CubeCount, bindings: KernelArguments, stream_id: StreamId, launch_mode: LaunchMode) {
        if let Err(err) = self.launch_checked(kernel, count, bindings, stream_id, launch_mode) {
            // fire-and-forget: the failure waits on the lane for the next flush / sync / read
            match self.timestamps.is_empty() {
                true => self.record(stream_id, err),
                false => self.timestamps.error(ProfileError::Server(Box::new(err))),
            }
        }
    }

    fn flush(&mut self, stream_id: StreamId) -> Result<(), ServerError> {
        let mut pass = self.pass_alone(stream_id, Pending::Surface)?;
        let lane = pass.lanes.current();
        lane.release_staging();
        lane.memory_management_gpu.storage().flush();
        Ok(())
    }

    fn graph_prepare(&mut self, stream_id: StreamId) -> Result<(), ServerError> {
        let mut pass = self.pass_alone(stream_id, Pending::Surface)?;
        let lane = pass.lanes.current();
        lane.capturing.prepare()?;
        // from here until end_capture the pools hand out slices that are never recycled: a replay finds every buffer
        // where the capture saw it
        lane.memory_management_gpu.capture_begin();
        lane.memory_management_cpu.capture_begin();
        Ok(())
    }

    fn begin_capture(&mut self, stream_id: StreamId) -> Result<(), ServerError> {
        let ctx = self.ctx;
        let mut pass = self.pass_alone(stream_id, Pending::Surface)?;
        let lane = pass.lanes.current();
        lane.capturing.begin()?;
        lane.release_staging();
        lane.release_staging();
        lane.memory_management_gpu.capture_priming_end();
        lane.memory_management_cpu.capture_priming_end();
        let rc = unsafe { mi355_graph_begin_capture(ctx, lane.sys) };
        if let Err(err) = error::check(ctx, rc) {
            lane.memory_management_gpu.capture_end();
            lane.memory_management_cpu.capture_end();
            lane.info_cache.capture_discard();
            lane.capturing.abort();
            return Err(err);
        }
        Ok(())
    }

    fn end_capture(&mut self, stream_id: StreamId) -> Result<GraphId, ServerError> {
        let ctx = self.ctx;
        let id = GraphId::new();
        let captured = {
            let mut pass = self.pass_alone(stream_id, Pending::Keep)?;
            let lane = pass.lanes.current();
            lane.capturing.end()?;
            let mut raw: *mut mi355_graph = core::ptr::null_mut();
            // the library instantiates the graph, refuses one that recorded a memory node, and releases its own pins
            // when this fails (cubecl_amd/csrc/runtime.cpp mi355_graph_end_capture)
            let ended = error::check(ctx, unsafe { mi355_graph_end_capture(ctx, lane.sys, &mut raw) });
            let mut pinned = lane.memory_management_gpu.capture_end();
            pinned.extend(lane.memory_management_cpu.capture_end());
            lane.release_staging();
            lane.release_staging();
            match ended {
                Ok(()) => {
                    lane.info_cache.capture_commit(id);
                    CapturedGraph { raw, _pinned: pinned }
                }
                Err(err) => {
                    lane.info_cache.capture_discard();
                    return Err(err);
                }
            }
        };
        self.graphs.insert(id, captured);
        Ok(id)
    }

    fn replay(&mut self, graph: GraphId, stream_id: StreamId) {
        if let Err(err) = self.replay_checked(graph, stream_id) {
            self.record(stream_id, err);
        }
    }

    fn graph_destroy(&mut self, graph: GraphId, stream_id: StreamId) {
        let Some(captured) = self.graphs.remove(&graph) else {
            return;
        };
        // mi355_graph_destroy waits for the streams the graph was replayed on before anything it pins is released
        // (ADVICE round 1); only after that may `_pinned` return its slices to the pools
        let destroyed = error::check(self.ctx, unsafe { mi355_graph_destroy(self.ctx, captured.raw) });
        drop(captured);
        if let Ok(mut lanes) = self.lanes.resolve(stream_id, [].into_iter(), false) {
            let lane = lanes.current();
            lane.info_cache.graph_release(graph);
            if let Err(err) = destroyed {
                lane.errors.push(err);
            }
        }
    }

    fn memory_usage(&mut self, stream_id: StreamId) -> Result<MemoryUsage, ServerError> {
        let mut pass = self.pass_alone(stream_id, Pending::Keep)?;
        Ok(pass.lanes.current().memory_management_gpu.memory_usage())
    }

    fn memory_report(&mut self, stream_id: StreamId) -> Result<MemoryReport, ServerError> {
        let mut pass = self.pass_alone(stream_id, Pending::Keep)?;
        Ok(pass.lanes.current().memory_management_gpu.memory_report())
    }

    fn stream_ids(&self) -> Vec<StreamId> {
        self.lanes.stream_ids().collect()
    }

    fn memory_cleanup(&mut self, stream_id: StreamId) {
        if let Ok(mut pass) = self.pass_alone(stream_id, Pending::Keep) {
            pass.cleanup();
        }
    }

    fn install_memory_pools(&mut self, config: MemoryConfiguration, stream_id: StreamId) -> Result<(), InstallMemoryPoolsError> {
        self.lanes.backend_mut().set_device_pools(config.clone());
        let (_, props) = self.lanes.backend_mut().device_pools();
        match self.pass_alone(stream_id, Pending::Keep) {
            Ok(mut pass) => pass.lanes.current().memory_management_gpu.install_pools(config, &props),
            Err(_) => Err(InstallMemoryPoolsError::StreamUnavailable),
        }
    }

    fn start_profile(&mut self, stream_id: StreamId) -> Result<ProfilingToken, ServerError> {
        cubecl_environment::future::block_on(self.sync(stream_id))?;
        Ok(self.timestamps.start())
    }

    fn end_profile(&mut self, stream_id: StreamId, token: ProfilingToken) -> Result<ProfileDuration, ProfileError> {
        if let Err(err) = cubecl_environment::future::block_on(self.sync(stream_id)) {
            self.timestamps.error(ProfileError::Server(Box::new(err)));
        }
        self.timestamps.stop(token)
    }

    fn allocation_mode(&mut self, mode: MemoryAllocationMode, stream_id: StreamId) {
        match self.pass_alone(stream_id, Pending::Keep) {
            Ok(mut pass) => pass.lanes.current().memory_management_gpu.mode(mode),
            Err(err) => unreachable!("{err}"),
        }
    }
}

impl Drop for Mi355Server {
    fn drop(&mut self) {
        for (_, graph) in self.graphs.drain() {
            unsafe { mi355_graph_destroy(self.ctx, graph.raw) };
        }
        comm::destroy_all(self);
        // the lanes drop after this body, each syncing and destroying its stream; `_context` goes last
    }
}
