//! `Mi355Server`: the `ComputeServer` + `ServerCommunication` implementation.  One instance per
//! `DeviceId`, created on that device's runner thread by `DeviceService::init`
//! (crates/cubecl-common/src/device/handle/channel.rs:55-66) and only ever touched from it -- the
//! single-thread-per-context contract of the C ABI.
//!
//! Memory management (pools, handles, pending drops) is the reference's own
//! `MemoryManagement<Mi355Storage>`; only the storage and the device operations are replaced.
use crate::error::check;
use crate::ffi::*;
use crate::storage::{Mi355Resource, Mi355Storage};
use cubecl_common::{bytes::Bytes, device::DeviceId, future::DynFut, profile::ProfileDuration, stream_id::StreamId};
use cubecl_ir::ElemType;
use cubecl_runtime::memory_management::{MemoryAllocationMode, MemoryManagement, MemoryUsage, ManagedMemoryHandle};
use cubecl_runtime::server::*;
use cubecl_runtime::storage::ManagedResource;
use std::collections::HashMap;
use std::sync::Arc;

/// A kernel this server can launch: either a code object the caller brought (hand-written or
/// produced elsewhere) addressed by symbol name, launched through the reference's pointer-array
/// ABI (crates/cubecl-cpp/src/hip/signature.rs:28-62), or nothing at all for the library ops in
/// `ops.rs`, which do not go through `launch`.
pub struct ExternalKernel {
    pub image: Arc<Vec<u8>>,
    pub entry: std::ffi::CString,
    pub cube_dim: CubeDim,
    pub shared_mem_bytes: u32,
}

pub struct Mi355Server {
    pub(crate) ctx: *mut mi355_ctx,
    pub(crate) props: mi355_device_props_t,
    device: DeviceId,
    memory: MemoryManagement<Mi355Storage>,
    comms: HashMap<CommunicationId, (*mut mi355_comm, Vec<DeviceId>)>,
    modules: HashMap<usize, (mi355_module, HashMap<std::ffi::CString, mi355_function>)>,
    utilities: Arc<ServerUtilities<Self>>,
}
unsafe impl Send for Mi355Server {} // moved once onto the runner thread, never shared (server.rs:163-167)

impl Mi355Server {
    fn resource(&mut self, binding: BufferBinding, stream_id: StreamId) -> Result<Mi355Resource, ServerError> {
        Ok(*self.get_resource(binding, stream_id)?.resource())
    }

    fn dtype_code(dtype: ElemType) -> Result<i32, ServerError> {
        use cubecl_ir::{FloatKind::*, IntKind::*, UIntKind::*};
        Ok(match dtype {
            ElemType::Float(F32) => MI355_DTYPE_F32,
            ElemType::Float(BF16) => MI355_DTYPE_BF16,
            ElemType::Float(F16) => MI355_DTYPE_F16,
            ElemType::Float(F64) => MI355_DTYPE_F64,
            ElemType::Int(I32) => MI355_DTYPE_I32,
            ElemType::UInt(U32) => MI355_DTYPE_U32,
            ElemType::Int(I64) => MI355_DTYPE_I64,
            ElemType::UInt(U64) => MI355_DTYPE_U64,
            ElemType::UInt(U8) => MI355_DTYPE_U8,
            ElemType::Int(I8) => MI355_DTYPE_I8,
            other => return Err(ServerError::Generic {
                reason: format!("collective on unsupported element type {other:?}"),
                backtrace: cubecl_common::backtrace::BackTrace::capture() }),
        })
    }
}

impl cubecl_common::device::DeviceService for Mi355Server {
    fn init(device_id: DeviceId) -> Self {
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { mi355_ctx_create(device_id.index_id as i32, &mut ctx) };
        // mi355_ctx_create refuses anything that is not wave64 gfx950; unlike the reference's
        // `AMDArchitecture::parse` assert (crates/cubecl-hip/src/runtime.rs:118-124) it knows gfx950.
        assert_eq!(rc, MI355_OK, "mi355_ctx_create: {}", crate::error::last_message(core::ptr::null_mut()));
        let mut props = unsafe { core::mem::zeroed::<mi355_device_props_t>() };
        unsafe { mi355_device_props(ctx, &mut props) };
        let storage = Mi355Storage::new(ctx, props.mem_alignment as usize);
        let (memory, utilities) = crate::runtime::build_memory_and_utilities(&props, storage, device_id);
        Self { ctx, props, device: device_id, memory, comms: HashMap::new(), modules: HashMap::new(), utilities }
    }

    fn utilities(&self) -> cubecl_common::device::ServerUtilitiesHandle {
        self.utilities.clone() as _
    }
}

impl ComputeServer for Mi355Server {
    type Kernel = Box<dyn cubecl_runtime::compiler::CubeTask<crate::runtime::AotCompiler>>;
    type Info = ();
    type MemoryLayoutPolicy = cubecl_runtime::allocator::PitchedMemoryLayoutPolicy;
    type Storage = Mi355Storage;

    fn initialize_memory(&mut self, memory: ManagedMemoryHandle, size: u64, _stream_id: StreamId) {
        self.memory.reserve_into(memory, size); // MemoryManagement -> Mi355Storage::alloc -> mi355_alloc
    }

    fn logger(&self) -> Arc<cubecl_runtime::logging::ServerLogger> {
        self.utilities.logger.clone()
    }

    fn utilities(&self) -> Arc<ServerUtilities<Self>> {
        self.utilities.clone()
    }

    fn read(&mut self, descriptors: Vec<CopyDescriptor>, stream_id: StreamId) -> DynFut<Result<Vec<Bytes>, ServerError>> {
        // mi355_read enqueues the D2H copy, waits for it and reports queued launch errors
        // (crates/cubecl-hip/src/compute/command.rs:244-265, :538-614)
        let mut out = Vec::with_capacity(descriptors.len());
        let mut status = Ok(());
        for d in descriptors {
            let elem = d.elem_size as u64;
            let rows: u64 = d.shape.iter().rev().skip(1).map(|x| *x as u64).product::<u64>().max(1);
            let width = d.shape.last().copied().unwrap_or(1) as u64 * elem;
            let pitch = if d.shape.len() >= 2 { d.strides[d.shape.len() - 2] as u64 * elem } else { width };
            let res = match self.resource(d.handle, stream_id) { Ok(r) => r, Err(e) => { status = Err(e); break } };
            let mut bytes = Bytes::from_bytes_vec(vec![0u8; (rows * width) as usize]);
            let rc = unsafe {
                if pitch == width {
                    mi355_read(self.ctx, core::ptr::null_mut(), bytes.as_mut_ptr() as _, res.ptr, rows * width)
                } else {
                    mi355_read_2d(self.ctx, core::ptr::null_mut(), bytes.as_mut_ptr() as _, width, res.ptr, pitch, width, rows)
                }
            };
            if let Err(e) = check(self.ctx, rc) { status = Err(e); break }
            out.push(bytes);
        }
        Box::pin(async move { status.map(|_| out) })
    }

    fn write(&mut self, descriptors: Vec<(CopyDescriptor, Bytes)>, stream_id: StreamId) {
        for (d, data) in descriptors {
            if data.is_empty() { continue }              // empty tensors skip the copy (command.rs:363-369)
            if let Ok(res) = self.resource(d.handle, stream_id) {
                // stream-ordered; failures are queued inside the library and surface at the next flush/sync/read
                unsafe { mi355_write(self.ctx, core::ptr::null_mut(), res.ptr, data.as_ptr() as _, data.len() as u64) };
                self.memory.keep_alive_until_flush(data); // host bytes must outlive the copy (command.rs:402)
            }
        }
    }

    fn sync(&mut self, _stream_id: StreamId) -> DynFut<Result<(), ServerError>> {
        let r = check(self.ctx, unsafe { mi355_sync(self.ctx, core::ptr::null_mut()) });
        Box::pin(async move { r })
    }

    fn get_resource(&mut self, binding: BufferBinding, _stream_id: StreamId)
        -> Result<ManagedResource<Mi355Resource>, ServerError> {
        self.memory.get_resource(binding).map_err(Into::into)
    }

    unsafe fn launch(&mut self, kernel: Self::Kernel, count: CubeCount, bindings: KernelArguments,
                     stream_id: StreamId, _mode: LaunchMode) {
        // CubeTask::compile() of an external kernel returns a CompiledKernel naming a PRE-BUILT gfx950 entry
        // point (README.md:222-224 "external kernels"); nothing is generated or JIT-compiled here.
        let compiled = kernel.compile(&mut crate::runtime::AotCompiler, &Default::default(), Default::default(), kernel.address_type());
        let Ok(compiled) = compiled else { return };
        let ext: &ExternalKernel = compiled.repr.as_ref().expect("AOT kernels carry their code object");
        let grid = match count {
            CubeCount::Static(x, y, z) => [x, y, z],
            CubeCount::Dynamic(b) => {   // blocking 12-byte read-back, as the reference does (server.rs:779-793)
                let r = self.resource(b, stream_id).expect("dynamic cube count binding");
                let mut g = [0u32; 3];
                unsafe { mi355_read(self.ctx, core::ptr::null_mut(), g.as_mut_ptr() as _, r.ptr, 12) };
                g
            }
        };
        if grid.iter().any(|d| *d == 0) { return }        // zero-sized grid = no-op (client.rs:880-884)
        let key = Arc::as_ptr(&ext.image) as usize;
        let ctx = self.ctx;
        let (module, funcs) = self.modules.entry(key).or_insert_with(|| {
            let mut m = core::ptr::null_mut();
            unsafe { mi355_module_load(ctx, ext.image.as_ptr() as _, ext.image.len(), &mut m) };
            (m, HashMap::new())
        });
        let func = *funcs.entry(ext.entry.clone()).or_insert_with(|| {
            let mut f = core::ptr::null_mut();
            unsafe { mi355_module_get_function(ctx, *module, ext.entry.as_ptr(), &mut f) };
            f
        });
        // one pointer per buffer binding, the info buffer last (crates/cubecl-hip/src/compute/server.rs:816)
        let mut ptrs: Vec<*mut core::ffi::c_void> = Vec::with_capacity(bindings.resources.len() + 1);
        for r in bindings.resources {
            if let KernelResource::Buffer(b) = r { if let Ok(res) = self.resource(b, stream_id) { ptrs.push(res.ptr) } }
        }
        let info = self.memory.upload_info(&bindings.info.data);   // pinned staging + content cache (server.rs:128-148)
        ptrs.push(info.ptr);
        let block = [ext.cube_dim.x, ext.cube_dim.y, ext.cube_dim.z];
        // resource-limit violations are QUEUED by the library and surface from flush() as
        // ServerUnhealthy{errors:[Launch(TooManyResources(..))]} (runtime_tests/launch.rs:226-348)
        unsafe { mi355_launch(self.ctx, core::ptr::null_mut(), func, grid.as_ptr(), block.as_ptr(), ext.shared_mem_bytes,
                              ptrs.as_ptr(), ptrs.len() as u32) };
    }

    fn flush(&mut self, _stream_id: StreamId) -> Result<(), ServerError> {
        check(self.ctx, unsafe { mi355_flush(self.ctx) })
    }

    fn memory_usage(&mut self, _stream_id: StreamId) -> Result<MemoryUsage, ServerError> {
        Ok(self.memory.memory_usage())
    }

    fn memory_report(&mut self, _stream_id: StreamId) -> Result<cubecl_runtime::memory_management::MemoryReport, ServerError> {
        Ok(self.memory.memory_report())
    }

    fn memory_cleanup(&mut self, _stream_id: StreamId) {
        self.memory.cleanup(true);
    }

    fn start_profile(&mut self, _stream_id: StreamId) -> Result<ProfilingToken, ServerError> {
        let mut token = 0u64;
        check(self.ctx, unsafe { mi355_profile_start(self.ctx, core::ptr::null_mut(), &mut token) })?;
        Ok(ProfilingToken { id: token })
    }

    fn end_profile(&mut self, _stream_id: StreamId, token: ProfilingToken) -> Result<ProfileDuration, ProfileError> {
        let mut nanos = 0u64;
        let rc = unsafe { mi355_profile_stop(self.ctx, core::ptr::null_mut(), token.id, &mut nanos) };
        if rc != MI355_OK { return Err(ProfileError::NotRegistered { backtrace: cubecl_common::backtrace::BackTrace::capture() }) }
        // TimingMethod::Device: GPU time from hipEvents (the reference HIP backend reports System time, runtime.rs:198)
        Ok(ProfileDuration::new_device_time(core::time::Duration::from_nanos(nanos)))
    }

    fn allocation_mode(&mut self, mode: MemoryAllocationMode, _stream_id: StreamId) {
        self.memory.mode(mode);
    }
}

impl ServerCommunication for Mi355Server {
    /// RCCL is present on every ROCm install; the library reports it in the property block.
    const SERVER_COMM_ENABLED: bool = true;

    fn comm_init(&mut self, device_ids: Vec<DeviceId>) -> Result<(), ServerError> {
        let id = CommunicationId::from(device_ids.clone());
        if self.comms.contains_key(&id) { return Ok(()) }
        let mut sorted = device_ids; sorted.sort();
        let rank = sorted.iter().position(|d| *d == self.device).expect("own device in the set") as i32;
        // one ncclUniqueId per device set, created by whoever asks first (communication.rs:14-25)
        let uid = crate::runtime::unique_id_for(&id);
        let mut comm = core::ptr::null_mut();
        check(self.ctx, unsafe { mi355_comm_init(self.ctx, uid.as_ptr(), rank, sorted.len() as i32, &mut comm) })?;
        self.comms.insert(id, (comm, sorted));
        Ok(())
    }

    fn all_reduce(&mut self, src: BufferBinding, dst: BufferBinding, dtype: ElemType, stream_id: StreamId,
                  op: ReduceOperation, device_ids: Vec<DeviceId>) -> Result<(), ServerError> {
        let (comm, _) = *self.comms.get(&CommunicationId::from(device_ids)).expect("comm_init first");
        let (s, d) = (self.resource(src, stream_id)?, self.resource(dst, stream_id)?);
        let code = Self::dtype_code(dtype)?;
        let count = s.size / dtype.size() as u64;                       // get_nccl_dtype_count (communication.rs:34-108)
        let op = match op { ReduceOperation::Sum => MI355_REDUCE_SUM, ReduceOperation::Mean => MI355_REDUCE_MEAN };
        // the library fences compute -> comm stream and issues ncclAllReduce on the comm stream (server.rs:705-780)
        check(self.ctx, unsafe { mi355_all_reduce(self.ctx, comm, core::ptr::null_mut(), s.ptr, d.ptr, count, code, op) })
    }

    fn sync_collective(&mut self, _stream_id: StreamId) -> Result<(), ServerError> {
        check(self.ctx, unsafe { mi355_sync_collective(self.ctx, core::ptr::null_mut()) })
    }

    fn send(&mut self, desc: CopyDescriptor, dtype: ElemType, stream_id: StreamId, dst: DeviceId) -> Result<(), ServerError> {
        let r = self.resource(desc.handle, stream_id)?;
        let (comm, ids) = self.comms.values().find(|(_, ids)| ids.contains(&dst)).expect("comm_init first").clone();
        let peer = ids.iter().position(|d| *d == dst).unwrap() as i32;
        check(self.ctx, unsafe { mi355_send(self.ctx, comm, core::ptr::null_mut(), r.ptr, r.size / dtype.size() as u64,
                                            Self::dtype_code(dtype)?, peer) })
    }

    fn recv(&mut self, handle: Handle, dtype: ElemType, stream_id: StreamId, src: DeviceId) -> Result<(), ServerError> {
        let r = self.resource(handle.binding(), stream_id)?;
        let (comm, ids) = self.comms.values().find(|(_, ids)| ids.contains(&src)).expect("comm_init first").clone();
        let peer = ids.iter().position(|d| *d == src).unwrap() as i32;
        check(self.ctx, unsafe { mi355_recv(self.ctx, comm, core::ptr::null_mut(), r.ptr, r.size / dtype.size() as u64,
                                            Self::dtype_code(dtype)?, peer) })
    }
}

impl Drop for Mi355Server {
    fn drop(&mut self) {
        for (comm, _) in self.comms.values() { unsafe { mi355_comm_destroy(self.ctx, *comm) }; }
        unsafe { mi355_ctx_destroy(self.ctx) };
    }
}

impl core::fmt::Debug for Mi355Server {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        f.debug_struct("Mi355Server").field("device", &self.device).finish()
    }
}
