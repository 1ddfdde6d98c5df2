//! One `Lane` per logical `StreamId`: a non-blocking `mi355_stream`, the reference's two memory pools on top of the
//! storages of [`crate::storage`], the queue of errors asynchronous launches left behind, the staging buffers still in
//! flight, the capture state and the cache of uploaded metadata words.
//!
//! `LaneBackend` is the `EventStreamBackend` (cubecl-runtime/src/stream/event.rs) `MultiStream` drives: when a binding
//! created on another lane shows up in `resolve`, `MultiStream` compares the cursor `handle_cursor` reports with what
//! the current lane last waited for and, if behind, calls `flush` on the origin lane and `wait_event` on this one --
//! which here become `mi355_event_record` and `mi355_stream_wait_event`.  That is the whole cross-stream contract;
//! nothing in this crate ever passes a null stream.
use crate::{
    fence::Fence,
    ffi::*,
    storage::{DeviceStorage, PINNED_ALIGNMENT, PinnedStorage},
};
use cubecl_ir::MemoryDeviceProperties;
use cubecl_runtime::{
    logging::ServerLogger,
    memory_management::{
        MemoryAllocationMode, MemoryConfiguration, MemoryManagement, MemoryManagementOptions,
        drop_queue::{FlushingPolicy, PendingDropQueue},
    },
    metadata_cache::{MetadataCachePolicy, MetadataInfoCache},
    server::{BufferBinding, Handle, ServerError},
    stream::{EventStreamBackend, StreamCaptureState},
};
use std::sync::Arc;

#[derive(Debug)]
pub struct Lane {
    pub(crate) ctx: *mut mi355_ctx,
    pub(crate) sys: mi355_stream,
    pub memory_management_gpu: MemoryManagement<DeviceStorage>,
    pub memory_management_cpu: MemoryManagement<PinnedStorage>,
    pub errors: Vec<ServerError>,
    pub drop_queue: PendingDropQueue<Fence>,
    pub capturing: StreamCaptureState,
    pub info_cache: MetadataInfoCache<Handle>,
}
unsafe impl Send for Lane {}

impl Lane {
    /// A fence behind everything this lane has enqueued.
    pub fn fence(&self) -> Fence {
        Fence::after(self.ctx, self.sys)
    }

    /// Lets staging buffers whose copies have completed go back to the pinned pool (two generations, as the queue keeps
    /// the last flushed batch until the next flush).
    pub fn release_staging(&mut self) {
        if self.capturing.is_recording() {
            return; // an event record/synchronise is not capturable
        }
        let (ctx, sys) = (self.ctx, self.sys);
        self.drop_queue.flush(|| Fence::after(ctx, sys));
    }
}

impl Drop for Lane {
    fn drop(&mut self) {
        unsafe {
            mi355_sync(self.ctx, self.sys);
            mi355_stream_destroy(self.ctx, self.sys);
        }
    }
}

#[derive(Debug)]
pub struct LaneBackend {
    ctx: *mut mi355_ctx,
    mem_props: MemoryDeviceProperties,
    mem_config: MemoryConfiguration,
    pools_override: Option<MemoryConfiguration>,
    logger: Arc<ServerLogger>,
}
unsafe impl Send for LaneBackend {}

impl LaneBackend {
    pub fn new(ctx: *mut mi355_ctx, mem_props: MemoryDeviceProperties, mem_config: MemoryConfiguration, logger: Arc<ServerLogger>) -> Self {
        Self { ctx, mem_props, mem_config, pools_override: None, logger }
    }

    /// The pool layout lanes created from now on get (`ComputeServer::install_memory_pools`).
    pub fn set_device_pools(&mut self, config: MemoryConfiguration) {
        self.pools_override = Some(config);
    }

    pub fn device_pools(&self) -> (MemoryConfiguration, MemoryDeviceProperties) {
        (self.pools_override.clone().unwrap_or_else(|| self.mem_config.clone()), self.mem_props.clone())
    }
}

impl EventStreamBackend for LaneBackend {
    type Stream = Lane;
    type Event = Fence;

    fn create_stream(&self) -> Self::Stream {
        let mut sys: mi355_stream = core::ptr::null_mut();
        let rc = unsafe { mi355_stream_create(self.ctx, &mut sys) };
        assert_eq!(rc, MI355_OK, "mi355_stream_create: {}", crate::error::last_message(self.ctx));

        let (device_config, device_props) = self.device_pools();
        let memory_management_gpu = MemoryManagement::from_configuration(
            DeviceStorage::new(self.ctx, self.mem_props.alignment as usize),
            &device_props,
            device_config,
            self.logger.clone(),
            MemoryManagementOptions::new("MI355X HBM"),
        );
        let memory_management_cpu = MemoryManagement::from_configuration(
            PinnedStorage::new(self.ctx),
            &MemoryDeviceProperties { max_page_size: self.mem_props.max_page_size, alignment: PINNED_ALIGNMENT as u64 },
            self.mem_config.clone(),
            self.logger.clone(),
            MemoryManagementOptions::new("Pinned host memory").mode(MemoryAllocationMode::Auto),
        );
        Lane {
            ctx: self.ctx,
            sys,
            memory_management_gpu,
            memory_management_cpu,
            errors: Vec::new(),
            drop_queue: PendingDropQueue::new(FlushingPolicy::default()),
            capturing: StreamCaptureState::NoCapture,
            info_cache: MetadataInfoCache::new(MetadataCachePolicy::default()),
        }
    }

    fn handle_cursor(stream: &Self::Stream, handle: &BufferBinding) -> u64 {
        // a slice that was freed or re-bound since has no cursor any more: u64::MAX forces the wait
        stream.memory_management_gpu.get_cursor(handle.memory.clone()).unwrap_or(u64::MAX)
    }

    fn is_healthy(stream: &Self::Stream) -> bool {
        stream.errors.is_empty()
    }

    fn flush(stream: &mut Self::Stream) -> Self::Event {
        stream.fence()
    }

    fn wait_event(stream: &mut Self::Stream, event: Self::Event) {
        event.hold(stream.sys);
    }

    fn wait_event_sync(event: Self::Event) -> Result<(), ServerError> {
        event.wait()
    }
}
