//! The two `ComputeStorage`s the reference's `MemoryManagement` pools sit on (storage/base.rs `ComputeStorage`):
//! device memory from `mi355_alloc` / `mi355_free` (plain driver allocations -- the pooling is the reference's, above
//! this), and page-locked host memory from `mi355_pinned_alloc` / `mi355_pinned_free` for staging and read-backs.
//! Frees are queued and handed to the library at `flush`, which is when the reference's pools say a page is really gone.
use crate::{error, ffi::*};
use cubecl_common::bytes::{AccessError, AccessPolicy, AllocationController, AllocationProperty};
use cubecl_environment::backtrace::BackTrace;
use cubecl_runtime::memory_management::ManagedMemoryBinding;
use cubecl_runtime::server::IoError;
use cubecl_runtime::storage::{ComputeStorage, StorageHandle, StorageId, StorageUtilization};
use std::collections::HashMap;

/// Alignment of `hipHostMalloc` memory the pinned pool advertises.
pub const PINNED_ALIGNMENT: usize = 4096;

/// What a kernel binding resolves to: a raw device pointer and the bytes behind it.
#[derive(Debug, Clone, Copy)]
pub struct DeviceSlice {
    pub ptr: *mut core::ffi::c_void,
    pub size: u64,
}
unsafe impl Send for DeviceSlice {}

/// A window of page-locked host memory.
#[derive(Debug, Clone, Copy)]
pub struct PinnedSlice {
    pub ptr: *mut u8,
    pub size: usize,
}
unsafe impl Send for PinnedSlice {}

#[derive(Debug)]
pub struct DeviceStorage {
    ctx: *mut mi355_ctx,
    alignment: usize,
    live: HashMap<StorageId, *mut core::ffi::c_void>,
    retired: Vec<*mut core::ffi::c_void>,
}
unsafe impl Send for DeviceStorage {}

impl DeviceStorage {
    pub fn new(ctx: *mut mi355_ctx, alignment: usize) -> Self {
        Self { ctx, alignment, live: HashMap::new(), retired: Vec::new() }
    }
}

fn missing(what: &'static str) -> IoError {
    IoError::StorageHandleNotFound { reason: what.into(), backtrace: BackTrace::capture() }
}

impl ComputeStorage for DeviceStorage {
    type Resource = DeviceSlice;

    fn alignment(&self) -> usize {
        self.alignment
    }

    fn get(&mut self, handle: &StorageHandle) -> Result<Self::Resource, IoError> {
        let base = *self.live.get(&handle.id).ok_or_else(|| missing("no device allocation under this storage id"))?;
        let ptr = unsafe { (base as *mut u8).add(handle.offset() as usize) } as *mut core::ffi::c_void;
        Ok(DeviceSlice { ptr, size: handle.size() })
    }

    fn alloc(&mut self, size: u64) -> Result<StorageHandle, IoError> {
        let mut dptr = core::ptr::null_mut();
        let rc = unsafe { mi355_alloc(self.ctx, size, &mut dptr) };
        if rc != MI355_OK {
            return Err(error::io(self.ctx, rc, size));
        }
        let id = StorageId::new();
        self.live.insert(id, dptr);
        Ok(StorageHandle::new(id, StorageUtilization { offset: 0, size }))
    }

    fn dealloc(&mut self, id: StorageId) {
        if let Some(ptr) = self.live.remove(&id) {
            self.retired.push(ptr);
        }
    }

    fn flush(&mut self) {
        for ptr in self.retired.drain(..) {
            unsafe { mi355_free(self.ctx, ptr) };
        }
        unsafe { mi355_flush(self.ctx) };
    }
}

/// A lane (or the whole server) going away takes its pools with it: `MemoryManagement` does not `dealloc` its pages on
/// drop, and `mi355_ctx_destroy` only frees what was queued through `mi355_free`.  Runs while the context is still alive
/// (`Mi355Server` declares its `ContextGuard` last).
impl Drop for DeviceStorage {
    fn drop(&mut self) {
        for ptr in self.retired.drain(..).chain(self.live.drain().map(|(_, p)| p)) {
            unsafe { mi355_free(self.ctx, ptr) };
        }
        unsafe { mi355_flush(self.ctx) };
    }
}

#[derive(Debug)]
pub struct PinnedStorage {
    ctx: *mut mi355_ctx,
    live: HashMap<StorageId, (*mut u8, usize)>,
    retired: Vec<*mut u8>,
}
unsafe impl Send for PinnedStorage {}

impl PinnedStorage {
    pub fn new(ctx: *mut mi355_ctx) -> Self {
        Self { ctx, live: HashMap::new(), retired: Vec::new() }
    }
}

impl ComputeStorage for PinnedStorage {
    type Resource = PinnedSlice;

    fn alignment(&self) -> usize {
        PINNED_ALIGNMENT
    }

    fn get(&mut self, handle: &StorageHandle) -> Result<Self::Resource, IoError> {
        let (base, _) = *self.live.get(&handle.id).ok_or_else(|| missing("no pinned allocation under this storage id"))?;
        Ok(PinnedSlice { ptr: unsafe { base.add(handle.offset() as usize) }, size: handle.size() as usize })
    }

    fn alloc(&mut self, size: u64) -> Result<StorageHandle, IoError> {
        let mut hptr = core::ptr::null_mut();
        let rc = unsafe { mi355_pinned_alloc(self.ctx, size, &mut hptr) };
        if rc != MI355_OK {
            return Err(error::io(self.ctx, rc, size));
        }
        let id = StorageId::new();
        self.live.insert(id, (hptr as *mut u8, size as usize));
        Ok(StorageHandle::new(id, StorageUtilization { offset: 0, size }))
    }

    fn dealloc(&mut self, id: StorageId) {
        if let Some((ptr, _)) = self.live.remove(&id) {
            self.retired.push(ptr);
        }
    }

    fn flush(&mut self) {
        for ptr in self.retired.drain(..) {
            unsafe { mi355_pinned_free(self.ctx, ptr as *mut core::ffi::c_void) };
        }
    }
}

impl Drop for PinnedStorage {
    fn drop(&mut self) {
        // Only pages the pool has already handed back (`dealloc`) are freed here.  LIVE pages are left to the process: a
        // `Bytes` returned by `read` / `staging` (`PinnedBytes`) holds a binding of the memory pool, not of this storage, and
        // may outlive the lane or the server that owned the storage -- freeing its page here would leave that `Bytes`
        // pointing at unmapped host memory (advisor, round 3).  The reference frees a pinned page in `dealloc` and has no `Drop`
        // for the storage either (crates/cubecl-hip/src/compute/storage/cpu.rs:131-141): live pages end with the process.
        for ptr in self.retired.drain(..) {
            unsafe { mi355_pinned_free(self.ctx, ptr as *mut core::ffi::c_void) };
        }
    }
}

/// Lets a `Bytes` live in a slice of the pinned pool: the binding keeps the slice reserved until the `Bytes` is dropped
/// (for a staging buffer that is when the stream's drop queue has seen its fence).
pub struct PinnedBytes {
    slice: PinnedSlice,
    _keep: ManagedMemoryBinding,
}

impl PinnedBytes {
    pub fn new(keep: ManagedMemoryBinding, slice: PinnedSlice) -> Self {
        Self { slice, _keep: keep }
    }

    fn span(&self) -> (*mut core::mem::MaybeUninit<u8>, usize) {
        match self.slice.size {
            // a zero-length slice still needs a non-null, aligned pointer
            0 => (core::ptr::without_provenance_mut(PINNED_ALIGNMENT), 0),
            n => (self.slice.ptr as *mut core::mem::MaybeUninit<u8>, n),
        }
    }
}

impl AllocationController for PinnedBytes {
    fn alloc_align(&self) -> usize {
        PINNED_ALIGNMENT
    }

    fn property(&self) -> AllocationProperty {
        AllocationProperty::Pinned
    }

    unsafe fn memory_mut(&mut self, _policy: AccessPolicy) -> Result<&mut [core::mem::MaybeUninit<u8>], AccessError> {
        let (ptr, len) = self.span();
        Ok(unsafe { core::slice::from_raw_parts_mut(ptr, len) })
    }

    fn memory(&self, _policy: AccessPolicy) -> Result<&[core::mem::MaybeUninit<u8>], AccessError> {
        let (ptr, len) = self.span();
        Ok(unsafe { core::slice::from_raw_parts(ptr, len) })
    }
}
