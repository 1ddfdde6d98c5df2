//! `ComputeStorage` over the C ABI: plain `hipMalloc` per allocation, deferred `hipFree` at flush,
//! exactly the policy of crates/cubecl-hip/src/compute/storage/gpu.rs:136-169.
use crate::{error, ffi::*};
use cubecl_runtime::server::IoError;
use cubecl_runtime::storage::{ComputeStorage, StorageHandle, StorageId, StorageUtilization};
use std::collections::HashMap;

/// What a kernel binding resolves to: a raw device pointer + the bytes in use.
#[derive(Debug, Clone, Copy)]
pub struct Mi355Resource {
    pub ptr: *mut core::ffi::c_void,
    pub size: u64,
}
unsafe impl Send for Mi355Resource {}

pub struct Mi355Storage {
    pub(crate) ctx: *mut mi355_ctx,
    alignment: usize,
    memory: HashMap<StorageId, *mut core::ffi::c_void>,
}
unsafe impl Send for Mi355Storage {}

impl Mi355Storage {
    pub fn new(ctx: *mut mi355_ctx, alignment: usize) -> Self {
        Self { ctx, alignment, memory: HashMap::new() }
    }
}

impl ComputeStorage for Mi355Storage {
    type Resource = Mi355Resource;

    fn alignment(&self) -> usize {
        self.alignment
    }

    fn get(&mut self, handle: &StorageHandle) -> Result<Self::Resource, IoError> {
        let base = *self.memory.get(&handle.id).ok_or_else(|| IoError::StorageHandleNotFound {
            backtrace: cubecl_common::backtrace::BackTrace::capture(),
        })?;
        let ptr = unsafe { (base as *mut u8).add(handle.offset() as usize) } as *mut core::ffi::c_void;
        Ok(Mi355Resource { ptr, size: handle.size() })
    }

    fn alloc(&mut self, size: u64) -> Result<StorageHandle, IoError> {
        let mut dptr = core::ptr::null_mut();
        let rc = unsafe { mi355_alloc(self.ctx, size, &mut dptr) };
        if rc != MI355_OK {
            // a driver OOM maps to OutOfMemory, an over-size request to BufferTooBig (server/base.rs:895-911)
            return match error::convert(rc, size, 0, error::last_message(self.ctx)) {
                cubecl_runtime::server::ServerError::Io(e) => Err(e),
                other => Err(IoError::Unknown { description: other.to_string(),
                                                backtrace: cubecl_common::backtrace::BackTrace::capture() }),
            };
        }
        let id = StorageId::new();
        self.memory.insert(id, dptr);
        Ok(StorageHandle::new(id, StorageUtilization { offset: 0, size }))
    }

    fn dealloc(&mut self, id: StorageId) {
        if let Some(ptr) = self.memory.remove(&id) {
            unsafe { mi355_free(self.ctx, ptr) }; // deferred inside the library until mi355_flush
        }
    }

    fn flush(&mut self) {
        unsafe { mi355_flush(self.ctx) };
    }
}
