//! A one-shot marker in a `mi355_stream`: recorded at creation, consumed by exactly one wait (the role of
//! crates/cubecl-hip/src/compute/fence.rs, over `mi355_event_create` / `_record` / `_sync` / `mi355_stream_wait_event`).
use crate::{error, ffi::*};
use cubecl_runtime::memory_management::drop_queue;
use cubecl_runtime::server::ServerError;

#[derive(Debug)]
pub struct Fence {
    ctx: *mut mi355_ctx,
    event: mi355_event,
}
unsafe impl Send for Fence {}

impl Fence {
    /// Records "everything enqueued on `stream` so far".
    pub fn after(ctx: *mut mi355_ctx, stream: mi355_stream) -> Self {
        let mut event: mi355_event = core::ptr::null_mut();
        unsafe {
            let rc = mi355_event_create(ctx, &mut event);
            assert_eq!(rc, MI355_OK, "mi355_event_create: {}", error::last_message(ctx));
            let rc = mi355_event_record(ctx, event, stream);
            assert_eq!(rc, MI355_OK, "mi355_event_record: {}", error::last_message(ctx));
        }
        Self { ctx, event }
    }

    /// Makes `stream` wait for the marker on the device; the host does not block.
    pub fn hold(self, stream: mi355_stream) {
        unsafe {
            let rc = mi355_stream_wait_event(self.ctx, stream, self.event);
            assert_eq!(rc, MI355_OK, "mi355_stream_wait_event: {}", error::last_message(self.ctx));
            mi355_event_destroy(self.ctx, self.event);
        }
    }

    /// Blocks the calling thread until the marker has been reached.
    pub fn wait(self) -> Result<(), ServerError> {
        let rc = unsafe { mi355_event_sync(self.ctx, self.event) };
        let waited = error::check(self.ctx, rc);
        unsafe { mi355_event_destroy(self.ctx, self.event) };
        waited
    }
}

impl drop_queue::Fence for Fence {
    fn sync(self) {
        let _ = self.wait();
    }
}
