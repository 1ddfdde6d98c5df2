//! Status-code conversion.  The C ABI never unwinds; each call returns an `int32` and queued
//! (asynchronous) launch failures are drained with `mi355_error_pop` when
//! `MI355_E_SERVER_UNHEALTHY` comes back from flush / sync / read -- the reference's contract in
//! crates/cubecl-hip/src/compute/server.rs (`launch` pushes onto the stream's `errors`, `command()` drains them into
//! `ServerError::ServerUnhealthy`).
use crate::ffi::*;
use cubecl_environment::backtrace::BackTrace;
use cubecl_runtime::server::{IoError, LaunchError, ResourceLimitError, ServerError};
use std::ffi::CStr;

pub(crate) fn last_message(ctx: *mut mi355_ctx) -> String {
    unsafe {
        let p = if ctx.is_null() { mi355_last_global_error() } else { mi355_last_error(ctx) };
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

/// One queued or immediate failure -> the reference's taxonomy (server/base.rs:177-332, :884-1019).
pub(crate) fn convert(code: i32, requested: u64, max: u64, message: String) -> ServerError {
    let bt = BackTrace::capture;
    match code {
        MI355_E_INVALID_ARGUMENT => ServerError::Validation { message, backtrace: bt() },
        MI355_E_OUT_OF_MEMORY => IoError::OutOfMemory { size: requested, backtrace: bt() }.into(),
        MI355_E_BUFFER_TOO_BIG => IoError::BufferTooBig { size: requested, backtrace: bt() }.into(),
        MI355_E_UNSUPPORTED_STRIDES => IoError::UnsupportedStrides { backtrace: bt() }.into(),
        MI355_E_NOT_FOUND => IoError::NotFound { backtrace: bt(), reason: message.into() }.into(),
        MI355_E_SHARED_MEMORY => LaunchError::TooManyResources(ResourceLimitError::SharedMemory {
            requested: requested as usize, max: max as usize, backtrace: bt() }).into(),
        MI355_E_UNITS => LaunchError::TooManyResources(ResourceLimitError::Units {
            requested: requested as u32, max: max as u32, backtrace: bt() }).into(),
        // the queue carries (requested, max) as two integers; the three extents travel in the message the library
        // formats ("... Requested (x, y, z), max is (a, b, c).", runtime.cpp mi355_launch)
        MI355_E_CUBE_DIM => match triples(&message) {
            Some((requested, max)) => LaunchError::TooManyResources(ResourceLimitError::CubeDim { requested, max, backtrace: bt() }).into(),
            None => LaunchError::Unknown { reason: message, backtrace: bt() }.into(),
        },
        MI355_E_MAX_UNITS_PER_CUBE => LaunchError::TooManyResources(ResourceLimitError::MaxUnitPerCube {
            requested: requested as u32, max: max as u32, backtrace: bt() }).into(),
        MI355_E_COMPILATION => LaunchError::CompilationError(
            cubecl_runtime::compiler::CompilationError::Generic { reason: message, backtrace: bt() }).into(),
        MI355_E_LAUNCH => LaunchError::Unknown { reason: message, backtrace: bt() }.into(),
        MI355_E_UNSUPPORTED => IoError::UnsupportedIoOperation { backtrace: bt() }.into(),
        MI355_E_PROFILE => cubecl_runtime::server::ProfileError::Unknown { reason: message, backtrace: bt() }.into(),
        // a failed stream / device synchronisation, a collective, a missing device: no closer variant than Generic
        MI355_E_EXECUTION | MI355_E_COMM | MI355_E_NO_DEVICE =>
            ServerError::Generic { reason: format!("mi355cube status {code}: {message}"), backtrace: bt() },
        _ => ServerError::Generic { reason: format!("mi355cube status {code}: {message}"), backtrace: bt() },
    }
}

/// The two "(x, y, z)" groups of a cube-dim message.
fn triples(message: &str) -> Option<((u32, u32, u32), (u32, u32, u32))> {
    let mut groups = message.split('(').skip(1).filter_map(|g| {
        let mut it = g.split(')').next()?.split(',').map(|v| v.trim().parse::<u32>());
        Some((it.next()?.ok()?, it.next()?.ok()?, it.next()?.ok()?))
    });
    Some((groups.next()?, groups.next()?))
}

/// `Ok(())` for `MI355_OK`; drains the error queue into `ServerUnhealthy { errors }` for status 14.
pub(crate) fn check(ctx: *mut mi355_ctx, rc: i32) -> Result<(), ServerError> {
    if rc == MI355_OK {
        return Ok(());
    }
    if rc != MI355_E_SERVER_UNHEALTHY {
        return Err(convert(rc, 0, 0, last_message(ctx)));
    }
    let mut errors = Vec::new();
    loop {
        let (mut code, mut req, mut max) = (0i32, 0u64, 0u64);
        let mut buf = [0 as core::ffi::c_char; 512];
        let rc = unsafe { mi355_error_pop(ctx, &mut code, &mut req, &mut max, buf.as_mut_ptr(), buf.len()) };
        if rc != MI355_OK {
            break; // MI355_E_NOT_FOUND: queue empty
        }
        let msg = unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
        errors.push(convert(code, req, max, msg));
    }
    Err(ServerError::ServerUnhealthy { errors, backtrace: BackTrace::capture() })
}

/// The `IoError` a memory call failed with (anything that is not an IO failure is wrapped, server/base.rs `IoError::Execution`).
pub(crate) fn io(ctx: *mut mi355_ctx, rc: i32, size: u64) -> IoError {
    match convert(rc, size, 0, last_message(ctx)) {
        ServerError::Io(err) => err,
        other => IoError::Execution(Box::new(other)),
    }
}
