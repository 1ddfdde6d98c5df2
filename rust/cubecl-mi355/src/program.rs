//! Generic `#[cube]` kernels: HIP C++ from the reference's dialect -> hiprtc -> code object -> `mi355_module_load`.
//! Only the hiprtc calls come from `cubecl-hip-sys`; the module and its launches belong to the library's context, so a
//! JIT kernel runs on the same streams and sees the same device pointers as the native ones.
use crate::{error, ffi::*};
use cubecl_environment::backtrace::BackTrace;
use cubecl_hip_sys::{
    get_hip_include_path, hiprtcCompileProgram, hiprtcCreateProgram, hiprtcDestroyProgram, hiprtcGetCode, hiprtcGetCodeSize,
    hiprtcGetProgramLog, hiprtcGetProgramLogSize, hiprtcProgram, hiprtcResult_HIPRTC_SUCCESS,
};
use cubecl_runtime::compiler::CompilationError;
use cubecl_runtime::server::{CubeDim, LaunchError};
use std::ffi::{CStr, CString};

/// A loaded entry point and what a launch of it needs.
#[derive(Debug, Clone, Copy)]
pub struct Program {
    pub function: mi355_function,
    pub cube_dim: CubeDim,
    pub shared_mem_bytes: usize,
}
unsafe impl Send for Program {}

fn failed(step: &str, detail: impl core::fmt::Display) -> CompilationError {
    CompilationError::Generic { reason: format!("{step}: {detail}"), backtrace: BackTrace::capture() }
}

/// Owns the hiprtc program for the duration of one compilation.
struct Rtc(hiprtcProgram);

impl Drop for Rtc {
    fn drop(&mut self) {
        unsafe { hiprtcDestroyProgram(&mut self.0) };
    }
}

impl Rtc {
    fn log(&self) -> String {
        let mut len = 0usize;
        if unsafe { hiprtcGetProgramLogSize(self.0, &mut len) } != hiprtcResult_HIPRTC_SUCCESS || len == 0 {
            return "(hiprtc kept no log)".into();
        }
        let mut text = vec![0 as core::ffi::c_char; len];
        if unsafe { hiprtcGetProgramLog(self.0, text.as_mut_ptr()) } != hiprtcResult_HIPRTC_SUCCESS {
            return "(the hiprtc log could not be read)".into();
        }
        unsafe { CStr::from_ptr(text.as_ptr()) }.to_string_lossy().lines().filter(|l| !l.is_empty()).collect::<Vec<_>>().join("\n    ")
    }
}

/// HIP C++ -> gfx950 code object.
pub fn compile_hip(source: &str) -> Result<Vec<core::ffi::c_char>, CompilationError> {
    let text = CString::new(source).map_err(|_| failed("kernel source", "contains a NUL byte"))?;
    let mut raw: hiprtcProgram = core::ptr::null_mut();
    let status = unsafe { hiprtcCreateProgram(&mut raw, text.as_ptr(), core::ptr::null(), 0, core::ptr::null_mut(), core::ptr::null_mut()) };
    if status != hiprtcResult_HIPRTC_SUCCESS {
        return Err(failed("hiprtcCreateProgram", status));
    }
    let program = Rtc(raw);

    let include = get_hip_include_path().unwrap(); // a ROCm install without headers cannot compile anything
    let flags = [CString::new("--std=c++17").unwrap(), CString::new(format!("-I{include}")).unwrap(), CString::new("-O3").unwrap(),
                 CString::new("--offload-arch=gfx950").unwrap()];
    let mut argv: Vec<*const core::ffi::c_char> = flags.iter().map(|f| f.as_ptr()).collect();
    let status = unsafe { hiprtcCompileProgram(program.0, argv.len() as i32, argv.as_mut_ptr()) };
    if status != hiprtcResult_HIPRTC_SUCCESS {
        return Err(failed("hiprtcCompileProgram", format!("status {status}\n    {}\n[source]\n{source}", program.log())));
    }

    let mut bytes = 0usize;
    let status = unsafe { hiprtcGetCodeSize(program.0, &mut bytes) };
    if status != hiprtcResult_HIPRTC_SUCCESS {
        return Err(failed("hiprtcGetCodeSize", status));
    }
    let mut image = vec![0 as core::ffi::c_char; bytes];
    let status = unsafe { hiprtcGetCode(program.0, image.as_mut_ptr()) };
    if status != hiprtcResult_HIPRTC_SUCCESS {
        return Err(failed("hiprtcGetCode", status));
    }
    Ok(image)
}

/// Code object -> entry point in the library's context.  The module stays loaded for the life of the context
/// (`mi355_ctx_destroy` unloads what is left), like the reference keeps its `hipModule_t`s.
pub fn load(ctx: *mut mi355_ctx, image: &[core::ffi::c_char], entrypoint: &str, cube_dim: CubeDim, shared_mem_bytes: usize) -> Result<Program, LaunchError> {
    let name = CString::new(entrypoint).map_err(|_| failed("entry point name", "contains a NUL byte"))?;
    let mut module: mi355_module = core::ptr::null_mut();
    let rc = unsafe { mi355_module_load(ctx, image.as_ptr() as *const core::ffi::c_void, image.len(), &mut module) };
    if rc != MI355_OK {
        return Err(failed("mi355_module_load", error::last_message(ctx)).into());
    }
    let mut function: mi355_function = core::ptr::null_mut();
    let rc = unsafe { mi355_module_get_function(ctx, module, name.as_ptr(), &mut function) };
    if rc != MI355_OK {
        let why = error::last_message(ctx);
        unsafe { mi355_module_unload(ctx, module) };
        return Err(failed("mi355_module_get_function", why).into());
    }
    Ok(Program { function, cube_dim, shared_mem_bytes })
}
