//! `cubecl-mi355` -- CubeCL backend glue for the MI355X-native matmul / reduction library.
//!
//! SOURCE ONLY: never compiled in the repository's build image (no Rust toolchain there); see
//! `tests/test_rust_shim_symbols.py` for what is checked instead.  Nothing in this crate computes: each trait operation
//! of the reference's backend surface forwards to entry points of `include/mi355cube.h`.
//!
//! Two kinds of kernel reach [`server::Mi355Server::launch`]:
//! * the hot path -- [`task::GemmTask`], [`task::ReduceTask`]: `CubeTask`s whose "compiled" form is a descriptor of an
//!   ahead-of-time gfx950 kernel inside `libmi355cube.so` (`mi355_gemm`, `mi355_reduce_sum`, ...);
//! * everything else -- any `#[cube]` kernel: lowered by the reference's own C++ dialect (`CppCompiler<Hip>`), compiled
//!   by hiprtc, then loaded and launched through `mi355_module_load` / `mi355_launch`, so that `testgen_all!` has a
//!   runtime to run on.
//!
//! Layout
//! * [`ffi`]      raw `extern "C"` declarations (1:1 with the header)
//! * [`error`]    `MI355_E_*` -> `ServerError` / `LaunchError` / `IoError` / `ResourceLimitError`
//! * [`storage`]  `ComputeStorage` for device memory (`mi355_alloc`) and pinned host memory (`mi355_pinned_alloc`)
//! * [`fence`]    one-shot event over `mi355_event_*`
//! * [`lane`]     `EventStreamBackend`: one `mi355_stream` + its memory pools per logical `StreamId`
//! * [`comm`]     `ServerCommunication` over RCCL (`mi355_comm_*`, `mi355_all_reduce`, `mi355_send` / `mi355_recv`)
//! * [`compiler`] `Mi355Compiler`: `Compiler` whose representation is either a native descriptor or HIP C++
//! * [`task`]     the native `CubeTask`s
//! * [`program`]  hiprtc -> code object -> `mi355_module_load`
//! * [`server`]   `Mi355Server`: `ComputeServer` + `ServerCommunication` + `DeviceService`
//! * [`runtime`]  `Mi355Runtime`: `Runtime`, `Mi355Device`: `Device`
//! * [`ops`]      `matmul` / `reduce_sum` / `argmax` launchers over `ComputeClient::launch`
#[macro_use]
extern crate alloc;

pub mod comm;
pub mod compiler;
pub mod error;
pub mod fence;
pub mod ffi;
pub mod lane;
pub mod ops;
pub mod program;
pub mod runtime;
pub mod server;
pub mod storage;
pub mod task;

pub use runtime::{Mi355Device, Mi355Runtime};
pub use server::Mi355Server;

/// The reference's conformance suites instantiated on this runtime, exactly the three macro calls of
/// crates/cubecl-hip/src/lib.rs:12-20 (the macros read `TestRuntime` from the enclosing module).
#[cfg(test)]
#[allow(unexpected_cfgs)]
mod tests {
    pub type TestRuntime = crate::Mi355Runtime;

    pub use half::{bf16, f16};

    cubecl_std::testgen!();
    cubecl_core::testgen_all!(f32: [f16, bf16, f32], i32: [i16, i32], u32: [u16, u32]);
    cubecl_core::testgen_launch_dynamic_count!();
}
