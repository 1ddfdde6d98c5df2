//! `cubecl-mi355` -- CubeCL backend glue for the MI355X-native matmul / reduction library.
//!
//! SOURCE ONLY: never compiled in the repository's build image (no Rust toolchain there).  It is
//! the thin layer `INTEGRATION.md` describes: every trait operation of the reference's backend
//! surface forwards to one entry point of `include/mi355cube.h`; nothing here computes.
//!
//! Layout
//! * [`ffi`]      raw `extern "C"` declarations (1:1 with the header)
//! * [`error`]    `MI355_E_*` -> `ServerError` / `LaunchError` / `IoError` / `ResourceLimitError`
//! * [`storage`]  `ComputeStorage` over `mi355_alloc` / `mi355_free` / `mi355_flush`
//! * [`server`]   `Mi355Server`: `ComputeServer` + `ServerCommunication` + `DeviceService`
//! * [`runtime`]  `Mi355Runtime`: `Runtime`
//! * [`ops`]      `matmul` / `reduce_sum` / `argmax` launchers over `TensorHandle` (what cubek's
//!                launchers call into; they bypass `Compiler`/`CubeTask` entirely)
pub mod error;
pub mod ffi;
pub mod ops;
pub mod runtime;
pub mod server;
pub mod storage;

pub use runtime::{Mi355Device, Mi355Runtime};
pub use server::Mi355Server;
