//! The hot path as `CubeTask`s (cubecl-runtime/src/compiler.rs `pub trait CubeTask`): what `ComputeClient::launch`
//! carries to `Mi355Server::launch` for a matmul or a reduction.  `compile` does not lower anything -- the kernel is
//! already machine code inside the library -- it just wraps the descriptor as the compiled representation; `define` has
//! to exist for the trait and returns an empty root scope that nothing reads.
use crate::compiler::{GemmKey, Mi355Compiler, Mi355Kernel, NativeOp};
use cubecl_ir::{
    AddressType, ElemType, Scope, UIntKind,
    metadata::Info,
    settings::{Dim3, ExecutionMode, KernelSettings},
};
use cubecl_runtime::{
    compiler::{CompilationError, Compiler, CubeTask},
    id::KernelId,
    kernel::{CompiledKernel, KernelDefinition, KernelMetadata},
    server::CubeDim,
};

/// A native launch; `op` is also the `KernelId` info, so two different shapes never share a cache entry.
#[derive(Debug, Clone, Copy)]
pub struct NativeTask {
    pub op: NativeOp,
}

pub type GemmTask = NativeTask;
pub type ReduceTask = NativeTask;

impl NativeTask {
    pub fn gemm(key: GemmKey) -> Self {
        Self { op: NativeOp::Gemm(key) }
    }

    pub fn gemm_add(key: GemmKey) -> Self {
        Self { op: NativeOp::GemmAdd(key) }
    }

    pub fn entrypoint(&self) -> &'static str {
        match self.op {
            NativeOp::Gemm(_) => "mi355_gemm",
            NativeOp::GemmAdd(_) => "mi355_gemm_add",
            NativeOp::Reduce { .. } => "mi355_reduce",
            NativeOp::ReduceAxis { .. } => "mi355_reduce_axis",
        }
    }
}

impl KernelMetadata for NativeTask {
    fn name(&self) -> &'static str {
        self.entrypoint()
    }

    fn id(&self) -> KernelId {
        KernelId::new::<Self>().info(self.op)
    }

    fn address_type(&self) -> ElemType {
        ElemType::UInt(UIntKind::U64)
    }
}

impl CubeTask<Mi355Compiler> for NativeTask {
    fn define(&self) -> KernelDefinition {
        let settings = KernelSettings::new(Dim3::new_single(), ExecutionMode::Unchecked, AddressType::U64);
        KernelDefinition { body: Scope::root(settings.clone()), info: Info::default(), settings }
    }

    fn compile(
        &self,
        _definition: KernelDefinition,
        _compiler: &mut Mi355Compiler,
        _compilation_options: &<Mi355Compiler as Compiler>::CompilationOptions,
    ) -> Result<CompiledKernel<Mi355Compiler>, CompilationError> {
        Ok(CompiledKernel {
            entrypoint_name: self.entrypoint().into(),
            debug_name: Some(self.entrypoint()),
            source: String::new(),
            repr: Some(Mi355Kernel::Native(self.op)),
            cube_dim: CubeDim::new_single(),
            debug_info: None,
        })
    }
}
