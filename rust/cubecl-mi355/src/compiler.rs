//! `Mi355Compiler`: the `Compiler` (cubecl-runtime/src/compiler.rs `pub trait Compiler`) of `Mi355Runtime`.
//!
//! What it produces is a [`Mi355Kernel`]: either a [`NativeOp`] -- the descriptor of a kernel that already exists as
//! gfx950 machine code inside `libmi355cube.so`, produced by the tasks of [`crate::task`] without looking at any IR --
//! or the HIP C++ the reference's own dialect emits for a `KernelDefinition`, which [`crate::program`] hands to hiprtc.
//! `compile` itself only ever sees IR (it is what `KernelTask::compile` calls), so it always takes the second road.
use crate::ffi::mi355_gemm_desc;
use core::fmt::{self, Display};
use cubecl_cpp::{
    ComputeKernel,
    shared::{CompilationOptions, CppCompiler},
    target::Hip,
};
use cubecl_runtime::{
    compiler::{CompilationError, Compiler},
    kernel::KernelDefinition,
};

/// The element-wise sum / arg-max family of `include/mi355cube.h` ("Reductions").
#[derive(Debug, Clone, Copy, PartialEq, Eq, Hash)]
pub enum ReduceKind {
    /// `mi355_reduce_sum`: whole-buffer sum, f32 result.
    Sum,
    /// `mi355_argmax`: value and index of the first maximum.
    Argmax,
    /// `mi355_sum_argmax`: both in one pass over the input.
    SumArgmax,
    /// `mi355_reduce_last_axis_sum`: one f32 per row.
    RowSum,
    /// `mi355_reduce_last_axis_argmax`: one u32 per row.
    RowArgmax,
    /// `mi355_reduce` with a value operation (`MI355_REDUCE_SUM / MEAN / MAX / MIN / PROD`): whole-buffer, f32 result.
    Value(i32),
    /// `mi355_argreduce` with an index operation (`MI355_REDUCE_ARGMAX / ARGMIN`): value and index of the first extremum.
    Index(i32),
}

/// A launch the library serves with ahead-of-time code.  Shapes are part of the descriptor the way comptime arguments
/// are part of a `KernelId`: a different shape is a different kernel.
#[derive(Debug, Clone, Copy, PartialEq, Eq, Hash)]
pub enum NativeOp {
    /// `mi355_gemm(desc, a, b, c)`; buffers bound in that order.
    Gemm(GemmKey),
    /// `mi355_gemm_add(desc, a, b, c, d)`: `D = A * B + C`.
    GemmAdd(GemmKey),
    /// Buffers: input, then the outputs the kind has (sum / value / index), then the workspace for the whole-buffer kinds.
    Reduce { kind: ReduceKind, dtype: i32, rows: u64, cols: u64, row_stride: u64 },
    /// `mi355_reduce_axis` / `mi355_argreduce_axis` over one axis of a contiguous `[outer][reduce][inner]` view; buffers: input, output
    /// (f32 values for a value operation, u32 indices for `MI355_REDUCE_ARGMAX / ARGMIN`).
    ReduceAxis { op: i32, dtype: i32, outer: u64, reduce: u64, inner: u64 },
}

/// `mi355_gemm_desc` with `Eq + Hash` (the C struct is all integers).
#[derive(Debug, Clone, Copy, PartialEq, Eq, Hash)]
pub struct GemmKey {
    pub m: i64,
    pub n: i64,
    pub k: i64,
    pub batch: i64,
    pub lda: i64,
    pub ldb: i64,
    pub ldc: i64,
    pub stride_a: i64,
    pub stride_b: i64,
    pub stride_c: i64,
    pub dtype_ab: i32,
    pub dtype_c: i32,
    pub trans_a: bool,
    pub trans_b: bool,
}

impl GemmKey {
    pub fn desc(&self) -> mi355_gemm_desc {
        mi355_gemm_desc {
            m: self.m,
            n: self.n,
            k: self.k,
            batch: self.batch,
            lda: self.lda,
            ldb: self.ldb,
            ldc: self.ldc,
            stride_a: self.stride_a,
            stride_b: self.stride_b,
            stride_c: self.stride_c,
            dtype_ab: self.dtype_ab,
            dtype_c: self.dtype_c,
            trans_a: self.trans_a as i32,
            trans_b: self.trans_b as i32,
            algo: 0, // MI355_GEMM_ALGO_AUTO: the library's own selection, as benched
            reserved: 0,
        }
    }
}

pub enum Mi355Kernel {
    Native(NativeOp),
    Hip(ComputeKernel),
}

impl Display for Mi355Kernel {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            Mi355Kernel::Native(op) => write!(f, "// ahead-of-time gfx950 kernel in libmi355cube.so\n// {op:?}\n"),
            Mi355Kernel::Hip(kernel) => Display::fmt(kernel, f),
        }
    }
}

#[derive(Clone, Debug, Default)]
pub struct Mi355Compiler {
    cpp: CppCompiler<Hip>,
}

impl Compiler for Mi355Compiler {
    type Representation = Mi355Kernel;
    type CompilationOptions = CompilationOptions;

    fn compile(
        &mut self,
        kernel: KernelDefinition,
        compilation_options: &Self::CompilationOptions,
    ) -> Result<Self::Representation, CompilationError> {
        self.cpp.compile(kernel, compilation_options).map(Mi355Kernel::Hip)
    }

    fn extension(&self) -> &'static str {
        "cpp"
    }
}
