//! The hot-path launchers: what a matmul / reduce front-end calls on this backend instead of expanding `#[cube]`
//! kernels.  Operands are `TensorHandle`s (cubecl-std/src/tensor/handle.rs); a launch is a `NativeTask` handed to
//! `ComputeClient::launch` with the operands as buffer bindings, so it is stream-ordered, fire-and-forget and
//! graph-capturable exactly like any other kernel of the reference (errors surface on the next `flush` / `sync`).
use crate::{
    Mi355Runtime,
    compiler::{GemmKey, NativeOp, ReduceKind},
    ffi::*,
    task::NativeTask,
};
use cubecl_environment::backtrace::BackTrace;
use cubecl_ir::{ElemType, FloatKind, IntKind, UIntKind};
use cubecl_runtime::{
    client::ComputeClient,
    server::{CubeCount, Handle, KernelArguments, ServerError},
};
use cubecl_std::tensor::{MatrixBatchLayout, TensorHandle, matrix_batch_layout};

type Tensor = TensorHandle<Mi355Runtime>;

fn refuse(message: impl Into<String>) -> ServerError {
    ServerError::Validation { message: message.into(), backtrace: BackTrace::capture() }
}

fn wire(dtype: ElemType) -> Result<i32, ServerError> {
    Ok(match dtype {
        ElemType::Float(FloatKind::F32) => MI355_DTYPE_F32,
        ElemType::Float(FloatKind::BF16) => MI355_DTYPE_BF16,
        ElemType::Float(FloatKind::F16) => MI355_DTYPE_F16,
        ElemType::Float(FloatKind::E4M3) => MI355_DTYPE_F8E4M3,
        ElemType::Float(FloatKind::E5M2) => MI355_DTYPE_F8E5M2,
        ElemType::Int(IntKind::I32) => MI355_DTYPE_I32,
        ElemType::UInt(UIntKind::U32) => MI355_DTYPE_U32,
        other => return Err(refuse(format!("{other:?} is not an operand type of the native kernels"))),
    })
}

/// `(transposed, leading dimension, batch stride)` of a `[.., rows, cols]` operand, read from its strides alone.
fn operand(t: &Tensor) -> Result<(bool, i64, i64), ServerError> {
    let rank = t.shape().len();
    if rank < 2 {
        return Err(refuse("a matmul operand needs at least two axes"));
    }
    if matches!(matrix_batch_layout(t.strides(), None), MatrixBatchLayout::HighlyPermuted) {
        return Err(refuse("operand is HighlyPermuted: make it contiguous first"));
    }
    let (row, col) = (t.strides()[rank - 2] as i64, t.strides()[rank - 1] as i64);
    let batch = if rank > 2 { t.strides()[rank - 3] as i64 } else { 0 };
    Ok(if col == 1 { (false, row, batch) } else { (true, col, batch) })
}

fn buffers(handles: &[&Handle]) -> KernelArguments {
    KernelArguments::new().with_buffers(handles.iter().map(|h| (*h).clone().binding()).collect())
}

/// `out[.., m, n] = sum_k lhs[.., m, k] * rhs[.., k, n]`, f32 accumulation (`mi355_gemm`).
pub fn matmul(client: &ComputeClient<Mi355Runtime>, lhs: &Tensor, rhs: &Tensor, out: &Tensor) -> Result<(), ServerError> {
    let rank = out.shape().len();
    let (trans_a, lda, stride_a) = operand(lhs)?;
    let (trans_b, ldb, stride_b) = operand(rhs)?;
    let (trans_c, ldc, stride_c) = operand(out)?;
    if trans_c {
        return Err(refuse("the output of a matmul must be row-major"));
    }
    let key = GemmKey {
        m: out.shape()[rank - 2] as i64,
        n: out.shape()[rank - 1] as i64,
        k: lhs.shape()[lhs.shape().len() - 1] as i64,
        batch: out.shape()[..rank - 2].iter().map(|d| *d as i64).product(),
        lda,
        ldb,
        ldc,
        stride_a,
        stride_b,
        stride_c,
        dtype_ab: wire(lhs.dtype)?,
        dtype_c: wire(out.dtype)?,
        trans_a,
        trans_b,
    };
    client.launch(Box::new(NativeTask::gemm(key)), CubeCount::new_single(), buffers(&[&lhs.handle, &rhs.handle, &out.handle]));
    Ok(())
}

fn whole(kind: ReduceKind, input: &Tensor) -> Result<(NativeTask, u64), ServerError> {
    let n: u64 = input.shape().iter().map(|d| *d as u64).product();
    let op = NativeOp::Reduce { kind, dtype: wire(input.dtype)?, rows: 1, cols: n, row_stride: n };
    Ok((NativeTask { op }, n))
}

/// Scratch the whole-buffer reductions need (`mi355_reduce_workspace_bytes`: a function of `n` only).
fn workspace(client: &ComputeClient<Mi355Runtime>, n: u64) -> Result<Handle, ServerError> {
    let mut bytes = 0u64;
    // no context needed for the size query: the layout of the partials is fixed by the kernel, not by the device
    crate::error::check(core::ptr::null_mut(), unsafe { mi355_reduce_workspace_bytes(core::ptr::null_mut(), n, &mut bytes) })?;
    Ok(client.empty(bytes as usize))
}

/// Sum of every element into `out[0]` (f32), one deterministic tree (`mi355_reduce_sum`).
pub fn reduce_sum(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out: &Tensor) -> Result<(), ServerError> {
    let (task, n) = whole(ReduceKind::Sum, input)?;
    let scratch = workspace(client, n)?;
    client.launch(Box::new(task), CubeCount::new_single(), buffers(&[&input.handle, &out.handle, &scratch]));
    Ok(())
}

/// Value (f32) and index (u64) of the first maximum (`mi355_argmax`; ties go to the lowest index, -0 == +0, a NaN ranks ABOVE every
/// number and the first NaN wins -- `include/mi355cube.h` "Reductions", numpy's rule).
pub fn argmax(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out_value: &Tensor, out_index: &Tensor) -> Result<(), ServerError> {
    let (task, n) = whole(ReduceKind::Argmax, input)?;
    let scratch = workspace(client, n)?;
    client.launch(Box::new(task), CubeCount::new_single(), buffers(&[&input.handle, &out_value.handle, &out_index.handle, &scratch]));
    Ok(())
}

/// Both of the above in one pass over the input (`mi355_sum_argmax`).
pub fn sum_argmax(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out_sum: &Tensor, out_value: &Tensor, out_index: &Tensor) -> Result<(), ServerError> {
    let (task, n) = whole(ReduceKind::SumArgmax, input)?;
    let scratch = workspace(client, n)?;
    client.launch(Box::new(task), CubeCount::new_single(),
                  buffers(&[&input.handle, &out_sum.handle, &out_value.handle, &out_index.handle, &scratch]));
    Ok(())
}

/// The value operations of `mi355_reduce` / `mi355_reduce_axis` (plane-level forms: `crates/cubecl-core/src/frontend/plane.rs:218`
/// sum, `:285` prod, `:352` max, `:370` min; `ReduceOperation::Mean`: `crates/cubecl-runtime/src/server/base.rs:623-628`).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum ValueOp {
    Sum,
    Mean,
    Max,
    Min,
    Prod,
}

impl ValueOp {
    fn code(self) -> i32 {
        match self {
            ValueOp::Sum => MI355_REDUCE_SUM,
            ValueOp::Mean => MI355_REDUCE_MEAN,
            ValueOp::Max => MI355_REDUCE_MAX,
            ValueOp::Min => MI355_REDUCE_MIN,
            ValueOp::Prod => MI355_REDUCE_PROD,
        }
    }
}

/// The index operations of `mi355_argreduce` / `mi355_argreduce_axis`: lowest index of the extremum, `-0 == +0`, the first NaN wins.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum IndexOp {
    ArgMax,
    ArgMin,
}

impl IndexOp {
    fn code(self) -> i32 {
        match self {
            IndexOp::ArgMax => MI355_REDUCE_ARGMAX,
            IndexOp::ArgMin => MI355_REDUCE_ARGMIN,
        }
    }
}

/// Any value reduction of every element into `out[0]` (f32) (`mi355_reduce`; max / min: NaN if any element is NaN, `-0 < +0`).
pub fn reduce(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out: &Tensor, op: ValueOp) -> Result<(), ServerError> {
    let (task, n) = whole(ReduceKind::Value(op.code()), input)?;
    let scratch = workspace(client, n)?;
    client.launch(Box::new(task), CubeCount::new_single(), buffers(&[&input.handle, &out.handle, &scratch]));
    Ok(())
}

/// Value (f32) and index (u64) of the first extremum (`mi355_argreduce`).
pub fn argreduce(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out_value: &Tensor, out_index: &Tensor, op: IndexOp) -> Result<(), ServerError> {
    let (task, n) = whole(ReduceKind::Index(op.code()), input)?;
    let scratch = workspace(client, n)?;
    client.launch(Box::new(task), CubeCount::new_single(), buffers(&[&input.handle, &out_value.handle, &out_index.handle, &scratch]));
    Ok(())
}

/// `[outer][reduce][inner]` of a contiguous tensor around `axis`.
fn around(input: &Tensor, axis: usize) -> Result<(u64, u64, u64), ServerError> {
    let shape = input.shape();
    if axis >= shape.len() {
        return Err(refuse("axis out of range"));
    }
    let mut expect = 1usize;
    for d in (0..shape.len()).rev() {
        if shape[d] != 1 && input.strides()[d] != expect {
            return Err(refuse("axis reductions need a contiguous tensor"));
        }
        expect *= shape[d];
    }
    let outer: u64 = shape[..axis].iter().map(|d| *d as u64).product();
    let inner: u64 = shape[axis + 1..].iter().map(|d| *d as u64).product();
    Ok((outer, shape[axis] as u64, inner))
}

/// A value reduction over one axis: `out` has the input's shape minus that axis, f32 (`mi355_reduce_axis`).
pub fn reduce_axis(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out: &Tensor, axis: usize, op: ValueOp) -> Result<(), ServerError> {
    let (outer, reduce, inner) = around(input, axis)?;
    let op = NativeOp::ReduceAxis { op: op.code(), dtype: wire(input.dtype)?, outer, reduce, inner };
    client.launch(Box::new(NativeTask { op }), CubeCount::new_single(), buffers(&[&input.handle, &out.handle]));
    Ok(())
}

/// An index reduction over one axis: u32 indices along it (`mi355_argreduce_axis`).
pub fn argreduce_axis(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out: &Tensor, axis: usize, op: IndexOp) -> Result<(), ServerError> {
    let (outer, reduce, inner) = around(input, axis)?;
    let op = NativeOp::ReduceAxis { op: op.code(), dtype: wire(input.dtype)?, outer, reduce, inner };
    client.launch(Box::new(NativeTask { op }), CubeCount::new_single(), buffers(&[&input.handle, &out.handle]));
    Ok(())
}

/// Per-row sum of a `[rows, cols]` tensor with unit column stride (`mi355_reduce_last_axis_sum`).
pub fn reduce_sum_rows(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out: &Tensor) -> Result<(), ServerError> {
    rows(client, ReduceKind::RowSum, input, out)
}

/// Per-row arg-max (u32 indices) (`mi355_reduce_last_axis_argmax`).
pub fn argmax_rows(client: &ComputeClient<Mi355Runtime>, input: &Tensor, out: &Tensor) -> Result<(), ServerError> {
    rows(client, ReduceKind::RowArgmax, input, out)
}

fn rows(client: &ComputeClient<Mi355Runtime>, kind: ReduceKind, input: &Tensor, out: &Tensor) -> Result<(), ServerError> {
    let rank = input.shape().len();
    if rank < 1 || input.strides()[rank - 1] != 1 {
        return Err(refuse("row reductions need a unit stride on the reduced axis"));
    }
    let cols = input.shape()[rank - 1] as u64;
    let rows: u64 = input.shape()[..rank - 1].iter().map(|d| *d as u64).product();
    let row_stride = if rank > 1 { input.strides()[rank - 2] as u64 } else { cols };
    let op = NativeOp::Reduce { kind, dtype: wire(input.dtype)?, rows, cols, row_stride };
    client.launch(Box::new(NativeTask { op }), CubeCount::new_single(), buffers(&[&input.handle, &out.handle]));
    Ok(())
}
