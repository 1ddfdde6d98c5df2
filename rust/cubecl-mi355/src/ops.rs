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

/// Value (f32) and index (u64) of the first maximum (`mi355_argmax`; ties go to the lowest index, NaN never wins).
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
