//! The hot-path launchers: what a matmul / reduce front-end (cubek) calls on this backend instead of
//! expanding `#[cube]` kernels.  Operands are `TensorHandle`s (crates/cubecl-std/src/tensor/handle.rs:13-23):
//! layout comes from strides alone, classified the way `matrix_batch_layout` does
//! (crates/cubecl-std/src/tensor/matrix_batch_layout.rs:21-79).
use crate::{error::check, ffi::*, Mi355Runtime};
use cubecl_runtime::{client::ComputeClient, server::ServerError};
use cubecl_std::tensor::{matrix_batch_layout, MatrixBatchLayout, TensorHandle};

fn dtype(t: &TensorHandle<Mi355Runtime>) -> i32 {
    use cubecl_ir::{ElemType::Float, FloatKind::*};
    match t.dtype.elem_type() { Float(BF16) => MI355_DTYPE_BF16, Float(F16) => MI355_DTYPE_F16, _ => MI355_DTYPE_F32 }
}

/// (transposed, leading dimension, batch stride) of a [.., rows, cols] operand.
fn operand(t: &TensorHandle<Mi355Runtime>) -> Result<(bool, i64, i64), ServerError> {
    let r = t.shape().len();
    let (rs, cs) = (t.strides()[r - 2] as i64, t.strides()[r - 1] as i64);
    match matrix_batch_layout(t.strides(), None) {
        MatrixBatchLayout::HighlyPermuted => Err(ServerError::Validation {
            message: "operand is HighlyPermuted: call into_contiguous first".into(),
            backtrace: cubecl_common::backtrace::BackTrace::capture() }),
        _ => {
            let bstride = if r > 2 { t.strides()[r - 3] as i64 } else { 0 };
            Ok(if cs == 1 { (false, rs, bstride) } else { (true, cs, bstride) })
        }
    }
}

/// out[.., m, n] = sum_k lhs[.., m, k] * rhs[.., k, n], f32 accumulate (cmma::execute semantics,
/// crates/cubecl-core/src/frontend/cmma.rs:1066-1110).  Stream-ordered, fire-and-forget like `launch`.
pub fn matmul(client: &ComputeClient<Mi355Runtime>, lhs: &TensorHandle<Mi355Runtime>, rhs: &TensorHandle<Mi355Runtime>,
              out: &TensorHandle<Mi355Runtime>) -> Result<(), ServerError> {
    let r = out.shape().len();
    let (m, n, k) = (out.shape()[r - 2] as i64, out.shape()[r - 1] as i64, lhs.shape()[lhs.shape().len() - 1] as i64);
    let batch: i64 = out.shape()[..r - 2].iter().map(|d| *d as i64).product();
    let (ta, lda, sa) = operand(lhs)?;
    let (tb, ldb, sb) = operand(rhs)?;
    let (_, ldc, sc) = operand(out)?;
    let desc = mi355_gemm_desc { m, n, k, batch, lda, ldb, ldc, stride_a: sa, stride_b: sb, stride_c: sc,
                                 dtype_ab: dtype(lhs), dtype_c: dtype(out), trans_a: ta as i32, trans_b: tb as i32,
                                 algo: 0, reserved: 0 };
    let (a, b, c) = (lhs.handle.clone().binding(), rhs.handle.clone().binding(), out.handle.clone().binding());
    client.with_server(move |server, stream| {     // runs on the device's runner thread
        let (pa, pb, pc) = (server.get_resource(a, stream)?, server.get_resource(b, stream)?, server.get_resource(c, stream)?);
        check(server.ctx, unsafe { mi355_gemm(server.ctx, core::ptr::null_mut(), &desc, pa.resource().ptr, pb.resource().ptr,
                                              pc.resource().ptr) })
    })
}

/// Array-wide f32 sum into out[0]; deterministic tree, one launch.
pub fn reduce_sum(client: &ComputeClient<Mi355Runtime>, input: &TensorHandle<Mi355Runtime>,
                  out: &TensorHandle<Mi355Runtime>) -> Result<(), ServerError> {
    let n: u64 = input.shape().iter().map(|d| *d as u64).product();
    let ws = client.empty(workspace_bytes(client, n)? as usize);
    let (i, o, w) = (input.handle.clone().binding(), out.handle.clone().binding(), ws.clone().binding());
    client.with_server(move |server, stream| {
        let (pi, po, pw) = (server.get_resource(i, stream)?, server.get_resource(o, stream)?, server.get_resource(w, stream)?);
        check(server.ctx, unsafe { mi355_reduce_sum_f32(server.ctx, core::ptr::null_mut(), pi.resource().ptr as _, n,
                                                        po.resource().ptr as _, pw.resource().ptr, pw.resource().size) })
    })
}

/// Array-wide argmax: out_index[0] (u64) = lowest index of the maximum; NaN ranks highest, -0 == +0.
pub fn argmax(client: &ComputeClient<Mi355Runtime>, input: &TensorHandle<Mi355Runtime>,
              out_index: &TensorHandle<Mi355Runtime>) -> Result<(), ServerError> {
    let n: u64 = input.shape().iter().map(|d| *d as u64).product();
    let ws = client.empty(workspace_bytes(client, n)? as usize);
    let (i, o, w) = (input.handle.clone().binding(), out_index.handle.clone().binding(), ws.clone().binding());
    client.with_server(move |server, stream| {
        let (pi, po, pw) = (server.get_resource(i, stream)?, server.get_resource(o, stream)?, server.get_resource(w, stream)?);
        check(server.ctx, unsafe { mi355_argmax_f32(server.ctx, core::ptr::null_mut(), pi.resource().ptr as _, n,
                                                    core::ptr::null_mut(), po.resource().ptr as _, pw.resource().ptr,
                                                    pw.resource().size) })
    })
}

fn workspace_bytes(client: &ComputeClient<Mi355Runtime>, n: u64) -> Result<u64, ServerError> {
    client.with_server(move |server, _| {
        let mut bytes = 0u64;
        check(server.ctx, unsafe { mi355_reduce_workspace_bytes(server.ctx, n, &mut bytes) }).map(|_| bytes)
    })
}
