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

/// out = lhs * rhs + acc, formed in f32 and rounded once (the C operand of `cmma::execute`,
/// crates/cubecl-core/src/frontend/cmma.rs:1066-1110).  `acc` has out's shape, strides and dtype and may be `out`.
pub fn matmul_add(client: &ComputeClient<Mi355Runtime>, lhs: &TensorHandle<Mi355Runtime>, rhs: &TensorHandle<Mi355Runtime>,
                  acc: &TensorHandle<Mi355Runtime>, out: &TensorHandle<Mi355Runtime>) -> Result<(), ServerError> {
    let r = out.shape().len();
    let (m, n, k) = (out.shape()[r - 2] as i64, out.shape()[r - 1] as i64, lhs.shape()[lhs.shape().len() - 1] as i64);
    let batch: i64 = out.shape()[..r - 2].iter().map(|d| *d as i64).product();
    let (ta, lda, sa) = operand(lhs)?;
    let (tb, ldb, sb) = operand(rhs)?;
    let (_, ldc, sc) = operand(out)?;
    if acc.shape() != out.shape() || acc.strides() != out.strides() || dtype(acc) != dtype(out) {
        return Err(ServerError::Validation { message: "accumulator must have the output's shape, strides and dtype".into(),
                                             backtrace: cubecl_common::backtrace::BackTrace::capture() });
    }
    let desc = mi355_gemm_desc { m, n, k, batch, lda, ldb, ldc, stride_a: sa, stride_b: sb, stride_c: sc,
                                 dtype_ab: dtype(lhs), dtype_c: dtype(out), trans_a: ta as i32, trans_b: tb as i32,
                                 algo: 0, reserved: 0 };
    let (a, b, c, d) = (lhs.handle.clone().binding(), rhs.handle.clone().binding(), acc.handle.clone().binding(),
                        out.handle.clone().binding());
    client.with_server(move |server, stream| {
        let (pa, pb) = (server.get_resource(a, stream)?, server.get_resource(b, stream)?);
        let (pc, pd) = (server.get_resource(c, stream)?, server.get_resource(d, stream)?);
        check(server.ctx, unsafe { mi355_gemm_add(server.ctx, core::ptr::null_mut(), &desc, pa.resource().ptr, pb.resource().ptr,
                                                  pc.resource().ptr, pd.resource().ptr) })
    })
}

/// Array-wide sum AND argmax in one pass over the data (config C4's local step): out_sum[0] f32, out_index[0] u64.
pub fn sum_argmax(client: &ComputeClient<Mi355Runtime>, input: &TensorHandle<Mi355Runtime>, out_sum: &TensorHandle<Mi355Runtime>,
                  out_index: &TensorHandle<Mi355Runtime>) -> Result<(), ServerError> {
    let n: u64 = input.shape().iter().map(|d| *d as u64).product();
    let dt = dtype(input);
    let ws = client.empty(workspace_bytes(client, n)? as usize);
    let (i, s, o, w) = (input.handle.clone().binding(), out_sum.handle.clone().binding(), out_index.handle.clone().binding(),
                        ws.clone().binding());
    client.with_server(move |server, stream| {
        let (pi, ps) = (server.get_resource(i, stream)?, server.get_resource(s, stream)?);
        let (po, pw) = (server.get_resource(o, stream)?, server.get_resource(w, stream)?);
        check(server.ctx, unsafe { mi355_sum_argmax(server.ctx, core::ptr::null_mut(), pi.resource().ptr, dt, n, ps.resource().ptr as _,
                                                    core::ptr::null_mut(), po.resource().ptr as _, pw.resource().ptr,
                                                    pw.resource().size) })
    })
}

/// Sum over one axis of a contiguous tensor (the book's `reduce_matrix` generalised, cubecl-book getting-started
/// v4-gpu.rs:47-70): out has the input's shape without `axis`, f32.
pub fn reduce_sum_axis(client: &ComputeClient<Mi355Runtime>, input: &TensorHandle<Mi355Runtime>, out: &TensorHandle<Mi355Runtime>,
                       axis: usize) -> Result<(), ServerError> {
    let shape = input.shape();
    let outer: u64 = shape[..axis].iter().map(|d| *d as u64).product();
    let (reduce, inner): (u64, u64) = (shape[axis] as u64, shape[axis + 1..].iter().map(|d| *d as u64).product());
    let dt = dtype(input);
    let (i, o) = (input.handle.clone().binding(), out.handle.clone().binding());
    client.with_server(move |server, stream| {
        let (pi, po) = (server.get_resource(i, stream)?, server.get_resource(o, stream)?);
        check(server.ctx, unsafe { mi355_reduce_axis_sum(server.ctx, core::ptr::null_mut(), pi.resource().ptr, dt,
                                                         po.resource().ptr as _, outer, reduce, inner) })
    })
}

fn layout(t: &TensorHandle<Mi355Runtime>) -> mi355_tensor_layout {
    let mut l = mi355_tensor_layout { rank: t.shape().len() as i32, reserved: 0, shape: [0; 8], strides: [0; 8] };
    for (d, (n, s)) in t.shape().iter().zip(t.strides().iter()).enumerate() {
        l.shape[d] = *n as i64;
        l.strides[d] = *s as i64;
    }
    l
}

/// `copy_into` (crates/cubecl-std/src/tensor/contiguous/base.rs): output[idx] = input[idx] for two views of one shape;
/// the library picks the mover (flat, rows, LDS-transposed tiles, packed gather / scatter) from the two layouts.
pub fn copy_into(client: &ComputeClient<Mi355Runtime>, input: &TensorHandle<Mi355Runtime>,
                 output: &TensorHandle<Mi355Runtime>) -> Result<(), ServerError> {
    let (li, lo, esz) = (layout(input), layout(output), input.dtype.size() as i32);
    let (i, o) = (input.handle.clone().binding(), output.handle.clone().binding());
    client.with_server(move |server, stream| {
        let (pi, po) = (server.get_resource(i, stream)?, server.get_resource(o, stream)?);
        check(server.ctx, unsafe { mi355_copy_strided(server.ctx, core::ptr::null_mut(), pi.resource().ptr, &li, po.resource().ptr,
                                                      &lo, esz) })
    })
}

/// `into_contiguous`: a fresh row-major tensor with the input's values (what a matmul front-end calls on a
/// `MatrixBatchLayout::HighlyPermuted` operand before `matmul`).
pub fn into_contiguous(client: &ComputeClient<Mi355Runtime>, input: &TensorHandle<Mi355Runtime>)
                       -> Result<TensorHandle<Mi355Runtime>, ServerError> {
    let n: usize = input.shape().iter().product();
    let out = TensorHandle::new_contiguous(input.shape().clone(), client.empty(n * input.dtype.size()), input.dtype);
    copy_into(client, input, &out)?;
    Ok(out)
}

/// `tensor::identity::launch` (crates/cubecl-std/src/tensor/identity.rs:36-86) on a [dim, dim] tensor with unit column stride.
pub fn identity(client: &ComputeClient<Mi355Runtime>, output: &TensorHandle<Mi355Runtime>) -> Result<(), ServerError> {
    let (dim, ld, dt) = (output.shape()[0] as u64, output.strides()[0] as u64, dtype(output));
    let o = output.handle.clone().binding();
    client.with_server(move |server, stream| {
        let po = server.get_resource(o, stream)?;
        check(server.ctx, unsafe { mi355_fill_identity(server.ctx, core::ptr::null_mut(), po.resource().ptr, dt, dim, ld) })
    })
}

fn workspace_bytes(client: &ComputeClient<Mi355Runtime>, n: u64) -> Result<u64, ServerError> {
    client.with_server(move |server, _| {
        let mut bytes = 0u64;
        check(server.ctx, unsafe { mi355_reduce_workspace_bytes(server.ctx, n, &mut bytes) }).map(|_| bytes)
    })
}
