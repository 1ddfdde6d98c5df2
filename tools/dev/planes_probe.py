"""dev: the short-axis mover (few channels interleaved <-> planar) on 512 MiB tensors (GPU box)."""
import sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
client = Mi355Runtime.client()
ev = bench.Events(client)
names = ["flat", "rows", "transpose", "generic", "two_sided"]
H = W = 224
cases = (("u8 NHWC->NCHW c3", ElemType.U8, [3566, 3, H, W], [3 * H * W, 1, W * 3, 3]),
         ("u8 NCHW->NHWC c3", ElemType.U8, [3566, H, W, 3], [3 * H * W, W, 1, H * W]),
         ("u8 NHWC->NCHW c4", ElemType.U8, [2674, 4, H, W], [4 * H * W, 1, W * 4, 4]),
         ("bf16 NHWC->NCHW c3", ElemType.BF16, [1783, 3, H, W], [3 * H * W, 1, W * 3, 3]),
         ("bf16 NCHW->NHWC c3", ElemType.BF16, [1783, H, W, 3], [3 * H * W, W, 1, H * W]),
         ("f32 complex->split", ElemType.F32, [2, 1 << 26], [1, 2]),
         ("f32 NHWC->NCHW c3", ElemType.F32, [891, 3, H, W], [3 * H * W, 1, W * 3, 3]),
         ("f32 NCHW->NHWC c4", ElemType.F32, [668, H, W, 4], [4 * H * W, W, 1, H * W]),
         ("u8 NHWC->NCHW c3 odd W (fallback)", ElemType.U8, [3566, 3, H, 223], [3 * H * 223, 1, 223 * 3, 3]))
for name, dt, shape, strides in cases:
    n = 1
    for d_ in shape:
        n *= d_
    span = sum((d_ - 1) * s_ for d_, s_ in zip(shape, strides)) + 1
    src = client.empty(span * dt.size()); dst = client.empty(n * dt.size())
    tin = TensorHandle.new(src, shape, strides, dt); tout = TensorHandle.new_contiguous(shape, dst, dt)
    path, access = ops.copy_plan(client, tin, tout)
    med, best = bench.samples_op(client, ev, lambda: ops.copy_into(client, tin, tout), samples=9, warmup=3)
    print(f"{name:36s} {names[path]:9s} access {access:2d}  {med * 1e3:7.1f} us  {2 * n * dt.size() / med / 1e6:7.1f} GB/s", flush=True)
