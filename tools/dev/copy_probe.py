"""dev: copy_into movers, GB/s (read + write) per layout and element size."""
import sys
sys.path.insert(0, ".")
import bench
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops
from cubecl_amd import _native as N
client = Mi355Runtime.client()
ev = bench.Events(client)
NAMES = ["flat", "rows", "transpose", "generic", "two_sided"]
DT = {1: ElemType.U8, 2: ElemType.BF16, 4: ElemType.F32, 8: ElemType.U64}
GEN_ONLY = "--generic" in sys.argv
TR_ONLY = "--transpose" in sys.argv


def run(label, es, shape, strides, out_strides=None, src_elems=None):
    n = 1
    for d in shape:
        n *= d
    src = client.empty((src_elems or n) * es)
    dst_elems = n if out_strides is None else sum((d - 1) * s for d, s in zip(shape, out_strides)) + 1
    dst = client.empty(dst_elems * es)
    tin = TensorHandle.new(src, shape, strides, DT[es])
    tout = TensorHandle.new(dst, shape, out_strides, DT[es]) if out_strides else TensorHandle.new_contiguous(shape, dst, DT[es])
    path, acc = ops.copy_plan(client, tin, tout)
    med, best = bench.samples_op(client, ev, lambda: ops.copy_into(client, tin, tout))
    b = 2 * n * es
    print(f"{label:34s} es {es}  {NAMES[path]:9s} x{acc:<2d}  {b / 2**20:7.0f} MiB moved  median {med * 1e3:8.1f} us  {b / med / 1e6:6.0f} GB/s  best {b / best / 1e6:6.0f}", flush=True)


for es in (4, 2, 1):
    e = (1 << 29) // es          # 512 MiB tensors
    if GEN_ONLY:
        run("NHWC->NCHW, 3 channels", es, [e // (3 * 224 * 224), 3, 224, 224], [3 * 224 * 224, 1, 224 * 3, 3])
        run("stride-2 gather", es, [e // 2], [2], src_elems=e)
        run("stride-7 gather", es, [e // 8], [7], src_elems=e)
        run("stride-3 scatter", es, [e // 4], [1], out_strides=[3])
        run("short rows [*, 24] of 32", es, [e // 32, 24], [32, 1], src_elems=e)
        run("small transposes [*,8,8]", es, [e // 64, 8, 8], [64, 1, 8])
        continue
    if TR_ONLY:
        side = {4: (8192, 16384), 2: (16384, 16384), 1: (16384, 32768)}[es]
        run(f"2-D transpose {side}", es, [side[1], side[0]], [1, side[1]])
        run(f"2-D transpose [{side[0] + 256},{side[1] - 512}]", es, [side[1] - 512, side[0] + 256], [1, side[1] - 512])
        b = e // (2048 * 2048)
        run(f"batched transpose {b} x 2048^2", es, [b, 2048, 2048], [2048 * 2048, 1, 2048])
        run(f"K^T [*,4096,64]", es, [e // (4096 * 64), 64, 4096], [4096 * 64, 1, 64])
        run("full axis reversal [64,64,64,*]", es, [e // 64**3, 64, 64, 64], [1, e // 64**3, e // 64**2, e // 64])
        continue
    run("flat 512 MiB", es, [e], [1])
    r = e // 4096
    run("rows, pitched input (4096+64)", es, [r, 4096], [4160, 1], src_elems=r * 4160)
    run("rows, batch swap [h,b,w]", es, [r // 16, 16, 4096], [4096, (r // 16) * 4096, 1])
    side = {4: (8192, 16384), 2: (16384, 16384), 1: (16384, 32768)}[es]
    run(f"2-D transpose {side}", es, [side[1], side[0]], [1, side[1]])
    b = e // (2048 * 2048)
    run(f"batched transpose {b} x 2048^2", es, [b, 2048, 2048], [2048 * 2048, 1, 2048])
    nb = e // (256 * 56 * 56)
    run(f"NCHW->NHWC [{nb},256,56,56]", es, [nb, 56, 56, 256], [256 * 3136, 56, 1, 3136])
    run(f"NHWC->NCHW [{nb},256,56,56]", es, [nb, 256, 56, 56], [256 * 3136, 1, 56 * 256, 256])
    run("NHWC->NCHW, 3 channels", es, [e // (3 * 224 * 224), 3, 224, 224], [3 * 224 * 224, 1, 224 * 3, 3])
    for dh in (128, 64):
        bh = e // (4096 * dh)
        run(f"K^T [{bh},4096,{dh}] -> [{bh},{dh},4096]", es, [bh, dh, 4096], [4096 * dh, 1, dh])
    run("stride-2 gather", es, [e // 2], [2], src_elems=e)
    run("stride-3 scatter", es, [e // 4], [1], out_strides=[3])
    run("full axis reversal [64,64,64,*]", es, [e // 64**3, 64, 64, 64], [1, e // 64**3, e // 64**2, e // 64])
run("8-byte transpose 8192 x 4096", 8, [4096, 8192], [1, 4096])
run("8-byte transpose 8448 x 3968", 8, [3968, 8448], [1, 3968])
