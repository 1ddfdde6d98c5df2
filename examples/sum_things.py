"""The reference's `examples/sum_things` (examples/sum_things/src/lib.rs:6-19, :180-226) on the MI355X path:
the four-element input of the example (-> 15), then config C1 (2^20 f32) through the array-wide reduction.

    python examples/sum_things.py          # needs the GPU and cubecl_amd/csrc/libmi355cube.so
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cubecl_amd import ElemType, Mi355Runtime, TensorHandle, ops  # noqa: E402


def main() -> None:
    client = Mi355Runtime.client()
    out = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    x = TensorHandle.from_numpy(client, np.array([-1.0, 10.0, 1.0, 5.0], dtype=np.float32))
    ops.reduce_sum(client, x, out)
    print("sum_things [-1, 10, 1, 5] =", float(out.to_numpy(client)[0]))          # the example prints 15
    n = 1 << 20
    big = TensorHandle.from_numpy(client, (np.arange(n) % 17).astype(np.float32))
    ops.reduce_sum(client, big, out)
    print(f"sum of {n} values (i % 17) =", float(out.to_numpy(client)[0]), "expected", float(((np.arange(n) % 17).astype(np.float64)).sum()))
    idx = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.U64)
    val = TensorHandle.new_contiguous((1,), client.empty(8), ElemType.F32)
    ops.argmax(client, big, idx, val)
    print("argmax =", int(idx.to_numpy(client)[0]), "value", float(val.to_numpy(client)[0]), "(lowest index of the maximum)")


if __name__ == "__main__":
    main()
