"""Seeded random tensor layouts through `copy_into` (bit-exact against oracle/layout.py) and random shapes through the
reductions over any axis (f32 / bf16 / f16; sums to 1e-5 of sum |x|, argmax bit-exact) -- the walk BETWEEN the hand-picked cases
of test_gpu_contiguous.py and test_gpu_reduce.py.  Layout semantics: crates/cubecl-std/src/tensor/contiguous/ (copy_into over
arbitrary strides, same element order as the logical tensor); reductions: cubecl-book .../v4-gpu.rs:47-70, v7-gpu.rs:50-77.
Every draw is a pure function of its seed, so a failure names its case."""
import numpy as np
import pytest

from cubecl_amd import ElemType, TensorHandle, ops
from oracle import layout as L
from test_gpu_contiguous import payload, run_copy

pytestmark = pytest.mark.gpu
import os
OFFSET = int(os.environ.get("MI355_FUZZ_OFFSET", "0"))     # soak runs: the same tests over another stretch of seeds
REL = 1e-5
DIMS = [1, 2, 3, 4, 5, 7, 8, 13, 16, 17, 24, 31, 32, 33, 48, 64, 65, 100, 128, 200, 256, 500, 1024]


def draw_layout(seed):
    rng = np.random.default_rng(0xC0B1 + seed + OFFSET)
    while True:
        rank = int(rng.integers(1, 6))
        base = [int(rng.choice(DIMS)) for _ in range(rank)]
        if 1 <= int(np.prod(base)) <= 3_000_000:
            break
    bstr = L.contiguous_strides(base)
    shape, strides, off = list(base), list(bstr), 0
    for ax in range(rank):                                   # a window and / or a step on some axes
        if rng.integers(0, 4) == 0 and shape[ax] > 1:
            d = int(rng.integers(1, shape[ax] + 1))
            start = int(rng.integers(0, shape[ax] - d + 1))
            off += start * strides[ax]
            shape[ax] = d
        if rng.integers(0, 6) == 0 and shape[ax] > 2:
            shape[ax] = (shape[ax] + 1) // 2
            strides[ax] *= 2
    perm = list(rng.permutation(rank))
    shape, strides = [shape[p] for p in perm], [strides[p] for p in perm]
    if rng.integers(0, 7) == 0 and int(np.prod(shape)) <= 500_000:   # a broadcast axis
        at = int(rng.integers(0, rank + 1))
        shape.insert(at, int(rng.integers(2, 6)))
        strides.insert(at, 0)
    es = int(rng.choice([1, 2, 4, 8]))
    mode = int(rng.integers(0, 4))
    out_strides = None
    if mode == 1 and len(shape) >= 2:                        # pitched rows
        padded = list(shape)
        padded[-1] += int(rng.choice([1, 3, 8, 16]))
        out_strides = L.contiguous_strides(padded)
    elif mode == 2 and len(shape) >= 2:                      # the permutation on the output side
        q = list(rng.permutation(len(shape)))
        cs = L.contiguous_strides([shape[i] for i in q])
        out_strides = [0] * len(shape)
        for pos, ax in enumerate(q):
            out_strides[ax] = cs[pos]
    return int(np.prod(base)), shape, strides, es, out_strides, off


@pytest.mark.parametrize("seed", range(400))
def test_copy_into_on_random_layouts(client, seed):
    n, shape, strides, es, out_strides, off = draw_layout(seed)
    run_copy(client, payload(n, es, seed=seed), shape, strides, es, out_strides=out_strides, offset=off)


def draw_reduce(seed):
    rng = np.random.default_rng(0xA715 + seed + OFFSET)
    while True:
        rank = int(rng.integers(1, 5))
        shape = tuple(int(rng.choice(DIMS + [4096, 10007, 70001])) for _ in range(rank))
        if 1 <= int(np.prod(shape)) <= 6_000_000:
            break
    return shape, int(rng.integers(-rank, rank)), [ElemType.F32, ElemType.F32, ElemType.BF16, ElemType.F16][int(rng.integers(0, 4))]


@pytest.mark.parametrize("seed", range(240))
def test_axis_reductions_on_random_shapes(client, oracle, seed):
    shape, axis, dtype = draw_reduce(seed)
    n = int(np.prod(shape))
    x = oracle.fill_uniform(n, 3000 + seed, -1.0, 1.0).reshape(shape)
    # a planted tie per reduced line keeps the lowest-index rule in play
    if shape[axis] >= 4:
        idx = [slice(None)] * len(shape)
        idx[axis] = shape[axis] // 2
        hi = [slice(None)] * len(shape)
        hi[axis] = shape[axis] - 1
        x[tuple(idx)] = 2.0
        x[tuple(hi)] = 2.0
    if dtype == ElemType.F32:
        t, xv = TensorHandle.from_numpy(client, x), x
    else:
        bits = (oracle.to_bf16 if dtype == ElemType.BF16 else oracle.to_f16)(x.reshape(-1))
        xv = (oracle.from_bf16 if dtype == ElemType.BF16 else oracle.from_f16)(bits).reshape(shape)
        t = TensorHandle.from_numpy(client, bits.reshape(shape), dtype)
    out_shape = tuple(d for i, d in enumerate(shape) if i != axis % len(shape))
    m = int(np.prod(out_shape)) if out_shape else 1
    s = TensorHandle.new_contiguous(out_shape or (1,), client.empty(max(m, 1) * 4), ElemType.F32)
    ops.reduce_sum_axis(client, t, s, axis)
    ref = oracle.reduce_axis_sum(xv, axis)
    bound = np.abs(xv).astype(np.float64).sum(axis=axis)
    assert np.all(np.abs(s.to_numpy(client).reshape(ref.shape).astype(np.float64) - ref) <= REL * bound + 1e-30), (shape, axis, dtype)
    a = TensorHandle.new_contiguous(out_shape or (1,), client.empty(max(m, 1) * 4), ElemType.U32)
    ops.argmax_axis(client, t, a, axis)
    assert np.array_equal(a.to_numpy(client).reshape(ref.shape), oracle.reduce_axis_argmax(xv, axis)), (shape, axis, dtype)
