"""copy_into / into_contiguous / into_contiguous_packed through the C ABI on a real MI355X, against oracle/layout.py.
Mirrors crates/cubecl-std/src/tests/tensor/into_contiguous.rs (the seven reference tests come first, same shapes and
payloads) and then covers every mover of cubecl_amd/csrc/copy_strided.hip at sizes that fill the chip.  Bit-exact."""
import itertools

import numpy as np
import pytest

from cubecl_amd import ElemType, ServerError, TensorHandle, ops
from cubecl_amd import _native as N
from oracle import layout as L

pytestmark = pytest.mark.gpu

NP = {1: (np.uint8, ElemType.U8), 2: (np.uint16, ElemType.BF16), 4: (np.uint32, ElemType.U32), 8: (np.uint64, ElemType.U64)}


def payload(n, es, seed=0):
    """Distinct-ish values in every byte position so a misplaced or truncated element shows."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=n * es, dtype=np.uint8).view(NP[es][0])


def flat(client, handle, n, es):
    return TensorHandle.new_contiguous((n,), handle, NP[es][1]).to_numpy(client)


def run_copy(client, base, shape, strides, es, out_shape=None, out_strides=None, out_elems=None, offset=0, expect_path=None):
    """Upload `base` (flat), view it as (shape, strides) `offset` elements in, copy into an output view and compare
    the whole output buffer (untouched bytes included) with the oracle."""
    out_shape = list(shape if out_shape is None else out_shape)
    out_strides = L.contiguous_strides(out_shape) if out_strides is None else list(out_strides)
    n = int(np.prod(shape))
    if out_elems is None:
        out_elems = (sum((d - 1) * s for d, s in zip(out_shape, out_strides)) + 1) if n else 0
    dt = NP[es][1]
    src = client.create_from_slice(base)
    if offset:
        src = src.offset_start_by(offset * es)
    sentinel = payload(out_elems, es, seed=99)
    dst = client.create_from_slice(sentinel)
    tin = TensorHandle.new(src, shape, strides, dt)
    tout = TensorHandle.new(dst, out_shape, out_strides, dt)
    if expect_path is not None:
        assert ops.copy_plan(client, tin, tout) == expect_path
    ops.copy_into(client, tin, tout)
    got = flat(client, dst, out_elems, es)
    want = L.copy_into(base[offset:], shape, strides, sentinel.copy(), out_shape, out_strides)
    assert np.array_equal(got, want), f"shape {shape} strides {strides} -> {out_shape} {out_strides} (es {es})"


# ---- the reference's own tests -----------------------------------------------------------------------------------------
def test_rank_mismatch(client):
    # into_contiguous.rs:139-185: NHWC storage viewed as NCHW [1, 2, 4, 1], copied into a rank-3 contiguous output
    data = np.arange(1, 9, dtype=np.float32)
    tin = TensorHandle.new(client.create_from_slice(data), (1, 2, 4, 1), (8, 1, 2, 2), ElemType.F32)
    tout = TensorHandle.new_contiguous((1, 2, 4), client.empty(32), ElemType.F32)
    ops.copy_into(client, tin, tout)
    assert np.array_equal(tout.to_numpy(client).reshape(-1), np.array([1, 3, 5, 7, 2, 4, 6, 8], dtype=np.float32))


def permuted_case(client, base_shape, perm, es):
    # run_permuted_case (:189-246): payload (i % 251) + 1, a permuted view of a contiguous buffer -> contiguous
    n = int(np.prod(base_shape))
    data = ((np.arange(n) % 251) + 1).astype(NP[es][0])
    bstr = L.contiguous_strides(base_shape)
    shape = [base_shape[p] for p in perm]
    strides = [bstr[p] for p in perm]
    t = TensorHandle.new(client.create_from_slice(data), shape, strides, NP[es][1])
    got = ops.into_contiguous(client, t).to_numpy(client)
    assert np.array_equal(got, data.reshape(base_shape).transpose(perm)), (base_shape, perm, es)


def test_permuted_unaligned_axis(client):
    permuted_case(client, [2, 2, 3], [2, 1, 0], 1)       # :251-253, the bool swap_dims(0, 2) repro


def test_permuted_sweep(client):
    # :257-283: every permutation of these shapes, for a 1-byte and a 4-byte element
    shapes = [[2, 3], [3, 2], [4, 6], [6, 4], [2, 2, 3], [3, 2, 2], [2, 3, 4], [4, 3, 2], [8, 3, 4], [3, 8, 5], [16, 5], [5, 16],
              [8, 8], [16, 16], [32, 2, 3], [2, 3, 4, 5]]
    for shape in shapes:
        for perm in itertools.permutations(range(len(shape))):
            permuted_case(client, shape, list(perm), 1)
            permuted_case(client, shape, list(perm), 4)


def repack_case(client, shape, in_pack_dim, packing, bits, word=4):
    # run_repack_case (:63-118): payload (q % 15) + 1 packed along in_pack_dim, re-packed onto the innermost axis
    rank = len(shape)
    n = int(np.prod(shape))
    npw, dt = (np.uint32, ElemType.U32) if word == 4 else (np.uint8, ElemType.U8)
    unpacked = ((np.arange(n) % ((1 << bits) - 1)) + 1).astype(np.uint32)
    storage = L.pack_along(unpacked, shape, in_pack_dim, packing, bits).astype(npw)
    in_shape = list(shape)
    in_shape[in_pack_dim] = -(-in_shape[in_pack_dim] // packing)
    expected = L.pack_along(unpacked, shape, rank - 1, packing, bits).astype(npw)
    tin = TensorHandle.new_contiguous(in_shape, client.create_from_slice(storage), dt)
    out = ops.into_contiguous_packed(client, tin, rank - 1 - in_pack_dim, shape, packing)
    got = out.to_numpy(client).reshape(-1)
    assert got.any()
    assert np.array_equal(got, expected), (shape, in_pack_dim)
    assert np.array_equal(got, L.into_contiguous_packed(storage, L.contiguous_strides(in_shape), shape, rank - 1 - in_pack_dim, packing,
                                                       word * 8))


@pytest.mark.parametrize("shape,dim", [([1, 8, 16], 1),      # test_into_contiguous_packed_repack (:122-124)
                                       ([1, 8, 8], 1),       # _vector_size_one (:127-129)
                                       ([4096, 256], 0),     # _multi_vector (:134-136)
                                       ([8192, 32], 0)])     # _halving (:140-142)
def test_packed_repack(client, shape, dim):
    repack_case(client, shape, dim, 8, 4)


def test_packed_repack_more(client):
    repack_case(client, [24, 6, 40], 0, 8, 4)
    repack_case(client, [6, 16, 40], 1, 4, 8)
    repack_case(client, [33, 64, 16], 0, 16, 2)
    repack_case(client, [5, 7, 32], 2, 8, 4)                  # already innermost: a plain copy
    repack_case(client, [16, 12, 8], 0, 2, 4, word=1)         # u8 words holding two 4-bit values
    repack_case(client, [9, 8, 24], 1, 8, 1, word=1)


# ---- every mover, at sizes that fill the chip ----------------------------------------------------------------------------
@pytest.mark.parametrize("es", [1, 2, 4, 8])
def test_flat_and_rows(client, es):
    n = 3 * 1000 * 1024
    base = payload(n + 64, es)
    run_copy(client, base, [n], [1], es, expect_path=(N.COPY_PATH_FLAT, 16))
    run_copy(client, base, [3, 1000, 1024], [1024000, 1024, 1], es, expect_path=(N.COPY_PATH_FLAT, 16))
    # pitched input rows (stride 1040 > 1024), contiguous output
    run_copy(client, payload(3000 * 1040, es), [3000, 1024], [1040, 1], es, expect_path=(N.COPY_PATH_ROWS, 16))
    # contiguous input, pitched output (into_contiguous_pitched's shape), untouched padding must stay
    run_copy(client, base, [3000, 1000], [1000, 1], es, out_strides=[1008, 1], expect_path=(N.COPY_PATH_ROWS, 16 if es > 1 else 8))
    # batch axes swapped, rows contiguous: [b, h, w] viewed as [h, b, w]
    run_copy(client, payload(12 * 300 * 256, es), [300, 12, 256], [256, 300 * 256, 1], es, expect_path=(N.COPY_PATH_ROWS, 16))
    # odd row length / odd offset: narrower accesses, same answer
    run_copy(client, payload(2000 * 77 + 8, es), [2000, 75], [77, 1], es, offset=1)
    # a sliced window of every axis
    run_copy(client, payload(40 * 50 * 60, es), [30, 20, 44], [3000, 60, 1], es, offset=5 * 3000 + 7 * 60 + 4)


@pytest.mark.parametrize("es", [1, 2, 4, 8])
def test_transpose_tiles(client, es):
    # whole tiles, 16-byte accesses on both sides
    run_copy(client, payload(1024 * 768, es), [768, 1024], [1, 768], es, expect_path=(N.COPY_PATH_TRANSPOSE, 16))
    # ragged edges in both directions, rows not 16-byte multiples
    run_copy(client, payload(517 * 301, es), [301, 517], [1, 301], es, expect_path=(N.COPY_PATH_TRANSPOSE, es))
    # whole rows 16-byte aligned but the extents ragged: interior tiles vectorised, edge tiles element-wise
    run_copy(client, payload(528 * 784, es), [784, 528], [1, 784], es, expect_path=(N.COPY_PATH_TRANSPOSE, 16))
    # batched transpose [b, m, n] -> [b, n, m], and with the batch axis in the middle of the source
    run_copy(client, payload(6 * 512 * 320, es), [6, 320, 512], [512 * 320, 1, 320], es, expect_path=(N.COPY_PATH_TRANSPOSE, 16))
    run_copy(client, payload(6 * 512 * 320, es), [320, 6, 512], [1, 320 * 512, 320], es, expect_path=(N.COPY_PATH_TRANSPOSE, 16))
    # NCHW -> NHWC and back (channels 48)
    run_copy(client, payload(4 * 48 * 40 * 56, es), [4, 40, 56, 48], [48 * 40 * 56, 56, 1, 40 * 56], es)
    run_copy(client, payload(4 * 48 * 40 * 56, es), [4, 48, 40, 56], [48 * 40 * 56, 1, 56 * 48, 48], es)
    # K^T of attention: [b, h, s, d] -> [b, h, d, s] with d = 64 (half a tile along P for 2-byte elements), and d = 24
    run_copy(client, payload(2 * 3 * 200 * 64, es), [2, 3, 64, 200], [3 * 200 * 64, 200 * 64, 1, 64], es, expect_path=(N.COPY_PATH_TRANSPOSE, 16 if (200 * es) % 16 == 0 and 200 % (16 // es) == 0 else es))
    run_copy(client, payload(2 * 3 * 208 * 24, es), [2, 3, 24, 208], [3 * 208 * 24, 208 * 24, 1, 24], es)
    # contiguous source scattered into a permuted destination (the transposition on the output side)
    run_copy(client, payload(640 * 384, es), [640, 384], [384, 1], es, out_strides=[1, 640], expect_path=(N.COPY_PATH_TRANSPOSE, 16))
    # transposed AND pitched on both sides, pointer offset by one row
    run_copy(client, payload(400 * 528 + 528, es), [512, 400], [1, 528], es, out_strides=[416, 1], offset=528)


@pytest.mark.parametrize("es", [1, 2, 4, 8])
def test_generic_and_two_sided(client, es):
    # short innermost axes: no tile to speak of
    run_copy(client, payload(1000 * 3 * 5, es), [5, 1000, 3], [3, 15, 1], es)
    run_copy(client, payload(7 * 9 * 11 * 13, es), [13, 11, 9, 7], [1, 13, 13 * 11, 13 * 11 * 9], es)
    # stride-2 gather and a broadcast axis (stride 0)
    pk = {1: 4, 2: 8, 4: 8, 8: 8}[es]                            # bytes per access of the contiguous side
    run_copy(client, payload(2 * 4096, es), [4096], [2], es, expect_path=(N.COPY_PATH_GENERIC, pk))
    run_copy(client, payload(4096, es), [4096], [1], es, out_strides=[3], expect_path=(N.COPY_PATH_GENERIC, pk))      # scatter
    run_copy(client, payload(3 * 4098, es), [4098], [3], es, expect_path=(N.COPY_PATH_GENERIC, min(2 * es, 8)))
    run_copy(client, payload(3000 * 64, es), [3000, 8, 8], [64, 1, 8], es, expect_path=(N.COPY_PATH_GENERIC, min(8 * es, 16)))   # 8 x 8 blocks
    run_copy(client, payload(3 * 4097, es), [4097], [3], es, out_strides=[2], expect_path=(N.COPY_PATH_GENERIC, es))
    run_copy(client, payload(512, es), [300, 512], [0, 1], es)
    run_copy(client, payload(300, es), [300, 512], [1, 0], es)
    # an axis too short to tile (8 < 16): generic, packed along the output's contiguous axis
    run_copy(client, payload(8 * 384, es), [384, 8], [1, 384], es, expect_path=(N.COPY_PATH_GENERIC, min(8 * es, 16)))
    run_copy(client, payload(8 * 384, es), [8, 384], [1, 8], es, expect_path=(N.COPY_PATH_GENERIC, {1: 4, 2: 8, 4: 8, 8: 8}[es]))
    # shapes with no common refinement: strided [2, 3] viewed into [3, 2], and a bigger one
    run_copy(client, payload(16, es), [2, 3], [1, 2], es, out_shape=[3, 2], expect_path=(N.COPY_PATH_GENERIC, es))
    run_copy(client, payload(16, es), [2, 3], [1, 2], es, out_shape=[3, 2], out_strides=[4, 1], expect_path=(N.COPY_PATH_TWO_SIDED, es))
    run_copy(client, payload(35 * 33 * 2, es), [35, 33], [1, 70], es, out_shape=[21, 55], out_strides=[56, 1])
    # rank mismatch with a common refinement: [6, 8, 10] (permuted) -> [48, 10] and -> [6, 80]
    run_copy(client, payload(480, es), [6, 8, 10], [10, 60, 1], es, out_shape=[48, 10])
    run_copy(client, payload(480, es), [6, 8, 10], [80, 10, 1], es, out_shape=[6, 80], out_strides=[96, 1])
    # 8 axes
    shape = [2, 3, 2, 3, 2, 3, 2, 5]
    perm = [7, 0, 6, 1, 5, 2, 4, 3]
    bstr = L.contiguous_strides(shape)
    run_copy(client, payload(int(np.prod(shape)), es), [shape[p] for p in perm], [bstr[p] for p in perm], es)


def test_empty_and_errors(client):
    t = TensorHandle.new_contiguous((0, 8), client.empty(0), ElemType.F32)
    assert ops.into_contiguous(client, t).num_elems() == 0
    a = TensorHandle.new_contiguous((4, 8), client.empty(128), ElemType.F32)
    b = TensorHandle.new_contiguous((4, 7), client.empty(128), ElemType.F32)
    with pytest.raises(ServerError) as e:
        ops.copy_into(client, a, b)                              # different element counts
    assert e.value.code == N.E_INVALID_ARGUMENT
    with pytest.raises(ServerError) as e:
        ops.copy_into(client, a, TensorHandle.new(b.handle, (4, 8), (0, 1), ElemType.F32))   # broadcasting output
    assert e.value.code == N.E_INVALID_ARGUMENT
    with pytest.raises(ServerError) as e:
        ops.copy_into(client, a, TensorHandle.new_contiguous((4, 8), client.empty(64), ElemType.BF16))
    assert e.value.code == N.E_INVALID_ARGUMENT


def test_pitched_output_and_round_trip(client):
    # into_contiguous_pitched (launch.rs:22-37): rows of 30 f32 land on the pitch create_tensor would pick
    data = np.arange(7 * 30 * 5, dtype=np.float32)
    t = TensorHandle.new(client.create_from_slice(data), (7, 30, 5), (150, 5, 1), ElemType.F32).permute([0, 2, 1])
    out = ops.into_contiguous_pitched(client, t)
    assert out.is_contiguous_pitched() and out.strides[-2] * 4 % 128 == 0
    assert np.array_equal(out.to_numpy(client), data.reshape(7, 30, 5).transpose(0, 2, 1))
    # 256 MiB of bf16: transposing twice is the identity, and a transposed copy holds the same multiset (checksum)
    n = 8192 * 16384
    x = TensorHandle.uniform(client, (8192, 16384), ElemType.BF16, seed=7, tensor_id=3, lo=-1.0, hi=1.0)
    y = ops.into_contiguous(client, x.permute([1, 0]))
    assert ops.copy_plan(client, x.permute([1, 0]), y) == (N.COPY_PATH_TRANSPOSE, 16)
    z = ops.into_contiguous(client, y.permute([1, 0]))
    xs, zs = flat(client, x.handle, n, 2), flat(client, z.handle, n, 2)
    assert np.array_equal(xs, zs)
    ys = flat(client, y.handle, n, 2)
    assert int(ys.astype(np.uint64).sum()) == int(xs.astype(np.uint64).sum())
    assert np.array_equal(ys.reshape(16384, 8192)[::97, ::89], xs.reshape(8192, 16384).T[::97, ::89])


def test_matmul_takes_highly_permuted_operands(client, oracle):
    # an lhs stored [k, b, m] (batch in the middle, no unit stride in the matrix axes of the [b, m, k] view): the launcher
    # makes it contiguous first, as the reference's matmul launchers do with into_contiguous
    b, m, n, k = 3, 96, 80, 64
    a = oracle.fill_uniform(b * m * k, 1, -1, 1).reshape(b, m, k)
    w = oracle.fill_uniform(b * k * n, 2, -1, 1).reshape(b, k, n)
    stored = np.ascontiguousarray(a.transpose(2, 0, 1))                                             # [k, b, m]
    ta = TensorHandle.from_numpy(client, stored).permute([1, 2, 0])                                  # view [b, m, k], strides (m, 1, b*m)
    assert ta.strides == (m, 1, b * m)
    tb = TensorHandle.from_numpy(client, w)
    tc = TensorHandle.zeros(client, (b, m, n), ElemType.F32)
    ops.matmul(client, ta, tb, tc)
    want = np.einsum("bmk,bkn->bmn", a.astype(np.float64), w.astype(np.float64))
    assert np.allclose(tc.to_numpy(client), want, rtol=1e-5, atol=1e-5)


# ---- tensor::identity (crates/cubecl-std/src/tests/tensor/identity.rs) ----------------------------------------------------
ONE = {ElemType.F32: (np.float32, 1.0), ElemType.F64: (np.float64, 1.0), ElemType.F16: (np.float16, 1.0), ElemType.I32: (np.int32, 1),
       ElemType.U32: (np.uint32, 1), ElemType.I64: (np.int64, 1), ElemType.U8: (np.uint8, 1), ElemType.BF16: (np.uint16, 0x3F80),
       ElemType.F8E4M3: (np.uint8, 0x38), ElemType.F8E5M2: (np.uint8, 0x3C)}


@pytest.mark.parametrize("dim", [4, 16, 256, 1024])            # test_tiny / _small / _normal / _large (test_macros/identity.rs:17-35)
def test_identity(client, dim):
    for dt, (npt, one) in ONE.items():
        t = TensorHandle.empty(client, (dim, dim), dt)          # pitched when the rows ask for it, as in the reference test
        ops.identity(client, t)
        want = np.zeros((dim, dim), dtype=npt)
        want[np.arange(dim), np.arange(dim)] = one              # identity_cpu (test_utils.rs:3-15): every (dim + 1)-th element
        assert np.array_equal(t.to_numpy(client), want), (dt, dim)


def test_identity_pitched_rows_and_matmul(client, oracle):
    # odd dim -> pitched rows; the padding between rows is left alone, and I x B returns B's bits through the GEMM
    dim = 100
    t = TensorHandle.empty(client, (dim, dim), ElemType.F32)
    assert t.strides[0] > dim
    raw = TensorHandle.new_contiguous((dim * t.strides[0],), t.handle, ElemType.F32)
    client.write(t.handle, np.full(t.handle.size // 4, 7.0, dtype=np.float32))
    ops.identity(client, t)
    got = raw.to_numpy(client)[: (dim - 1) * t.strides[0] + dim].copy()
    img = np.full(dim * t.strides[0], 7.0, dtype=np.float32)
    img2 = img.reshape(dim, t.strides[0])
    img2[:, :dim] = np.eye(dim, dtype=np.float32)
    assert np.array_equal(got, img[: got.size])
    b = oracle.fill_uniform(dim * 40, 5, -1, 1).reshape(dim, 40)
    tb = TensorHandle.from_numpy(client, b)
    tc = TensorHandle.zeros(client, (dim, 40), ElemType.F32)
    ops.matmul(client, t, tb, tc)
    assert np.array_equal(tc.to_numpy(client), b)
    with pytest.raises(ServerError):
        ops.identity(client, TensorHandle.new_contiguous((4, 5), client.empty(80), ElemType.F32))


# ---- the short-axis mover (plane_copy_kernel): 2..4 interleaved elements <-> planes --------------------------------------
@pytest.mark.parametrize("es", [1, 2, 4])
@pytest.mark.parametrize("planes", [2, 3, 4])
def test_few_channels_interleaved_to_planar_and_back(client, es, planes):
    # NHWC <-> NCHW with 2..4 channels, both as into_contiguous of a permuted view (logical order = the output's) and as
    # copy_into of a contiguous input into a permuted output view (logical order = the input's): the four ways the two
    # innermost joint axes can be arranged.  H x W = 24 x 40 = 960 positions (a multiple of 16 / es), 5 images.
    n, h, w, c = 5, 24, 40, planes
    base = payload(n * h * w * c, es, seed=planes * 10 + es)
    nhwc, nchw = [h * w * c, w * c, c, 1], [c * h * w, h * w, w, 1]
    want = (N.COPY_PATH_GENERIC, 16)
    # NHWC buffer viewed as [N, C, H, W] -> contiguous NCHW (gather, q innermost)
    run_copy(client, base, [n, c, h, w], [nhwc[0], nhwc[3], nhwc[1], nhwc[2]], es, expect_path=want)
    # NCHW buffer viewed as [N, H, W, C] -> contiguous NHWC (scatter, p innermost)
    run_copy(client, base, [n, h, w, c], [nchw[0], nchw[2], nchw[3], nchw[1]], es, expect_path=want)
    # contiguous NHWC input [N, H, W, C] -> NCHW buffer through a permuted OUTPUT view (gather, p innermost)
    run_copy(client, base, [n, h, w, c], nhwc, es, out_strides=[nchw[0], nchw[2], nchw[3], nchw[1]], expect_path=want)
    # contiguous NCHW input [N, C, H, W] -> NHWC buffer through a permuted output view (scatter, q innermost)
    run_copy(client, base, [n, c, h, w], nchw, es, out_strides=[nhwc[0], nhwc[3], nhwc[1], nhwc[2]], expect_path=want)


def test_few_channels_with_padded_planes_and_batches(client):
    # planes and images further apart than they need to be (pitched destination), two batch axes
    es, c, hw = 2, 3, 256
    base = payload(2 * 3 * hw * c, es, seed=5)
    run_copy(client, base, [2, 3, c, hw], [3 * hw * c, hw * c, 1, c], es, out_strides=[3 * c * 320 + 64, c * 320, 320, 1],
             expect_path=(N.COPY_PATH_GENERIC, 16))


def test_few_channels_fall_back_when_the_vectors_do_not_fit(client):
    # 250 positions (not a multiple of 16): the element-wise generic mover; still exact
    for es in (1, 4):
        base = payload(4 * 250 * 3, es, seed=7)
        run_copy(client, base, [4, 3, 250], [750, 1, 3], es)
        run_copy(client, base, [4, 250, 3], [750, 1, 250], es)
    # a plane stride that is not a multiple of 16 bytes
    base = payload(2 * 3 * 64 + 64, 1, seed=8)
    run_copy(client, base, [2, 64, 3], [200, 1, 66], 1)


def test_few_channels_at_size(client):
    # 64 MiB of u8 RGB pixels, NHWC -> NCHW and back: the bench's case at an eighth of its size, bit for bit
    n, h, w, c = 446, 224, 224, 3
    base = payload(n * h * w * c, 1, seed=11)
    src = TensorHandle.new(client.create_from_slice(base), [n, c, h, w], [h * w * c, 1, w * c, c], ElemType.U8)
    planar = ops.into_contiguous(client, src)
    assert np.array_equal(planar.to_numpy(client).reshape(n, c, h, w), base.reshape(n, h, w, c).transpose(0, 3, 1, 2))
    back = ops.into_contiguous(client, TensorHandle.new(planar.handle, [n, h, w, c], [c * h * w, w, 1, h * w], ElemType.U8))
    assert np.array_equal(back.to_numpy(client).reshape(-1), base)
