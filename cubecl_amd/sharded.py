"""Multi-GPU form of the hot path: one process per GPU, units sharded, one small exchange step.

The reference drives several devices from one process with one server thread per device and
exchanges data through `ServerCommunication` (crates/cubecl-runtime/src/server/base.rs:632-737;
the CUDA/NCCL implementation crates/cubecl-cuda/src/compute/server.rs:666-797).  Here every rank
is its own process (one `mi355_ctx`, RCCL over xGMI underneath `ComputeClient.all_reduce /
all_gather`); this module holds the part that is the same on every transport:

* how units are partitioned (contiguous 1/N slices of an array, contiguous runs of a batch);
* how partial results are combined: sums through an all-reduce (`ReduceOperation.Sum`, the only
  reduction besides Mean the reference enum has, server/base.rs:623-628), argmax through an
  all-gather of (value, global index) pairs and the SAME combine rule on every rank -- larger
  value wins, NaN ranks above every number, -0 == +0, ties keep the LOWEST global index (the rule
  `mi355_argmax_f32` implements inside one GPU; an API delta over the reference).

The transport is a small interface so that the same code runs over RCCL on GPUs (`RcclExchange`)
and over `torch.distributed` with the gloo backend in the CPU test-suite (`TorchExchange`).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


# ---------------------------------------------------------------------------- partitioning ----

def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [start, start+count) of n units owned by `rank`: the first n % world ranks
    get one extra unit, so slices differ by at most one and cover [0, n) exactly once."""
    if world <= 0 or not (0 <= rank < world) or n < 0:
        raise ValueError(f"shard_range: bad arguments n={n} rank={rank} world={world}")
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def shard_aligned_range(n: int, rank: int, world: int, align: int) -> Tuple[int, int]:
    """Like shard_range but slice boundaries fall on multiples of `align` elements (so every GPU's
    slice of a 16-byte aligned array stays 16-byte aligned: align = 4 for f32); the ragged tail
    goes to the last rank."""
    blocks = -(-n // align)
    b0, bc = shard_range(blocks, rank, world)
    start = min(b0 * align, n)
    end = min((b0 + bc) * align, n)
    return start, end - start


# ---------------------------------------------------------------------------- argmax rule -----

def argmax_key(value: float) -> int:
    """Order-preserving integer key of an f32 (oracle/oracle.c argmax_key, reduce.hip argmax_key):
    IEEE order, -0 == +0, every NaN above +inf."""
    (u,) = struct.unpack("<I", struct.pack("<f", np.float32(value)))
    if (u & 0x7FFFFFFF) > 0x7F800000:
        return 0xFFFFFFFF
    if u == 0x80000000:
        u = 0
    return (~u & 0xFFFFFFFF) if (u & 0x80000000) else (u | 0x80000000)


def combine_argmax(pairs: Sequence[Tuple[float, int]]) -> Tuple[float, int]:
    """(value, global index) of the winner among per-shard winners.  Shards that held no element
    pass index < 0 and are ignored; an all-empty input gives (-inf, 0) like mi355_argmax_f32."""
    best: Optional[Tuple[int, int, float]] = None
    for value, index in pairs:
        if index < 0:
            continue
        k = argmax_key(value)
        if best is None or k > best[0] or (k == best[0] and index < best[1]):
            best = (k, int(index), float(value))
    if best is None:
        return float("-inf"), 0
    return best[2], best[1]


# ---------------------------------------------------------------------------- transports ------

class Exchange:
    """What the combine step needs from a transport."""
    rank: int
    world: int

    def all_reduce_sum_f32(self, value: float) -> float:
        raise NotImplementedError

    def all_gather_pairs(self, value: float, index: int) -> List[Tuple[float, int]]:
        raise NotImplementedError

    def barrier(self) -> None:
        raise NotImplementedError


class LocalExchange(Exchange):
    """world == 1: nothing to exchange."""

    def __init__(self):
        self.rank, self.world = 0, 1

    def all_reduce_sum_f32(self, value: float) -> float:
        return float(np.float32(value))

    def all_gather_pairs(self, value: float, index: int):
        return [(float(value), int(index))]

    def barrier(self) -> None:
        pass


class TorchExchange(Exchange):
    """torch.distributed transport (gloo on CPU in the tests; any initialised backend works)."""

    def __init__(self, device: str = "cpu"):
        import torch.distributed as dist
        self._dist = dist
        self._device = device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def all_reduce_sum_f32(self, value: float) -> float:
        import torch
        t = torch.tensor([value], dtype=torch.float32, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t[0])

    def all_gather_pairs(self, value: float, index: int):
        import torch
        # value travels as raw f32 bits so NaN payloads and -0 survive the trip
        bits = struct.unpack("<I", struct.pack("<f", np.float32(value)))[0]
        mine = torch.tensor([bits, index], dtype=torch.int64, device=self._device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self._dist.all_gather(out, mine)
        res = []
        for t in out:
            b, i = int(t[0]), int(t[1])
            res.append((struct.unpack("<f", struct.pack("<I", b & 0xFFFFFFFF))[0], i))
        return res

    def barrier(self) -> None:
        self._dist.barrier()


def _checked_ids(device_ids, rank: int):
    """The device list of a communicator as the library will see it: comm_init defines a rank as the position in the SORTED
    list (crates/cubecl-cuda/src/compute/server.rs:669-703), so an unsorted list with an explicit rank would silently talk
    to the wrong peers."""
    ids = list(device_ids)
    if ids != sorted(ids):
        raise ValueError(f"device_ids must be sorted (rank = position in the sorted list): {ids}")
    if not 0 <= rank < len(ids):
        raise ValueError(f"rank {rank} outside the {len(ids)} devices of the communicator")
    return ids


class RcclExchange(Exchange):
    """RCCL over xGMI through the C ABI (`mi355_all_reduce` / `mi355_all_gather`), i.e. the
    ServerCommunication operations of the reference on device buffers.  `client.comm_init` must
    have been called for `device_ids`."""

    def __init__(self, client, device_ids, rank: int):
        from .runtime import ElemType, ReduceOperation
        self._c, self._ids = client, _checked_ids(device_ids, rank)
        self._ElemType, self._Sum = ElemType, ReduceOperation.Sum
        self.rank, self.world = rank, len(self._ids)
        self._buf = client.empty(64 + 16 * self.world)

    def all_reduce_sum_device(self, handle) -> None:
        """In-place all-reduce of one f32 living on the device (the timed form: no host round trip)."""
        self._c.all_reduce(handle, handle, self._ElemType.F32, self._ids, self._Sum)
        self._c.sync_collective()

    def all_reduce_sum_f32(self, value: float) -> float:
        h = self._buf.offset_end_by(self._buf.size - 4)
        self._c.write(h, np.array([value], dtype=np.float32))
        self.all_reduce_sum_device(h)
        return float(np.frombuffer(self._c.read_one(h), dtype=np.float32)[0])

    def all_gather_pairs(self, value: float, index: int):
        src = self._buf.offset_start_by(16).offset_end_by(self._buf.size - 32)
        dst = self._buf.offset_start_by(64)
        rec = np.zeros(2, dtype=np.uint64)
        rec[0] = struct.unpack("<I", struct.pack("<f", np.float32(value)))[0]
        rec[1] = np.uint64(index & 0xFFFFFFFFFFFFFFFF)
        self._c.write(src, rec)
        self._c.all_gather(src, dst, self._ElemType.U64, self._ids)
        self._c.sync_collective()
        raw = np.frombuffer(self._c.read_one(dst), dtype=np.uint64)[: 2 * self.world].reshape(self.world, 2)
        out = []
        for b, i in raw:
            idx = int(i)
            if idx >= 1 << 63:
                idx -= 1 << 64
            out.append((struct.unpack("<f", struct.pack("<I", int(b) & 0xFFFFFFFF))[0], idx))
        return out

    def barrier(self) -> None:
        self._c.sync()

    def exchange_on_device(self, rec, index_base, out_sum, out_value, out_index, mode: str = "gather") -> None:
        """The whole exchange step without a host round trip (what the timed multi-GPU job runs).  `rec` is the 16-byte
        record the fused local pass wrote: {f32 max, f32 partial sum, u64 local index} (`mi355_sum_argmax_f32` with
        out_val = rec, out_sum = rec + 4, out_idx = rec + 8).

        mode "gather" (default since round 5): ONE collective -- `rec` is all-gathered, and behind the comm -> compute fence a
        64-lane kernel adds the partial sums in rank order and folds the argmax candidates with the single-GPU rule
        (`mi355_sum_argmax_combine_f32`): the same bits on every rank, and one RCCL launch latency instead of two.
        mode "all_reduce": the reference's shape (`ServerCommunication::all_reduce`, crates/cubecl-cuda/src/compute/
        server.rs:705-780) -- the partial sum is all-reduced (Sum) into `out_sum`, the record all-gathered for the argmax.

        Afterwards the global sum, maximum and its global index are in device memory on EVERY rank, ordered on the compute
        stream like any other kernel output."""
        from . import ops
        gathered = self._buf.offset_start_by(64)
        if mode == "gather":       # one library call (round 6: three calls through the binding were 6-10 us of a 12 us exchange)
            self._c.sum_argmax_exchange(rec, gathered, index_base, out_sum, out_value, out_index, self._ids)
        elif mode == "gather3":    # the same exchange as the three calls it is made of (kept for the A/B and the tests)
            self._c.all_gather(rec, gathered, self._ElemType.U64, self._ids)
            self._c.sync_collective()
            ops.sum_argmax_combine(self._c, gathered, self.world, index_base, out_sum, out_value, out_index)
        elif mode == "all_reduce":
            part = rec.offset_start_by(4).offset_end_by(rec.size_in_used() - 8)      # the record's second word
            self._c.all_reduce(part, out_sum, self._ElemType.F32, self._ids, self._Sum)
            self._c.all_gather(rec, gathered, self._ElemType.U64, self._ids)
            self._c.sync_collective()
            ops.argmax_combine(self._c, gathered, self.world, index_base, out_value, out_index)
        else:
            raise ValueError(f"exchange_on_device: mode {mode!r} (gather, gather3 or all_reduce)")


class RcclJob:
    """The job-level collectives of a one-process-per-GPU launch over the library's OWN communicator -- barrier and MAX /
    SUM of a few host numbers -- so that a multi-GPU run needs no second collective library in the process (bench.py until
    round 3 initialised torch's NCCL process group next to `mi355_comm_init`: two communicators, possibly two librccl copies).

    Rendezvous: rank 0 draws the unique id (`mi355_comm_unique_id`) and publishes it in the launcher's key-value store
    (`store.set / store.get`: torch.distributed.TCPStore, or anything with those two methods); every rank then joins
    `client.comm_init(ids, uid, rank)` -- the reference's comm_init (crates/cubecl-cuda/src/compute/server.rs:669-703: rank =
    position of the own device in the sorted id list).  `device_ids[rank]` must be this client's device."""

    KEY = "mi355cube/unique_id"

    def __init__(self, client, device_ids, rank: int, store, key: str = KEY):
        from .runtime import ElemType, ReduceOperation
        self._c, self._ids = client, _checked_ids(device_ids, rank)
        self._E, self._R = ElemType, ReduceOperation
        self.rank, self.world = rank, len(self._ids)
        if rank == 0:
            store.set(key, bytes(client.comm_unique_id()))
        uid = bytes(store.get(key))
        client.comm_init(self._ids, uid, rank=rank)
        self._buf = client.empty(256)
        client.write(self._buf, np.zeros(256, dtype=np.uint8))      # the barrier all-reduces its first word: never garbage / NaN

    def barrier(self) -> None:
        """Every rank's compute stream has drained and every rank has arrived: an all-reduce of one f32 behind client.sync()."""
        self._c.sync()
        h = self._buf.offset_end_by(self._buf.size - 4)
        self._c.all_reduce(h, h, self._E.F32, self._ids, self._R.Sum)
        self._c.sync_collective()
        self._c.sync()

    def _reduce_f64(self, values: Sequence[float], op) -> List[float]:
        n = len(values)
        if not 0 < n <= 16:
            raise ValueError("RcclJob: 1..16 values per call")
        h = self._buf.offset_start_by(64).offset_end_by(self._buf.size - 64 - 8 * n)
        self._c.write(h, np.asarray(values, dtype=np.float64))
        self._c.all_reduce(h, h, self._E.F64, self._ids, op)
        self._c.sync_collective()
        return [float(v) for v in np.frombuffer(self._c.read_one(h), dtype=np.float64)[:n]]

    def max_over_ranks(self, values: Sequence[float]) -> List[float]:
        return self._reduce_f64(values, self._R.Max)

    def sum_over_ranks(self, values: Sequence[float]) -> List[float]:
        return self._reduce_f64(values, self._R.Sum)


# ---------------------------------------------------------------------------- sharded ops -----

@dataclass
class ShardedReduceResult:
    total: float
    max_value: float
    max_index: int


def sharded_sum_argmax(n_total: int, ex: Exchange,
                       local_pass: Callable[[int, int], Tuple[float, float, int]]) -> ShardedReduceResult:
    """Array-wide sum + argmax of an n_total-element f32 array partitioned over ex.world ranks
    (config C4).  `local_pass(start, count)` reduces this rank's slice and returns
    (partial_sum, max_value, LOCAL index of the maximum or -1 for an empty slice) -- on a GPU
    that is one `mi355_sum_argmax_f32` pass over the resident slice."""
    start, count = shard_aligned_range(n_total, ex.rank, ex.world, 4)
    psum, pmax, pidx = local_pass(start, count) if count > 0 else (0.0, float("-inf"), -1)
    total = ex.all_reduce_sum_f32(psum)
    pairs = ex.all_gather_pairs(pmax, (start + pidx) if pidx >= 0 else -1)
    value, index = combine_argmax(pairs)
    return ShardedReduceResult(total, value, index)


def sharded_batch(batch: int, ex: Exchange) -> Tuple[int, int]:
    """Batched GEMM (config C5): the batch dimension is cut into contiguous runs, one per rank;
    the matrices are independent, so there is no data-path collective."""
    return shard_range(batch, ex.rank, ex.world)
