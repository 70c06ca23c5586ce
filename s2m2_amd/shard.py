"""Multi-GPU path: stereo pairs are independent units (every op carries the batch dim independently), so
pairs shard across ranks with NO data-path collective; the only exchange is the final gather of the three output maps
(3 x H x W fp32 per pair) to one rank.  One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).  With RCCL, ``gather`` is a grouped send/recv: every peer writes straight to the root
over its own xGMI link (no ring), which is the right shape for a fully connected 8-GPU node."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch


def shard_indices(n_pairs: int, rank: int, world: int) -> List[int]:
    """Pair p runs on rank p % world (round robin keeps ranks within one pair of each other)."""
    return list(range(rank, n_pairs, world))


class _BufferPool:
    """Staging and receive buffers of the output gather, allocated once per (shape, dtype, device, world) and recycled: a step of the
    benchmark gathers 3 x H x W fp32 per pair from every rank -- at c5 (2432x2048, 8 ranks) 480 MB of receive buffers that the first
    version allocated and stacked anew every step.  A set goes back to the pool when its gather has been waited for; more gathers in
    flight than pooled sets simply allocate another set."""

    def __init__(self):
        self.free = {}

    def take(self, key, make):
        sets = self.free.get(key)
        return sets.pop() if sets else make()

    def give(self, key, item):
        sets = self.free.setdefault(key, [])
        if len(sets) < 4:
            sets.append(item)


_POOL = _BufferPool()


class PendingGather:
    """An output gather in flight (``gather_outputs_async``): ``wait()`` completes it and returns, on ``dst``, the tuple of tensors with
    the rank dimension folded into batch, rank-major: (world*B,1,H,W); None elsewhere.  With RCCL the collective runs on its own
    stream, so the next forward can be enqueued before the maps of this one have arrived (``wait`` makes the CURRENT stream wait,
    not the host).  The returned tensors are views of ONE fresh allocation (the pooled receive buffers are recycled)."""

    def __init__(self, work, key, buf: torch.Tensor, bufs, n_out: int, is_dst: bool):
        self.work, self.key, self.buf, self.bufs, self.n_out, self.is_dst = work, key, buf, bufs, n_out, is_dst

    def wait(self, stack: bool = True) -> Optional[Tuple[torch.Tensor, ...]]:
        """stack=False: only complete the collective; nothing is returned."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        res = None
        if self.is_dst and stack and self.bufs is not None:
            world = len(self.bufs)
            allb = torch.empty((self.n_out, world) + tuple(self.buf.shape[1:]), dtype=self.buf.dtype, device=self.buf.device)
            for r, b in enumerate(self.bufs):                                 # one copy per rank into the result, no intermediate stack
                allb[:, r].copy_(b)
            res = tuple(allb[i].reshape(-1, *self.buf.shape[2:]) for i in range(self.n_out))
        if self.buf is not None:                                              # (copies above are ordered on the current stream before reuse)
            _POOL.give(self.key, (self.buf, self.bufs))
            self.buf = self.bufs = None
        return res


def gather_outputs_async(out: Sequence[torch.Tensor], dist, dst: int = 0, group=None) -> PendingGather:
    """Start the gather of same-shaped (disp, occ, conf) tuples from all ranks to ``dst``: one collective per call (the three maps
    travel as one stacked buffer, copied out of the forward's output tensors first, so a hipGraph replay may overwrite those)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shape = (len(out),) + tuple(out[0].shape)                                # (3,B,1,H,W)
    dev = out[0].device
    key = (shape, str(dev), world, rank == dst)

    def make():
        b = torch.empty(shape, dtype=torch.float32, device=dev)
        return b, ([torch.empty_like(b) for _ in range(world)] if rank == dst else None)

    buf, bufs = _POOL.take(key, make)
    for i, o in enumerate(out):
        buf[i].copy_(o)                                                       # (casts to fp32)
    work = dist.gather(buf, bufs, dst=dst, group=group, async_op=True)
    return PendingGather(work, key, buf, bufs, len(out), rank == dst)


def gather_outputs(out: Sequence[torch.Tensor], dist, dst: int = 0, group=None) -> Optional[Tuple[torch.Tensor, ...]]:
    """Gather same-shaped (disp, occ, conf) tuples from all ranks to ``dst`` and wait for them.

    Returns on dst a tuple of tensors with the rank dimension folded into batch, rank-major: (world*B,1,H,W);
    None elsewhere."""
    return gather_outputs_async(out, dist, dst, group).wait()


def padded_shard(n_pairs: int, rank: int, world: int) -> Tuple[List[int], int]:
    """This rank's pair indices padded to the common shard size ceil(n / world) -> (indices, number that are real).  Ranks whose share is
    one short (n % world != 0) repeat their last pair -- or pair 0 if they have none -- so that every rank runs the same launch sequence
    and the gather stays ONE fixed-size collective; the padding results are dropped by the receiver."""
    idx = shard_indices(n_pairs, rank, world)
    real = len(idx)
    per = -(-n_pairs // world)
    return idx + [idx[-1] if idx else 0] * (per - real), real


def run_sharded(forward: Callable[[torch.Tensor, torch.Tensor], Sequence[torch.Tensor]], left: torch.Tensor,
                right: torch.Tensor, dist, dst: int = 0, group=None):
    """Every rank holds (or can index) the full batch of pairs; it processes its shard and rank ``dst`` receives all
    outputs in the original pair order.  Any n_pairs >= 1: shards are padded to ceil(n / world) pairs (``padded_shard``: pad and drop),
    so the exchange is one fixed-size gather whatever n is."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = left.shape[0]
    if n < 1:
        raise ValueError("run_sharded: no pairs")
    idx, _ = padded_shard(n, rank, world)
    out = forward(left[idx], right[idx])
    g = gather_outputs(out, dist, dst, group)
    if g is None:
        return None
    per = -(-n // world)
    res = [torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in g]
    for r in range(world):                                  # rank-major blocks of `per` entries; the first `real` of each are pairs
        real_idx = shard_indices(n, r, world)
        if real_idx:
            sel = torch.tensor(real_idx, device=g[0].device)
            for dst_t, src_t in zip(res, g):
                dst_t[sel] = src_t[r * per:r * per + len(real_idx)]
    return tuple(res)
