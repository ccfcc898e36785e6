"""Multi-GPU sharding of the batched path: alignments are independent, so reads are split into
contiguous index ranges, one per rank (one process per GPU); the reference and the 25-byte matrix are replicated.
The only exchange is the gather of per-read results: ONE all-gather per step (RCCL over xGMI on GPUs; gloo in the
CPU tests) of a byte slab that holds a rank's scores and statuses back to back, so the kernel writes its results
straight into the buffer the collective sends (no packing pass, no second collective).
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Reads [first, first+count) of rank `rank`: contiguous ranges [i*n/G, (i+1)*n/G)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    first = (rank * n_total) // world
    last = ((rank + 1) * n_total) // world
    return first, last - first


def shard_capacity(n_total: int, world: int) -> int:
    """Reads of the largest shard (shards differ by at most one read)."""
    return max(shard_range(n_total, r, world)[1] for r in range(world))


def slab_bytes(cap: int) -> int:
    """4*cap bytes of scores, then cap status bytes rounded up to 16 (so every rank's slab starts 16-byte aligned)."""
    cap = max(int(cap), 1)
    return 4 * cap + (cap + 15) // 16 * 16


class ResultSlab:
    """One rank's per-read results as a single byte buffer: `cap` u32 scores, then `cap` u8 statuses.

    `score` and `status` are views into the buffer; passing them as the output arrays of a score call makes the
    kernel write the collective's send buffer directly."""

    def __init__(self, cap: int, device=None):
        import torch

        self.cap = int(cap)
        self.buf = torch.zeros(slab_bytes(self.cap), dtype=torch.uint8, device=device)
        self.score = self.buf[: 4 * max(self.cap, 1)].view(torch.int32)
        self.status = self.buf[4 * max(self.cap, 1) :][: max(self.cap, 1)]


class GatheredResults:
    """All ranks' slabs, [world, slab_bytes(cap)] bytes; rank r's reads are row r (`counts[r]` of them are real)."""

    def __init__(self, world: int, cap: int, counts: List[int], device=None):
        import torch

        self.world, self.cap, self.counts = world, int(cap), list(counts)
        self.buf = torch.zeros(world * slab_bytes(self.cap), dtype=torch.uint8, device=device)

    def _rows(self):
        return self.buf.view(self.world, slab_bytes(self.cap))

    def score_of_rank(self, r: int):
        import torch

        return self._rows()[r, : 4 * max(self.cap, 1)].view(torch.int32)[: self.counts[r]]

    def status_of_rank(self, r: int):
        return self._rows()[r, 4 * max(self.cap, 1) :][: self.counts[r]]

    def scores(self):
        """Scores of all reads in read order (a copy: the shards are `cap`-strided in the gathered buffer)."""
        import torch

        return torch.cat([self.score_of_rank(r) for r in range(self.world)])

    def statuses(self):
        import torch

        return torch.cat([self.status_of_rank(r) for r in range(self.world)])


def gather_slabs(slab: ResultSlab, out: GatheredResults, group=None, async_op: bool = False):
    """The one collective of the path: all-gather of the ranks' result slabs. Returns the work handle when
    `async_op` (the caller keeps `slab` alive and calls .wait() before reusing either buffer)."""
    import torch.distributed as dist

    return dist.all_gather_into_tensor(out.buf, slab.buf, group=group, async_op=async_op)


def all_gather_results(score, status, n_total: int, group=None):
    """Gathers per-read scores (int32/uint32) and statuses (uint8) of all ranks in read order.

    Shards follow shard_range(). Convenience form over gather_slabs() for results that were not written into a slab
    (one copy into the slab, one collective). Works with any torch.distributed backend (nccl = RCCL on ROCm; gloo on CPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return score, status
    counts = [shard_range(n_total, r, world)[1] for r in range(world)]
    cap = max(counts)
    slab = ResultSlab(cap, device=score.device)
    n = score.numel()
    slab.score[:n] = score.view(torch.int32) if score.dtype != torch.int32 else score
    slab.status[:n] = status
    out = GatheredResults(world, cap, counts, device=score.device)
    gather_slabs(slab, out, group=group)
    return out.scores().view(score.dtype), out.statuses()
