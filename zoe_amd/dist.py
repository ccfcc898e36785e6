"""Multi-GPU sharding of the batched path: alignments are independent, so reads are split into
contiguous index ranges, one per rank (one process per GPU); the reference and the 25-byte matrix are replicated.
The only exchange is the gather of per-read results (RCCL all-gather over xGMI on GPUs; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Reads [first, first+count) of rank `rank`: contiguous ranges [i*n/G, (i+1)*n/G)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    first = (rank * n_total) // world
    last = ((rank + 1) * n_total) // world
    return first, last - first


def all_gather_results(score, status, n_total: int, group=None):
    """Gathers per-read scores (int32/uint32) and statuses (uint8) of all ranks in read order.

    Shards follow shard_range(); equal shards use one all_gather_into_tensor per array, ragged shards pad to the
    longest shard. Works with any torch.distributed backend (nccl = RCCL on ROCm; gloo on CPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return score, status
    counts = [shard_range(n_total, r, world)[1] for r in range(world)]
    mx = max(counts)
    if min(counts) == mx:
        all_score = torch.empty(world * mx, dtype=score.dtype, device=score.device)
        all_status = torch.empty(world * mx, dtype=status.dtype, device=status.device)
        dist.all_gather_into_tensor(all_score, score.contiguous(), group=group)
        dist.all_gather_into_tensor(all_status, status.contiguous(), group=group)
        return all_score, all_status
    ps = torch.zeros(mx, dtype=score.dtype, device=score.device)
    pt = torch.zeros(mx, dtype=status.dtype, device=status.device)
    ps[: score.numel()] = score
    pt[: status.numel()] = status
    gs = torch.empty(world * mx, dtype=score.dtype, device=score.device)
    gt = torch.empty(world * mx, dtype=status.dtype, device=status.device)
    dist.all_gather_into_tensor(gs, ps, group=group)
    dist.all_gather_into_tensor(gt, pt, group=group)
    parts_s = [gs[r * mx : r * mx + counts[r]] for r in range(world)]
    parts_t = [gt[r * mx : r * mx + counts[r]] for r in range(world)]
    return torch.cat(parts_s), torch.cat(parts_t)
