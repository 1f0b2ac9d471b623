"""Data-parallel plumbing over torch.distributed (RCCL on ROCm: backend "nccl"; "gloo" in CPU tests).

The path shards by independent units: news of the pool (table mode) and impressions.  The one exchange
step is the all-gather of the per-rank news-embedding shards ``[n_r, D]`` into the full table on every
rank; everything else is local.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .synth import shard_range


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def balanced_news_shards(lengths: np.ndarray, world_size: int, flops_per_len) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) ranges of the news pool with (nearly) equal encoder FLOPs per rank."""
    cost = np.cumsum(np.asarray([flops_per_len(int(l)) for l in np.unique(lengths)], dtype=np.float64)[
        np.searchsorted(np.unique(lengths), lengths)])
    total = cost[-1]
    cuts = [0] + [int(np.searchsorted(cost, total * r / world_size)) for r in range(1, world_size)] + [len(lengths)]
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def equal_news_shards(n_news: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) ranges of ceil(n / W) rows each (the last one shorter): the layout in which ONE
    ``all_gather_into_tensor`` writes every rank's block straight into its place in the table.  With the pool in its
    natural order the lengths are i.i.d., so equal rows are equal FLOPs to within a fraction of a percent;
    ``balanced_news_shards`` is for pools that are ordered by length."""
    mx = -(-n_news // world_size)
    return [(min(r * mx, n_news), min((r + 1) * mx, n_news)) for r in range(world_size)]


def all_gather_table(local: torch.Tensor, shards: List[Tuple[int, int]], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Every rank contributes rows [lo_r, hi_r) of the table; returns the full [N, D] table.

    One ``all_gather_into_tensor`` of equal-size blocks — a single large collective, which on xGMI's point-to-point
    links is what RCCL moves at link rate.  ``local`` may already be the padded [max shard, D] block (rows beyond
    hi - lo are ignored); with ``equal_news_shards`` the received buffer IS the table (no copy, no compaction).  Ragged
    shards (``balanced_news_shards``) are compacted with one gather."""
    rank, ws = world()
    n_total, d = shards[-1][1], local.shape[1]
    mx = max(hi - lo for lo, hi in shards)
    own = shards[rank][1] - shards[rank][0]
    assert local.shape[0] in (own, mx)
    if ws == 1:
        return local[:own]
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx, d), dtype=local.dtype, device=local.device)
        send[:own] = local
    recv = torch.empty((ws * mx, d), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous())
    if all(lo == r * mx for r, (lo, hi) in enumerate(shards)):
        return recv[:n_total]                                   # blocks landed in place
    rows = torch.cat([torch.arange(r * mx, r * mx + (hi - lo), device=local.device) for r, (lo, hi) in enumerate(shards)])
    if out is None:
        return recv.index_select(0, rows)
    torch.index_select(recv, 0, rows, out=out)
    return out


def impression_shard(n_impressions: int) -> Tuple[int, int]:
    rank, ws = world()
    return shard_range(n_impressions, rank, ws)


def allreduce_metric_sums(values: torch.Tensor) -> torch.Tensor:
    """Sum of per-rank (sum nDCG, count, ...) accumulators — the only other collective of the path."""
    _, ws = world()
    if ws > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values
