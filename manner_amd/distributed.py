"""Data-parallel plumbing over torch.distributed (RCCL on ROCm: backend "nccl"; "gloo" in CPU tests).

The path shards by independent units: news of the pool (table mode) and impressions.  The one exchange
step is the all-gather of the per-rank news-embedding shards ``[n_r, D]`` into the full table on every
rank; everything else is local.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import os

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


class MeshTableGather:
    """The table's one exchange step as a DIRECT FULL MESH, overlapped with the encoding.

    xGMI is point to point — 7 links per GPU — so the fastest way to replicate ``W`` shards is for every rank to send its
    block to each of the other ``W - 1`` ranks at once, one transfer per link (a ring moves the same bytes in ``W - 1``
    serial, per-link-bound steps).  The table ``[N, D]`` is allocated once on every rank; a rank's encoder writes piece
    ``c`` of its own shard STRAIGHT into the table rows (``local_out(c)``), and ``post(c)`` then — on a side stream, behind
    an event of the compute stream — sends those rows to every peer and receives the peers' piece ``c`` into THEIR rows of
    the table (contiguous row blocks: no staging copy, no compaction).  All 2(W-1) transfers of a piece go out as ONE
    ``batch_isend_irecv`` group, peers visited in the staggered order rank+k / rank-k so that no link carries two messages
    of a step.  While piece ``c`` is on the links the encoder is already busy with piece ``c + 1``; only the last piece's
    transfer is exposed.  ``wait()`` orders the caller's stream behind all of it.  Works on "nccl" (RCCL) and — for the
    CPU tests — "gloo"."""

    def __init__(self, n_news: int, dim: int, device, dtype: torch.dtype = torch.float32, pieces: int = 4):
        self.rank, self.ws = world()
        self.shards = equal_news_shards(n_news, self.ws)
        self.pieces = max(1, int(pieces))
        self.table = torch.empty((n_news, dim), dtype=dtype, device=device)
        self._cuda = self.table.is_cuda
        self._comm = torch.cuda.Stream(device=device) if (self._cuda and self.ws > 1) else None
        self._works: list = []
        # gloo has no point-to-point transfers of GPU tensors (the one-GPU rehearsal of the multi-rank logic): the pieces are
        # then exchanged by ONE all_gather of equal blocks in wait() — same table, no overlap.  RCCL ("nccl") runs the mesh.
        # MANNER_TABLE_EXCHANGE=collective forces that path on RCCL too (one all_gather_into_tensor after the encoding).
        self._collective_fallback = self.ws > 1 and ((self._cuda and dist.get_backend() != "nccl") or
                                                     os.environ.get("MANNER_TABLE_EXCHANGE", "mesh") == "collective")

    def piece_rows(self, rank: int, c: int) -> Tuple[int, int]:
        """Table rows [a, b) of piece ``c`` of rank ``rank``'s shard (the same split on every rank)."""
        lo, hi = self.shards[rank]
        n = hi - lo
        return lo + n * c // self.pieces, lo + n * (c + 1) // self.pieces

    def local_out(self, c: int) -> torch.Tensor:
        a, b = self.piece_rows(self.rank, c)
        return self.table[a:b]

    def post(self, c: int) -> None:
        """Piece ``c`` of this rank's shard has been ENQUEUED on the current stream: exchange it with every peer."""
        if self.ws == 1 or self._collective_fallback:
            return
        ops = []
        for k in range(1, self.ws):
            dst, src = (self.rank + k) % self.ws, (self.rank - k) % self.ws
            a, b = self.piece_rows(self.rank, c)
            if b > a:
                ops.append(dist.P2POp(dist.isend, self.table[a:b], dst))
            a, b = self.piece_rows(src, c)
            if b > a:
                ops.append(dist.P2POp(dist.irecv, self.table[a:b], src))
        if not ops:
            return
        if self._cuda:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ev)          # the transfers start when the piece's last kernel has finished
                self._works += dist.batch_isend_irecv(ops)
        else:
            self._works += dist.batch_isend_irecv(ops)

    def wait(self) -> torch.Tensor:
        """The complete table; the current stream is ordered behind every transfer."""
        if self._collective_fallback:
            lo, hi = self.shards[self.rank]
            self.table.copy_(all_gather_table(self.table[lo:hi].clone(), self.shards))
            return self.table
        for w in self._works:
            w.wait()
        self._works = []
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
        return self.table


def balanced_impression_shards(hist_off: np.ndarray, cand_off: np.ndarray, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) blocks of impressions with (nearly) equal scorer work sum(h_i + c_i) per rank (SURVEY.md §8e,
    phase C): the fused scorer reads one table row per history / candidate occurrence, so occurrences — not impressions — are
    what a rank's time is proportional to.  Every impression lands in exactly one block; blocks may be empty only when there
    are fewer impressions than ranks."""
    n = int(len(hist_off)) - 1
    work = (np.asarray(hist_off[1:], dtype=np.int64) - int(hist_off[0])) + (np.asarray(cand_off[1:], dtype=np.int64) - int(cand_off[0]))
    total = int(work[-1]) if n > 0 else 0
    cuts = [0]
    for r in range(1, world_size):
        cut = int(np.searchsorted(work, total * r / world_size, side="left")) + 1 if n > 0 else 0   # first impression that crosses the mark stays left
        cuts.append(min(max(cut, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def impression_shard(n_impressions: int) -> Tuple[int, int]:
    rank, ws = world()
    return shard_range(n_impressions, rank, ws)


def allreduce_metric_sums(values: torch.Tensor) -> torch.Tensor:
    """Sum of per-rank (sum nDCG, count, ...) accumulators — the only other collective of the path."""
    _, ws = world()
    if ws > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values
