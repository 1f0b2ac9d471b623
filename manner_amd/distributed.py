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


def device_backend(device_type: str) -> str:
    """Name of the backend that serves tensors of ``device_type`` ("cuda" / "cpu") in the default group.  ``dist.get_backend()``
    returns one name ("nccl", "gloo") or, for a group created without naming one, the composite "cpu:gloo,cuda:nccl"."""
    b = str(dist.get_backend()).lower()
    if ":" in b:
        table = dict(part.split(":", 1) for part in b.split(",") if ":" in part)
        return table.get(device_type, b)
    return b


class MeshTableGather:
    """The table's one exchange step (SURVEY.md §8e phase B), two interchangeable forms over the same in-place ``[N, D]`` table:

    ``exchange == "collective"`` (THE DEFAULT since round 4): the encoder writes its shard's rows into the table, ``wait()`` runs
    ONE ``all_gather_into_tensor`` of equal blocks (``all_gather_table``) — RCCL's own algorithm choice, a code path every RCCL
    installation exercises.

    ``exchange == "mesh"`` (``MANNER_TABLE_EXCHANGE=mesh`` or ``exchange="mesh"``): a DIRECT FULL MESH overlapped with the encoding.
    xGMI is point to point — 7 links per GPU — so the fastest way to replicate ``W`` shards is for every rank to send its block to
    each of the other ``W - 1`` ranks at once, one transfer per link (a ring moves the same bytes in ``W - 1`` serial, per-link-bound
    steps).  A rank's encoder writes piece ``c`` of its own shard STRAIGHT into the table rows (``local_out(c)``), and ``post(c)``
    then — on a side stream, behind an event of the compute stream — sends those rows to every peer and receives the peers' piece
    ``c`` into THEIR rows of the table (contiguous row blocks: no staging copy, no compaction).  All 2(W-1) transfers of a piece go
    out as ONE ``batch_isend_irecv`` group, peers visited in the staggered order rank+k / rank-k so that no link carries two
    messages of a step.  While piece ``c`` is on the links the encoder is already busy with piece ``c + 1``; only the last piece's
    transfer is exposed.  **Status: verified bit-equal to the collective on gloo at world sizes 2, 3, 4 and 8
    (tests/test_host.py); it has never run on RCCL over real xGMI links (no multi-GPU node in rounds 1-4)** — which is why it is not
    the default: `bench.py` measures it NEXT TO the collective and checks the two tables bit for bit before quoting it.

    ``wait()`` is bounded: the host waits for the exchange at most ``timeout_s`` seconds (``MANNER_TABLE_EXCHANGE_TIMEOUT_S``,
    default 120) and raises ``TimeoutError`` — a mismatched point-to-point group then ends the process with an error instead of
    hanging it.  Where GPU point-to-point transfers do not exist (gloo over device tensors: the one-GPU rehearsal) "mesh" falls
    back to the collective and says so in ``exchange_why``."""

    def __init__(self, n_news: int, dim: int, device, dtype: torch.dtype = torch.float32, pieces: int = 4,
                 exchange: Optional[str] = None, timeout_s: Optional[float] = None):
        self.rank, self.ws = world()
        self.shards = equal_news_shards(n_news, self.ws)
        self.pieces = max(1, int(pieces))
        # The table lives inside a [W * mx, D] buffer (mx = rows of the largest shard; `table` is the view of its first N rows): rank r's
        # block starts at row r * mx = shards[r][0], so the collective form gathers IN PLACE — send buffer = this rank's block of the
        # receive buffer, the layout NCCL / RCCL define as the in-place all-gather — with no staging clone and no copy back (round 5;
        # rounds 3-4 cloned the shard, gathered into a fresh buffer and copied 495 MB back per module).
        self.block_rows = max(hi - lo for lo, hi in self.shards)
        self._padded = torch.empty((self.ws * self.block_rows, dim), dtype=dtype, device=device)
        self.table = self._padded[:n_news]
        self._cuda = self.table.is_cuda
        self._works: list = []
        asked = (exchange or os.environ.get("MANNER_TABLE_EXCHANGE", "collective")).lower()
        if asked not in ("mesh", "collective"):
            raise ValueError(f"table exchange {asked!r}: expected 'mesh' or 'collective'")
        self.exchange, self.exchange_why = asked, "requested" if (exchange or "MANNER_TABLE_EXCHANGE" in os.environ) else "default"
        if self.ws == 1:
            self.exchange, self.exchange_why = "none", "world size 1"
        elif asked == "mesh" and self._cuda and device_backend("cuda") != "nccl":
            # gloo has no point-to-point transfers of GPU tensors (the one-GPU rehearsal of the multi-rank logic)
            self.exchange, self.exchange_why = "collective", f"mesh requested, but backend {device_backend('cuda')!r} has no GPU point-to-point transfers"
        self._comm = torch.cuda.Stream(device=device) if (self._cuda and self.exchange == "mesh") else None
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get("MANNER_TABLE_EXCHANGE_TIMEOUT_S", "120"))

    def piece_rows(self, rank: int, c: int) -> Tuple[int, int]:
        """Table rows [a, b) of piece ``c`` of rank ``rank``'s shard (the same split on every rank)."""
        lo, hi = self.shards[rank]
        n = hi - lo
        return lo + n * c // self.pieces, lo + n * (c + 1) // self.pieces

    def local_out(self, c: int) -> torch.Tensor:
        a, b = self.piece_rows(self.rank, c)
        return self.table[a:b]

    def post(self, c: int) -> None:
        """Piece ``c`` of this rank's shard has been ENQUEUED on the current stream: exchange it with every peer (mesh only; the
        collective form moves everything in ``wait()``)."""
        if self.exchange != "mesh":
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

    def _bounded_sync(self, what: str) -> None:
        """Host-side wait for everything enqueued so far on the current stream, at most ``timeout_s`` seconds."""
        if not self._cuda:
            return
        import time
        ev = torch.cuda.Event()
        ev.record()
        t0 = time.monotonic()
        while not ev.query():
            if time.monotonic() - t0 > self.timeout_s:
                raise TimeoutError(f"MeshTableGather.wait(): the {what} exchange of rank {self.rank}/{self.ws} did not complete within "
                                   f"{self.timeout_s:.0f} s (peers out of step, or a transfer that cannot make progress)")
            time.sleep(0.0005)

    def wait(self) -> torch.Tensor:
        """The complete table; the current stream is ordered behind every transfer and the host has seen it complete (bounded)."""
        if self.exchange == "collective":
            mx = self.block_rows
            own = self._padded[self.rank * mx:(self.rank + 1) * mx]          # rows past the shard's end (last rank): padding, sent as is
            dist.all_gather_into_tensor(self._padded, own)                    # in place: own block = output block `rank`
            self._bounded_sync("collective")
            return self.table
        if self.exchange == "mesh":
            import datetime
            for w in self._works:
                if self._cuda:
                    w.wait()                          # RCCL: orders the current stream behind the transfer, does not block the host
                else:                                 # gloo: blocks the host; its own time-out error becomes ours
                    try:
                        done = w.wait(timeout=datetime.timedelta(seconds=self.timeout_s))
                    except RuntimeError as e:
                        done = False
                        cause = str(e).splitlines()[0][:200]
                    else:
                        cause = ""
                    if done is False:
                        raise TimeoutError(f"MeshTableGather.wait(): a transfer of rank {self.rank}/{self.ws} did not complete within "
                                           f"{self.timeout_s:.0f} s (peers out of step?) {cause}")
            self._works = []
            if self._comm is not None:
                torch.cuda.current_stream().wait_stream(self._comm)
            self._bounded_sync("mesh")
        return self.table


def balanced_impression_shards(hist_off: np.ndarray, cand_off: np.ndarray, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) blocks of impressions with (nearly) equal scorer work sum(h_i + c_i) per rank (SURVEY.md §8e,
    phase C): the fused scorer reads one table row per history / candidate occurrence, so occurrences — not impressions — are
    what a rank's time is proportional to.  Every impression lands in exactly one block.  Each cut is the impression boundary
    whose cumulative work is CLOSEST to the r/W mark (round 3 always kept the crossing impression on the left, which with one
    heavy impression could hand a rank nothing: cumulative work [3, 11, 23] at W = 2 gave (0,3),(3,3) where (0,2),(2,3) splits
    11 / 12); with at least W impressions no block is empty — boundaries are forced apart by one impression where the marks
    fall inside the same impression."""
    n = int(len(hist_off)) - 1
    if n <= 0:
        return [(0, 0)] * world_size
    work = (np.asarray(hist_off[1:], dtype=np.int64) - int(hist_off[0])) + (np.asarray(cand_off[1:], dtype=np.int64) - int(cand_off[0]))
    total = int(work[-1])
    cuts = [0]
    for r in range(1, world_size):
        mark = total * r / world_size
        j = int(np.searchsorted(work, mark, side="left"))           # work[j-1] < mark <= work[j]: impression j crosses the mark
        below = float(work[j - 1]) if j > 0 else 0.0                # cumulative work of a cut in front of impression j
        above = float(work[min(j, n - 1)])                          # ... and behind it
        cut = j if (mark - below) <= (above - mark) else j + 1
        if n >= world_size:                                         # keep every block non-empty
            cut = max(cut, cuts[-1] + 1)
            cut = min(cut, n - (world_size - r))
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
