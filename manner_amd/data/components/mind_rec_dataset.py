"""Device-side collate for the evaluation path.

The reference assembles every batch on the host: ``MINDRecDatasetTest.__getitem__`` does two
``DataFrame.loc`` lookups per impression (manner/data/components/mind_rec_dataset.py:87-99) and
``MINDCollate.__call__`` concatenates the frames and re-tokenises every news of the batch
(:114-137, :146-168).  Here the news are tokenised once into a device-resident ``NewsStore``, the
behaviours are parsed once into CSR index arrays (``ParsedBehaviors``), and ``DeviceCollate`` builds the
same ``MINDRecBatch`` tensors with four HIP kernels (``csrc/collate.hip``) — the host only slices offsets.

Host code is numpy; everything on the GPU goes through ``manner_amd.hip`` (no CPU fallback).
"""
from __future__ import annotations

import io
import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

from ... import hip
from .mind_batch import MINDRecBatch


# ---------------------------------------------------------------------------------------------- behaviours
@dataclass
class ParsedBehaviors:
    """CSR form of the behaviours frame (columns user / history / candidates / labels, mind_dataframe.py:359)."""
    users: np.ndarray        # int64 [B]
    hist_rows: np.ndarray    # int32 [sum h_i]  store rows, history already cut to max_history_length
    hist_off: np.ndarray     # int64 [B+1]
    cand_rows: np.ndarray    # int32 [sum c_i]
    cand_off: np.ndarray     # int64 [B+1]
    labels: np.ndarray       # float32 [sum c_i]

    def __len__(self) -> int:
        return int(self.users.shape[0])


def _ids_of_list_literal(field: str) -> List[str]:
    """The reference's converter for the cached frame (mind_dataframe.py:281-284):
    ``x.strip("[]").replace("'", "").split(", ")`` — note that "[]" yields [""]."""
    return field.strip("[]").replace("'", "").split(", ")


def parse_behaviors(source: Union[str, io.TextIOBase, Iterable[str]], nid2row: Dict[str, int], max_history_length: int,
                    uid2index: Optional[Dict[str, int]] = None) -> ParsedBehaviors:
    """Parse a behaviours file into CSR arrays.

    Accepts both wire formats of the reference: the cached ``parsed_behaviors.tsv`` (header
    ``user / history / candidates / labels``, lists serialised as ``['N1', 'N2']`` / ``[1, 0]`` —
    mind_dataframe.py:278-288) and the raw MIND ``behaviors.tsv`` (``impid uid time history impressions``,
    impressions ``N1-1 N2-0``; rows without history dropped, ``user = uid2index.get(uid, 0)`` — :291-357).
    History is cut to its first ``max_history_length`` items as MINDRecDatasetTest.__getitem__ does (:91).
    Unknown news ids raise KeyError, like ``news.loc``.
    """
    if isinstance(source, (str, os.PathLike)):
        with open(source, "r", encoding="utf-8") as f:
            return parse_behaviors(f, nid2row, max_history_length, uid2index)
    users: List[int] = []
    hist: List[int] = []
    cand: List[int] = []
    labels: List[float] = []
    hist_off = [0]
    cand_off = [0]
    parsed_format: Optional[bool] = None
    uid2index = uid2index or {}
    for line in source:
        line = line.rstrip("\n").rstrip("\r")
        if not line:
            continue
        cols = line.split("\t")
        if parsed_format is None:
            parsed_format = cols[:4] == ["user", "history", "candidates", "labels"]
            if parsed_format:
                continue
        if parsed_format:
            user = int(cols[0])
            h_ids = _ids_of_list_literal(cols[1])
            c_ids = _ids_of_list_literal(cols[2])
            lab = [int(x) for x in cols[3].strip("[]").split(", ")]
        else:
            if len(cols) < 5:
                raise ValueError(f"behaviors.tsv row with {len(cols)} columns")
            h_ids = cols[3].split()
            if not h_ids:                              # "drop interactions of users without history" (:311-314)
                continue
            user = int(uid2index.get(cols[1], 0))
            imps = cols[4].split()
            c_ids = [x.split("-")[0] for x in imps]
            lab = [int(x.split("-")[1]) for x in imps]
        h_ids = h_ids[:max_history_length]
        users.append(user)
        hist.extend(nid2row[n] for n in h_ids)
        cand.extend(nid2row[n] for n in c_ids)
        labels.extend(lab)
        hist_off.append(len(hist))
        cand_off.append(len(cand))
    if len(labels) != len(cand):
        raise ValueError("candidates and labels differ in length")
    return ParsedBehaviors(np.asarray(users, np.int64), np.asarray(hist, np.int32), np.asarray(hist_off, np.int64),
                           np.asarray(cand, np.int32), np.asarray(cand_off, np.int64), np.asarray(labels, np.float32))


# ---------------------------------------------------------------------------------------------- news store
class NewsStore:
    """Pre-tokenised news, resident in HBM.  Row order defines ``nid2row``.

    ``token_ids`` holds what ``tokenizer(text, truncation=True)`` returns for each news (special tokens
    included, cut to ``tokenizer_max_length`` = 96, configs/data/mind_rec.yaml:41); ``entities`` the filtered
    entity index lists (title + abstract concatenated when both aspects are used, mind_rec_dataset.py:147-158).
    """

    def __init__(self, nids: Sequence[str], token_ids: Sequence[Sequence[int]], pad_id: int,
                 entities: Optional[Sequence[Sequence[int]]] = None, category: Optional[Sequence[int]] = None,
                 sentiment: Optional[Sequence[int]] = None, sentiment_score: Optional[Sequence[float]] = None,
                 device: Union[str, torch.device] = "cuda"):
        n = len(nids)
        self.nid2row = {nid: i for i, nid in enumerate(nids)}
        if len(self.nid2row) != n:
            raise ValueError("duplicate news ids")
        self.pad_id = int(pad_id)
        self.lengths = np.fromiter((len(t) for t in token_ids), np.int32, n)
        width = max(int(self.lengths.max()) if n else 0, 1)
        ids = np.full((n, width), pad_id, np.int32)
        for i, t in enumerate(token_ids):
            ids[i, :len(t)] = t
        entities = entities if entities is not None else [[]] * n
        self.ent_counts = np.fromiter((len(e) for e in entities), np.int32, n)
        ent = np.zeros((n, max(int(self.ent_counts.max()) if n else 0, 1)), np.int32)
        for i, e in enumerate(entities):
            ent[i, :len(e)] = e
        zeros = np.zeros(n, np.int32)
        dev = torch.device(device)
        self.ids_d = torch.from_numpy(ids).to(dev)
        self.len_d = torch.from_numpy(self.lengths).to(dev)
        self.ent_d = torch.from_numpy(ent).to(dev)
        self.cnt_d = torch.from_numpy(self.ent_counts).to(dev)
        self.cat_d = torch.from_numpy(np.asarray(category if category is not None else zeros, np.int32)).to(dev)
        self.sent_d = torch.from_numpy(np.asarray(sentiment if sentiment is not None else zeros, np.int32)).to(dev)
        self.score_d = torch.from_numpy(np.asarray(sentiment_score if sentiment_score is not None else zeros, np.float32)).to(dev)
        self.device = dev

    @classmethod
    def from_arrays(cls, ids: np.ndarray, lengths: np.ndarray, pad_id: int, device: Union[str, torch.device] = "cuda"):
        """Store over an already padded token matrix int32 [N, L] (row i = news i; nids are the row numbers)."""
        self = cls.__new__(cls)
        n = ids.shape[0]
        dev = torch.device(device)
        self.nid2row = None
        self.pad_id = int(pad_id)
        self.lengths = np.ascontiguousarray(lengths, np.int32)
        self.ent_counts = np.zeros(n, np.int32)
        self.ids_d = torch.from_numpy(np.ascontiguousarray(ids, np.int32)).to(dev)
        self.len_d = torch.from_numpy(self.lengths).to(dev)
        self.ent_d = torch.zeros((n, 1), dtype=torch.int32, device=dev)
        self.cnt_d = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.cat_d = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.sent_d = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.score_d = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.device = dev
        return self

    def __len__(self) -> int:
        return int(self.lengths.shape[0])


# ---------------------------------------------------------------------------------------------- collate
class DeviceCollate:
    """``MINDCollate`` with the store on the device: ``collate(indices) -> MINDRecBatch``.

    ``indices`` is the list of behaviour rows a DataLoader batch holds (any order); a contiguous ``range``
    is served from device-resident slices without any host-to-device copy.
    """

    def __init__(self, store: NewsStore, behaviors: ParsedBehaviors):
        self.store, self.bhv = store, behaviors
        dev = store.device
        self.hist_rows_d = torch.from_numpy(behaviors.hist_rows).to(dev)
        self.cand_rows_d = torch.from_numpy(behaviors.cand_rows).to(dev)
        self.labels_d = torch.from_numpy(behaviors.labels).to(dev)
        self.users_d = torch.from_numpy(behaviors.users).to(dev)

    def _side(self, rows_d: torch.Tensor, rows_h: np.ndarray, sizes: np.ndarray):
        st = self.store
        off = np.zeros(sizes.shape[0] + 1, np.int64)
        np.cumsum(sizes, out=off[1:])
        off_d = torch.from_numpy(off).to(st.device, non_blocking=True)
        seg = hip.collate_segments_sized(off_d, int(off[-1]))
        lp = int(st.lengths[rows_h].max()) if rows_h.size else 0          # tokenizer padding=True: batch max
        width = int(st.ent_counts[rows_h].max()) if rows_h.size else 0    # _tokenize_entities: batch max
        ids, mask = hip.collate_text(st.ids_d, st.len_d, rows_d, lp, st.pad_id)
        ent = hip.collate_entities(st.ent_d, st.cnt_d, rows_d, width)
        cat, sent, score = hip.collate_aspects(st.cat_d, st.sent_d, st.score_d, rows_d)
        x = {"text": {"input_ids": ids, "attention_mask": mask}, "entities": ent, "category": cat, "sentiment": sent,
             "sentiment_score": score}
        return seg, x

    def __call__(self, indices: Union[range, Sequence[int]]) -> MINDRecBatch:
        b = self.bhv
        contiguous = isinstance(indices, range) and indices.step == 1 and len(indices) > 0
        if contiguous:
            i0, i1 = indices.start, indices.stop
            h0, h1, c0, c1 = int(b.hist_off[i0]), int(b.hist_off[i1]), int(b.cand_off[i0]), int(b.cand_off[i1])
            hist_h, cand_h = b.hist_rows[h0:h1], b.cand_rows[c0:c1]
            hist_d, cand_d = self.hist_rows_d[h0:h1], self.cand_rows_d[c0:c1]
            labels, users = self.labels_d[c0:c1], self.users_d[i0:i1]
            hs, cs = np.diff(b.hist_off[i0:i1 + 1]), np.diff(b.cand_off[i0:i1 + 1])
        else:
            idx = np.asarray(list(indices), np.int64)
            hs, cs = (b.hist_off[idx + 1] - b.hist_off[idx]), (b.cand_off[idx + 1] - b.cand_off[idx])
            take = lambda off, sizes: (np.concatenate([np.arange(off[i], off[i] + s) for i, s in zip(idx, sizes)])
                                       if idx.size else np.zeros(0, np.int64))
            hsel, csel = take(b.hist_off, hs), take(b.cand_off, cs)
            hist_h, cand_h = b.hist_rows[hsel], b.cand_rows[csel]
            dev = self.store.device
            hist_d, cand_d = torch.from_numpy(hist_h).to(dev), torch.from_numpy(cand_h).to(dev)
            labels, users = torch.from_numpy(b.labels[csel]).to(dev), torch.from_numpy(b.users[idx]).to(dev)
        batch_hist, x_hist = self._side(hist_d, hist_h, hs)
        batch_cand, x_cand = self._side(cand_d, cand_h, cs)
        batch = MINDRecBatch(batch_hist=batch_hist, batch_cand=batch_cand, x_hist=x_hist, x_cand=x_cand, labels=labels,
                             users=users)
        # host-known extras (not part of the reference's TypedDict; its consumers ignore them): the widths of the dense
        # [B, max, *] views, so that K9 needs no device read-back (to_dense_batch computes them with a sync)
        batch["hist_max"] = int(hs.max()) if hs.size else 0
        batch["cand_max"] = int(cs.max()) if cs.size else 0
        return batch
