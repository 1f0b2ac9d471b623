"""Device-side composition of the hot path above the C ABI.

* ``cr_forward`` / ``ensemble_forward`` restate ``CRModule.forward`` (reference
  manner/models/cr_module.py:105-131) and ``EnsembleModule.forward`` / ``_submodel_forward``
  (reference manner/models/ensemble_module.py:95-151) on the fused HIP kernels, returning the same
  dense ``[B, Cmax]`` score matrix (padded slots: 0 for the CR-Module, the z-scored zero for the ensemble)
  — mode R of SURVEY.md §8d: every history and candidate occurrence is encoded.
* ``encode_table`` / ``score_impressions`` are the table architecture (mode T): each unique news is
  encoded once per module into ``[N_news, D]``, impressions are scored by index.  Valid for
  ``use_entities=False`` only.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import hip, train

Tensor = torch.Tensor


def segment_offsets(batch: Tensor, num_segments: int) -> Tensor:
    """Sorted segment ids (reference ``_make_batch_assignees``, mind_rec_dataset.py:171-174) -> CSR
    offsets int64 [B+1], on device, no host sync: off[i] = first position whose id is >= i (the ids are sorted ascending, as the
    reference's collate produces them).  Round 5: ``torch.bincount`` — used here before — reads its input's maximum back to the host
    to size its output, i.e. it WAITS for everything enqueued so far; in a training step that was the whole encoder forward
    (2.7 ms of host stall per call, after which the GPU idled through the backward's enqueue latency)."""
    marks = torch.arange(num_segments + 1, dtype=batch.dtype, device=batch.device)
    return torch.searchsorted(batch.contiguous(), marks, right=False).to(torch.int64)


def _width(batch: Dict, key: str, off: Tensor) -> int:
    """Width of a dense [B, max, *] view: the collate's host-known maximum when the batch carries it (``hist_max`` /
    ``cand_max``, set by DeviceCollate), else one device read — the same sync ``to_dense_batch`` performs."""
    w = batch.get(key) if hasattr(batch, "get") else None
    if w is not None:
        return int(w)
    return int((off[1:] - off[:-1]).max()) if off.numel() > 1 else 0


def ragged_to_dense(values: Tensor, off: Tensor, width: Optional[int] = None, fill: Optional[Tensor] = None) -> Tensor:
    """K9 for a ragged vector / row matrix: [sum c_i, *] -> [B, Cmax, *] (``to_dense_batch`` layout) through
    ``manner_hip_to_dense``; padded slots hold 0, or ``fill[b]``."""
    if width is None:
        width = int((off[1:] - off[:-1]).max()) if off.numel() > 1 else 0   # the same sync to_dense_batch performs
    return hip.to_dense(values, off, width, fill=fill)


def _late_fusion_ragged(hist_vec: Tensor, cand_vec: Tensor, hist_off: Tensor, cand_off: Tensor) -> Tensor:
    table = torch.cat([hist_vec, cand_vec], dim=0)
    nh, nc = hist_vec.shape[0], cand_vec.shape[0]
    hidx = torch.arange(nh, dtype=torch.int32, device=table.device)
    cidx = torch.arange(nh, nh + nc, dtype=torch.int32, device=table.device)
    return hip.score_late_fusion(table, hidx, hist_off, cidx, cand_off, total_cand=nc)


def cr_forward(news_encoder, batch: Dict, late_fusion: bool = True, user_encoder=None, click_predictor=None,
               dense: bool = True) -> Tensor:
    """CRModule.forward: scores [B, Cmax] (or the ragged [sum c_i] vector with ``dense=False``).  No torch indexing
    and — when the batch carries ``hist_max`` / ``cand_max`` — no host synchronisation.  Padded slots are exactly 0, as in the
    reference — but for an impression WITHOUT history rows (which the reference's loader drops, mind_dataframe.py:313): its real
    candidates score NaN here as there (0 / 0), its padded slots 0 here and NaN there (NaN user . zero vector)."""
    nb = batch["users"].numel() if "users" in batch and batch["users"] is not None else int(batch["batch_cand"].max()) + 1
    hip.status_poll(batch["batch_cand"].device)
    hist_vec = news_encoder(batch["x_hist"])
    cand_vec = news_encoder(batch["x_cand"])
    hist_off = segment_offsets(batch["batch_hist"], nb)
    cand_off = segment_offsets(batch["batch_cand"], nb)
    if late_fusion:
        ragged = _late_fusion_ragged(hist_vec, cand_vec, hist_off, cand_off)
    else:
        # early fusion: the unmasked additive pooler sees the ZERO-PADDED history
        hist_dense = hip.to_dense(hist_vec, hist_off, _width(batch, "hist_max", hist_off))
        user = user_encoder(hist_dense)
        if click_predictor is not None and dense:
            # the reference's own call shape: DotProduct(user [B,1,D], cand^T [B,D,Cmax]) on the dense candidates
            cand_dense = hip.to_dense(cand_vec, cand_off, _width(batch, "cand_max", cand_off))
            return click_predictor(user.unsqueeze(1), cand_dense.permute(0, 2, 1))
        cidx = torch.arange(cand_vec.shape[0], dtype=torch.int32, device=cand_vec.device)
        ragged = hip.score_user(cand_vec, user, cidx, cand_off)
    hip.status_arm(cand_vec.device)
    return hip.to_dense(ragged, cand_off, _width(batch, "cand_max", cand_off)) if dense else ragged


def _cat_tokens(a: Dict, b: Dict, pad_id: int) -> Dict:
    """Two tokenised news sets as one call: rows of `a` then rows of `b`, right-padded to the longer padded length."""
    la, lb = a["input_ids"].shape[1], b["input_ids"].shape[1]
    lp = max(la, lb)

    def pad(t, l, value):
        return t if l == lp else torch.nn.functional.pad(t, (0, lp - l), value=value)

    return {"input_ids": torch.cat([pad(a["input_ids"], la, pad_id), pad(b["input_ids"], lb, pad_id)]),
            "attention_mask": torch.cat([pad(a["attention_mask"], la, 0), pad(b["attention_mask"], lb, 0)])}


def encode_hist_and_cand(news_encoder, x_hist: Dict, x_cand: Dict):
    """The two news_encoder calls of CRModule.forward (cr_module.py:107,113).  Without entities a news embedding does not
    depend on its batch, so both sets go through ONE call — one pass over the weights, GEMMs twice as tall —
    and are split afterwards; with use_entities=True the entity attention couples the news of a call (Q1) and the
    reference's two separate calls are kept."""
    keyed = isinstance(x_hist, dict) and "text" in x_hist
    if getattr(news_encoder, "use_entities", False):
        return news_encoder(x_hist), news_encoder(x_cand)
    th, tc = (x_hist["text"], x_cand["text"]) if keyed else (x_hist, x_cand)
    pad_id = getattr(getattr(getattr(news_encoder, "text_encoder", None), "plm_model", None), "cfg", None)
    merged = _cat_tokens(th, tc, pad_id.pad_id if pad_id is not None else 0)
    both = news_encoder({"text": merged} if keyed else merged)
    n_hist = th["input_ids"].shape[0]
    return both[:n_hist], both[n_hist:]


def cr_train_step(news_encoder, batch: Dict, supcon: bool = True, temperature: float = 0.1):
    """CRModule.model_step for training (cr_module.py:140-171 with late_fusion=True, the shipped MANNeR setting): the
    encoder in train() mode, the fused late-fusion scorer and the loss, all with autograd on the HIP engine.
    Returns (loss, ragged scores [sum c_i] detached, cand_off) — call ``loss.backward()`` and step the reference's optimiser."""
    nb = batch["users"].numel() if "users" in batch and batch["users"] is not None else int(batch["batch_cand"].max()) + 1
    hip.status_poll(batch["batch_cand"].device)       # (encode_train arms the word again after its own kernels)
    hist_off = segment_offsets(batch["batch_hist"], nb)
    cand_off = segment_offsets(batch["batch_cand"], nb)
    hist_vec, cand_vec = encode_hist_and_cand(news_encoder, batch["x_hist"], batch["x_cand"])
    scores = train.late_fusion_scores(hist_vec, hist_off, cand_vec, cand_off)
    c_max = None if supcon else _width(batch, "cand_max", cand_off)
    loss, _ = train.model_step_loss(scores, batch["labels"].to(torch.float32), cand_off, supcon=supcon, temperature=temperature,
                                    c_max=c_max)
    return loss, scores.detach(), cand_off


def a_train_step(news_encoder, batch: Dict, temperature: float = 0.1):
    """AModule.model_step for training (a_module.py:102-108): news embeddings in train() mode and the supervised contrastive
    loss over their aspect labels, both with autograd on the HIP engine.  Returns (loss, embeddings detached)."""
    emb = news_encoder(batch["news"])
    loss, _ = train.supcon_embedding_loss(emb, batch["labels"], temperature=temperature)
    return loss, emb.detach()


def ensemble_forward(news_encoders: Sequence, batch: Dict, weights: Sequence[float], dense: bool = True) -> Tensor:
    """EnsembleModule.forward: CR scores + weighted A-module scores, each z-normalised per impression.
    ``news_encoders[0]`` is the CR-Module's encoder; a zero weight skips that module's encoder entirely
    (ensemble_module.py:100,105).  ``dense=True`` returns the reference's [B, Cmax] matrix slot for slot: its z-score
    runs over the whole zero-padded row (ensemble_module.py:145-149), so PADDED slots hold
    ``sum_k w_k (0 - mean_k) / std_k`` rather than 0 (the reference's consumers mask them away; they are reproduced
    here so that the output is identical, not merely equivalent)."""
    nb = batch["users"].numel()
    hip.status_poll(batch["users"].device)
    hist_off = segment_offsets(batch["batch_hist"], nb)
    cand_off = segment_offsets(batch["batch_cand"], nb)
    planes, used = [], []
    for k, enc in enumerate(news_encoders):
        if k > 0 and weights[k - 1] == 0:
            continue
        planes.append(_late_fusion_ragged(enc(batch["x_hist"]), enc(batch["x_cand"]), hist_off, cand_off))
        if k > 0:
            used.append(weights[k - 1])
    hip.status_arm(cand_off.device)
    if not dense:
        return hip.zscore_fuse(torch.stack(planes), used, cand_off)
    fused, pad = hip.zscore_fuse(torch.stack(planes), used, cand_off, with_pad_value=True)
    return hip.to_dense(fused, cand_off, _width(batch, "cand_max", cand_off), fill=pad)


# ------------------------------------------------------------------------------------- table mode

def encode_table(encoder: hip.HipEncoder, ids: Tensor, mask: Tensor, precision: str = "bf16",
                 host_lengths: Optional[np.ndarray] = None, max_chunk_tokens: int = 65536,
                 out: Optional[Tensor] = None) -> Tensor:
    """Encode a (shard of the) news pool once: [N, Lp] tokens -> [N, D] float32 table rows."""
    return encoder.encode_cls(ids, mask, precision=precision, host_lengths=host_lengths,
                              max_chunk_tokens=max_chunk_tokens, out=out)


def score_impressions(tables: Sequence[Tensor], imp: Dict[str, Tensor], weights: Sequence[float] = (),
                      labels: Optional[Tensor] = None, k: int = 10, fused: Optional[bool] = None) -> Dict[str, Tensor]:
    """Score impressions against per-module tables, fuse, rank.  ``imp``: hist_idx/cand_idx int32,
    hist_off/cand_off int64 (device).  With a single table and no weights the scores are the raw
    late-fusion dot products (CRModule.forward); otherwise the ensemble's z-scored fusion.
    ``fused``: SURVEY §8e phase C as ONE launch (``hip.score_fuse_rank``: float32 tables with D = 768 / 1024, the K score planes stay in
    LDS) or as K scorer launches -> ``zscore_fuse`` -> ``rank_ndcg``; the two give the same bits.  Default (MANNER_PHASE_C=auto): one
    launch for one or two tables or tables that fit the Infinity Cache together, separate launches otherwise; MANNER_PHASE_C=fused /
    separate force either."""
    import os
    hip.status_poll(tables[0].device)                 # an out-of-range news index of an earlier call raises here (IndexError in the reference)
    can_fuse = all(isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] in (768, 1024) for t in tables) \
        and 1 <= len(tables) <= 9
    if fused is None:
        # One launch wins for one or two tables (K = 1: 0.99 vs 1.13 ms MIND-small, 5.8 vs 7.0 ms on the 660 MB roberta-large table;
        # K = 2: 1.82 vs 1.91 ms) and for tables that stay cache-resident together; with three 495 MB tables it interleaves 1.5 GB of row
        # gathers per impression where the separate launches sweep ONE table at a time through the 256 MiB Infinity Cache: 15.0 vs 13.6 ms
        # for the MIND-large dev set (profiles/r4_final/bench_config3.json) — so the default follows the footprint.
        env = os.environ.get("MANNER_PHASE_C", "auto")
        small = len(tables) <= 2 or sum(t.numel() * 4 for t in tables) <= 256 * 2 ** 20
        fused = can_fuse and env != "separate" and (env == "fused" or small)
    if fused:
        if not can_fuse:
            raise ValueError("score_impressions(fused=True): float32 tables with D = 768 or 1024 only")
        res = hip.score_fuse_rank(tables, list(weights), imp["hist_idx"], imp["hist_off"], imp["cand_idx"], imp["cand_off"], labels=labels, k=k)
        hip.status_arm(tables[0].device)
        return res
    planes = []
    for j, t in enumerate(tables):
        if j > 0 and weights[j - 1] == 0:
            continue
        planes.append(hip.score_late_fusion(t, imp["hist_idx"], imp["hist_off"], imp["cand_idx"], imp["cand_off"]))
    hip.status_arm(tables[0].device)
    if len(tables) == 1 and len(weights) == 0:
        scores = planes[0]
    else:
        scores = hip.zscore_fuse(torch.stack(planes), [w for w in weights if w != 0], imp["cand_off"])
    topk, ndcg, mrr = hip.rank_ndcg(scores, labels, imp["cand_off"], k, with_mrr=True)
    return {"scores": scores, "topk": topk, "ndcg": ndcg, "mrr": mrr}


# ------------------------------------------------------------------------------------- epoch-end metrics

def epoch_end_metrics(scores: Tensor, labels: Tensor, cand_off: Tensor, *, cand_categories: Optional[Tensor] = None,
                       cand_sentiments: Optional[Tensor] = None, hist_categories: Optional[Tensor] = None,
                       hist_sentiments: Optional[Tensor] = None, hist_off: Optional[Tensor] = None,
                       num_categ_classes: int = 19, num_sent_classes: int = 4, with_auc_mrr: bool = True,
                       prefix: str = "test/") -> Dict[str, Tensor]:
    """What ``on_test_epoch_end`` logs, from the ragged epoch tensors and entirely on the device:

    * CRModule (cr_module.py:78-87, 264-273): ``auc``, ``mrr``, ``ndcg@5``, ``ndcg@10``;
    * EnsembleModule (ensemble_module.py:50-84, 214-252): ``ndcg@5/10`` plus, when the aspect tensors are given,
      ``categ_div@5/10``, ``sent_div@5/10`` (manner/metrics/functional.py:8-28) and ``categ_pers@5/10``,
      ``sent_pers@5/10`` (:31-62).
    Retrieval metrics are means over impressions; AUC is one curve over all pairs (hip.auc).  Keys carry ``prefix``.
    """
    out: Dict[str, Tensor] = {}
    tops = {}
    for k in (5, 10):
        topk, ndcg, mrr = hip.rank_ndcg(scores, labels, cand_off, k, with_mrr=True)
        tops[k] = topk
        out[f"{prefix}ndcg@{k}"] = ndcg.mean()
        if k == 10 and with_auc_mrr:
            out[f"{prefix}mrr"] = mrr.mean()
    if with_auc_mrr:
        out[f"{prefix}auc"] = hip.auc(scores, labels)
    for name, cand, hist, classes in (("categ", cand_categories, hist_categories, num_categ_classes),
                                      ("sent", cand_sentiments, hist_sentiments, num_sent_classes)):
        if cand is None:
            continue
        for k in (5, 10):
            div, pers = hip.aspect_metrics(tops[k], cand.to(torch.int32), cand_off, classes,
                                           None if hist is None else hist.to(torch.int32), hist_off)
            out[f"{prefix}{name}_div@{k}"] = div.mean()
            if pers is not None:
                out[f"{prefix}{name}_pers@{k}"] = pers.mean()
    return out
