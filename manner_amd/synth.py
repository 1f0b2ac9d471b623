"""Synthetic MIND-shaped inputs.

No MIND data or tokenizer is reachable offline, so benches and parity tests run
on seeded synthetic inputs of the shapes the reference's collate produces
(reference manner/data/components/mind_rec_dataset.py:114-174):
``input_ids`` / ``attention_mask`` right-padded int64 [N, Lp] and ragged
history / candidate index lists per impression (CSR offsets here; the reference
uses sorted segment ids, ``segment_ids`` converts).
"""
from __future__ import annotations

from typing import Dict, Sequence, Optional

import numpy as np

from .config import ARCH_BERT, EncoderConfig

MIND_SMALL = {"n_news": 65238, "n_impressions": 73152}
MIND_LARGE = {"n_news": 161013, "n_impressions": 376471}


def _rng(seed: int, stream: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, stream]))


def synth_lengths(n: int, seed: int, max_len: int = 96, profile: str = "title") -> np.ndarray:
    g = _rng(seed, 1)
    title = np.clip(np.rint(g.lognormal(np.log(16.0), 0.35, n)), 5, max_len)
    if profile == "title":
        return title.astype(np.int64)
    if profile == "title_abstract":
        abstract = np.clip(np.rint(g.lognormal(np.log(60.0), 0.6, n)), 0, 400)
        return np.minimum(max_len, title + abstract + 1).astype(np.int64)
    raise ValueError(profile)


def synth_news_tokens(n: int, cfg: EncoderConfig, seed: int = 42, max_len: int = 96,
                      profile: str = "title", lengths: Optional[np.ndarray] = None,
                      pad_to: Optional[int] = None):
    """ids, mask int64 [n, Lp]; Lp = longest news (tokenizer ``padding=True``) or ``pad_to``."""
    if lengths is None:
        lengths = synth_lengths(n, seed, max_len, profile)
    lengths = np.asarray(lengths, dtype=np.int64)
    assert lengths.shape == (n,) and lengths.min() >= 2
    lp = int(pad_to or lengths.max())
    g = _rng(seed, 2)
    lo = min(1000, cfg.vocab // 2)
    ids = g.integers(lo, cfg.vocab, size=(n, lp), dtype=np.int64)
    cls_id, sep_id = (101, 102) if cfg.arch == ARCH_BERT else (0, 2)
    col = np.arange(lp)[None, :]
    mask = (col < lengths[:, None]).astype(np.int64)
    ids[:, 0] = cls_id
    ids[np.arange(n), lengths - 1] = sep_id
    ids[mask == 0] = cfg.pad_id
    return ids, mask


def synth_impressions(n_imp: int, n_news: int, seed: int = 42, max_hist: int = 50,
                      max_cand: int = 300, zipf_a: float = 1.1, distinct_candidates: bool = True) -> Dict[str, np.ndarray]:
    """Ragged impressions over a news pool: hist/cand CSR offsets (int64 [B+1]), int32 indices
    and float32 labels (1 positive + Bernoulli(0.04) extras, >= 1 positive per impression).

    ``distinct_candidates`` (default): the candidates of ONE impression are drawn without replacement — a MIND impression
    lists distinct news (the reference's ``behaviors.tsv`` rows, mind_dataframe.py:278-288); the Zipf draw with replacement
    of rounds 1-3 repeated popular news inside an impression, which made exact score ties that only rounding noise orders.
    Histories keep the plain draw (a click history may repeat; a repeat only re-weights the mean).  Candidate counts are
    capped at the pool size."""
    g = _rng(seed, 3)
    h = np.clip(np.rint(g.lognormal(np.log(22.0), 0.9, n_imp)), 1, max_hist).astype(np.int64)
    c = np.clip(np.rint(g.lognormal(np.log(24.0), 0.9, n_imp)), 2, max_cand).astype(np.int64)
    if distinct_candidates:
        c = np.minimum(c, n_news)
    hist_off = np.concatenate([[0], np.cumsum(h)]).astype(np.int64)
    cand_off = np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
    # Zipf(a) popularity over the pool through a seeded rank -> news permutation
    cdf = np.cumsum(np.arange(1, n_news + 1, dtype=np.float64) ** (-zipf_a))
    cdf /= cdf[-1]
    perm = g.permutation(n_news).astype(np.int32)

    def draw(m):
        return perm[np.minimum(np.searchsorted(cdf, g.random(m)), n_news - 1)]

    hist_idx = draw(int(hist_off[-1]))
    cand_idx = draw(int(cand_off[-1]))
    if distinct_candidates:
        # rejection: every later occurrence of a news inside its impression is redrawn until none is left (the first occurrence keeps
        # its place, so the popular news stay as popular as the Zipf law makes them); impressions that ask for more than half the pool
        # (tiny test pools) are completed from a seeded permutation of the news they lack instead of waiting for the tail
        seg = np.repeat(np.arange(n_imp, dtype=np.int64), c)

        def later_occurrences(elems):
            """Positions (among ``elems``) holding a news that an earlier position of the same impression already holds."""
            key = seg[elems] * n_news + cand_idx[elems]
            order = np.argsort(key, kind="stable")
            ks = key[order]
            return elems[order[1:][ks[1:] == ks[:-1]]]

        def elements_of(imps):
            cnt = c[imps]
            start = np.repeat(cand_off[imps] - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
            return start + np.arange(int(cnt.sum()), dtype=np.int64)

        dup = later_occurrences(np.arange(cand_idx.shape[0], dtype=np.int64))
        for _ in range(200):                                       # each round only re-examines the impressions it touched
            if dup.size == 0:
                break
            cand_idx[dup] = draw(dup.size)
            dup = later_occurrences(elements_of(np.unique(seg[dup])))
        bad = np.unique(seg[dup])
        for i in bad:                                              # rare (tiny pools): exact completion
            a, e = int(cand_off[i]), int(cand_off[i + 1])
            first = np.unique(cand_idx[a:e], return_index=True)[1]
            keep = cand_idx[a:e][np.sort(first)]
            rest = np.setdiff1d(np.arange(n_news, dtype=np.int32), keep)
            cand_idx[a:e] = np.concatenate([keep, g.permutation(rest)[: (e - a) - keep.size]])
    labels = (g.random(int(cand_off[-1])) < 0.04).astype(np.float32)
    labels[cand_off[:-1] + (g.random(n_imp) * c).astype(np.int64)] = 1.0
    return {"hist_idx": hist_idx, "hist_off": hist_off, "cand_idx": cand_idx, "cand_off": cand_off,
            "labels": labels}


def synth_impression_blocks(block_ids: Sequence[int], block: int, n_news: int, seed: int = 42) -> Dict[str, np.ndarray]:
    """Concatenation of independent ``block``-impression draws, one per id: block ``b`` is the same whatever the
    other ids are, so a benchmark's step s of rank r sees the same batch for every step count and world size."""
    parts = [synth_impressions(block, n_news, seed=seed + 7919 * (int(b) + 1)) for b in block_ids]
    out = {k: np.concatenate([p[k] for p in parts]) for k in ("hist_idx", "cand_idx", "labels")}
    for k in ("hist_off", "cand_off"):
        off, base = [np.zeros(1, np.int64)], 0
        for p in parts:
            off.append(p[k][1:] + base)
            base += int(p[k][-1])
        out[k] = np.concatenate(off)
    return out


def segment_ids(offsets: np.ndarray) -> np.ndarray:
    """CSR offsets -> sorted segment ids (reference ``_make_batch_assignees``,
    mind_rec_dataset.py:171-174)."""
    sizes = np.diff(offsets)
    return np.repeat(np.arange(sizes.shape[0], dtype=np.int64), sizes)


def shard_range(n: int, rank: int, world: int):
    """Contiguous [lo, hi) block of ``n`` units for ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
