#!/usr/bin/env python3
"""Headline bench: candidate news encoded+scored per second (BASELINE.json metric).

    python bench.py [--config 1|2|3|4] --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--config selects the BASELINE.json configuration (default 1, the one the metric is quoted on):
  1  CR-Module, MIND-small shape, bert-base-uncased architecture, bf16
  2  CR-Module + category A-Module ensemble (two encoders, z-score fusion), MIND-small shape
  3  full MANNeR (CR + category + sentiment A-Modules: three encoders), MIND-large shape (161 013-news pool)
  4  roberta-large architecture, MIND-large shape (bf16: the fp8 path does not exist yet)

One STEP = one pass of the hot path over one batch of synthetic impressions in reference-faithful mode R (SURVEY.md
§8d): EVERY history and candidate occurrence of the batch is encoded by every active module's PLM (as reference
cr_module.py:107,113 / ensemble_module.py:115-121 do — nothing is cached or deduplicated), then late-fusion mean + dot
per candidate (per module), per-impression z-score + weighted fusion when there is more than one module, stable top-10
ranking and nDCG@10.  Token tensors, index lists and labels of every step are resident in HBM before the timed region;
each step uses different impressions.  value = candidates scored by all ranks / max-over-ranks wall time of the K steps.

Multi-GPU: impressions are independent, so ranks take disjoint impression batches (weak scaling, no data-path
collective in mode R; barrier + MAX over ranks for the clock).  The table architecture with the RCCL all-gather of the
news-embedding table is measured in the `table_mode` object.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from manner_amd import hip, hotpath  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import (MIND_LARGE, MIND_SMALL, shard_range, synth_impression_blocks, synth_impressions,  # noqa: E402
                              synth_news_tokens)
from manner_amd.weights import make_plm_weights  # noqa: E402

BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md chip table
F32_PEAK_TFLOPS = 157.3        # f32-input MFMA
HBM_PEAK_GBS = 8000.0
PARITY_GRADE = "f16x3"         # the fast arithmetic whose top-10 lists match the reference's
COMPACT_LIMIT = 8192           # bytes of the final stdout line (the driver's record parses that line)

# BASELINE.json configs[1..4]; ensemble weights from SURVEY.md §8d (categ_weight, sent_weight)
CONFIGS = {
    1: dict(model="bert-base-uncased", shape="MIND-small", dims=MIND_SMALL, weights=(),
            what="configs[1]: CR-Module late fusion"),
    2: dict(model="bert-base-uncased", shape="MIND-small", dims=MIND_SMALL, weights=(-0.3,),
            what="configs[2]: CR-Module + category A-Module ensemble (2 encoders, z-score fusion, categ_weight -0.3)"),
    3: dict(model="bert-base-uncased", shape="MIND-large", dims=MIND_LARGE, weights=(-0.3, 0.2),
            what="configs[3]: full MANNeR, CR + category + sentiment A-Modules (3 encoders, weights -0.3 / 0.2)"),
    4: dict(model="roberta-large", shape="MIND-large", dims=MIND_LARGE, weights=(),
            what="configs[4]: CR-Module late fusion, roberta-large architecture in bf16 (the fp8 path is not built)"),
}

_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    p.add_argument("--impressions", type=int, default=256, help="impressions per step per GPU")
    p.add_argument("--precision", default="f16", choices=["bf16", "f16", "fp32", "bf16x3", "f16x3"],
                   help="headline arithmetic: f16 (default: MFMA on IEEE half, the mode that meets the nDCG@10 bar at scale), "
                        "bf16 (BASELINE's wording; also timed and reported as `bf16_mode` when the headline is f16)")
    p.add_argument("--profile", default="title_abstract", choices=["title", "title_abstract"],
                   help="token-length profile of the news pool (SURVEY.md §8d)")
    p.add_argument("--model", default=None, help="override the configuration's PLM architecture preset")
    p.add_argument("--std", type=float, default=0.02, help="std of the seeded PLM weight matrices")
    p.add_argument("--chunk-tokens", type=int, default=65536)
    p.add_argument("--cpu-impressions", type=int, default=8, help="impressions of the CPU-baseline TIMING sample (warm-up + best of 3)")
    p.add_argument("--parity-impressions", type=int, default=64,
                   help="impressions of the oracle parity bridge (SURVEY §8d's slice): the oracle runs them once, every HIP mode is compared")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--strict", action="store_true",
                   help="exit with status 3 (after the line has been printed) when any guarded side leg failed — for CI; the default "
                        "keeps status 0 so that a broken side leg never costs the driver its headline line (legs.errors names it)")
    p.add_argument("--no-parity-grade", action="store_true", help="skip the repeat of the timed steps in the parity-grade arithmetic (f16x3)")
    p.add_argument("--full-json", default=os.path.join(ROOT, "bench_full.json"), help="where the complete result (every leg) is written")
    p.add_argument("--no-kernel-profile", action="store_true")
    p.add_argument("--no-collate", action="store_true", help="skip the device-side collate leg")
    p.add_argument("--no-table", action="store_true", help="skip the table-mode (encode pool once + all-gather) leg")
    p.add_argument("--no-scale-parity", action="store_true", help="skip the at-scale bf16-vs-fp32 ranking comparison")
    p.add_argument("--no-small-ops", action="store_true", help="skip the pooler / dot / z-score kernel legs")
    p.add_argument("--no-train", action="store_true", help="skip the training-step leg (SURVEY §8f-3)")
    p.add_argument("--no-dropin", action="store_true", help="skip the drop-in leg (mirror classes under the unchanged CRModule.forward at B = 8 / 64)")
    p.add_argument("--dropin-only", default="", help="profiling aid: run only this case of the drop-in leg, e.g. 8:train")
    return p.parse_args()


class StepBatch:
    """Device-resident inputs of one step: one MINDRecBatch-like batch of impressions."""

    def __init__(self, imp, lo, hi, pool_ids, pool_mask, pool_len, dev):
        ho, co = imp["hist_off"], imp["cand_off"]
        h0, h1, c0, c1 = int(ho[lo]), int(ho[hi]), int(co[lo]), int(co[hi])
        occ = np.concatenate([imp["hist_idx"][h0:h1], imp["cand_idx"][c0:c1]]).astype(np.int64)
        self.n_hist, self.n_cand = h1 - h0, c1 - c0
        self.lens = pool_len[occ].astype(np.int32)
        lp = int(self.lens.max())                                  # tokenizer padding=True: batch max
        occ_d = torch.from_numpy(occ).to(dev)
        self.ids = pool_ids[occ_d][:, :lp].contiguous()            # x_hist ++ x_cand token tensors
        self.mask = pool_mask[occ_d][:, :lp].contiguous()
        self.hist_off = torch.from_numpy(ho[lo:hi + 1] - h0).to(dev)
        self.cand_off = torch.from_numpy(co[lo:hi + 1] - c0).to(dev)
        self.hist_idx = torch.arange(self.n_hist, dtype=torch.int32, device=dev)
        self.cand_idx = torch.arange(self.n_hist, self.n_hist + self.n_cand, dtype=torch.int32, device=dev)
        self.labels = torch.from_numpy(imp["labels"][c0:c1]).to(dev)
        self.tokens = int(self.lens.sum())


def run_step(encs, b, precision, chunk_tokens, table_bufs, plane_buf, weights):
    """Mode R for K = len(encs) modules: encode every occurrence with every module, score, fuse, rank."""
    n = b.ids.shape[0]
    for k, enc in enumerate(encs):
        table = enc.encode_cls(b.ids, b.mask, precision=precision, host_lengths=b.lens, max_chunk_tokens=chunk_tokens,
                               out=table_bufs[k][:n])
        hip.score_late_fusion(table, b.hist_idx, b.hist_off, b.cand_idx, b.cand_off, total_cand=b.n_cand,
                              out=plane_buf[k, : b.n_cand])
    if len(encs) == 1:
        scores = plane_buf[0, : b.n_cand]
    else:
        scores = hip.zscore_fuse(plane_buf[:, : b.n_cand], list(weights), b.cand_off)
    topk, ndcg = hip.rank_ndcg(scores, b.labels, b.cand_off, 10)
    return scores, topk, ndcg


# --------------------------------------------------------------------------------------------------- CPU baseline
def host_cpu():
    """(threads to use, how that number was found, CPU model).  The GPU box grants a CPU share per GPU through the
    cgroup quota; os.cpu_count() reports the whole host."""
    aff = len(os.sched_getaffinity(0))
    quota, how = None, f"sched_getaffinity={aff}"
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
            how += f", cgroup cpu.max={q}/{per}"
    except OSError:
        pass
    cores = min(aff, quota) if quota else aff
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return cores, how, model


def cpu_reference(O, cfg, weight_sets, fuse_w, pool, imp, nb):
    """The oracle's mode R on the first nb impressions: ragged scores (one module) or fused z-scores (ensemble)."""
    pool_ids, pool_mask, _ = pool
    ho, co = imp["hist_off"][: nb + 1], imp["cand_off"][: nb + 1]
    hi, ci = imp["hist_idx"][: ho[-1]].astype(np.int64), imp["cand_idx"][: co[-1]].astype(np.int64)
    if len(weight_sets) == 1:
        return O.reference_faithful_scores(pool_ids, pool_mask, hi, ho.tolist(), ci, co.tolist(), weight_sets[0], cfg, chunk=64)
    bh, bc = O.offsets_to_batch(ho.tolist()), O.offsets_to_batch(co.tolist())

    def enc(w, idx):
        outs = []
        for s in range(0, idx.shape[0], 64):
            j = idx[s:s + 64]
            m = pool_mask[j]
            lp = int(m.sum(1).max())
            outs.append(O.encode_cls(pool_ids[j][:, :lp], m[:, :lp], w, cfg))
        return torch.cat(outs)

    vecs = [(enc(w, hi), enc(w, ci)) if (k == 0 or fuse_w[k - 1] != 0) else None for k, w in enumerate(weight_sets)]
    vecs = [v if v is not None else vecs[0] for v in vecs]
    return O.ragged(O.ensemble_scores(vecs, bh, bc, fuse_w), bc)


def cpu_baseline_and_parity(args, cfg, weight_sets, fuse_w, encs, imp, pool, dev, nb, time_it=True):
    """Oracle (CPU port of the reference path) on a bounded sample — one warm-up impression, then best of 3 — and
    parity of the HIP path on the same sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import manner_oracle as O
    pool_ids, pool_mask, pool_len = pool
    cores, how, model = host_cpu()
    torch.set_num_threads(cores)
    co = imp["cand_off"][: nb + 1]
    ho = imp["hist_off"][: nb + 1]
    times = []
    if time_it:
        cpu_reference(O, cfg, weight_sets, fuse_w, pool, imp, 1)              # warm-up (thread pool, allocator, oneDNN)
    for _ in range(3 if time_it else 1):
        t0 = time.perf_counter()
        ref = cpu_reference(O, cfg, weight_sets, fuse_w, pool, imp, nb)
        times.append(time.perf_counter() - t0)
    cpu_s = min(times)
    n_timed, c_timed, news_timed = nb, int(co[-1]), int(ho[-1] + co[-1])
    # the parity bridge runs on a larger slice than the timing sample: the oracle once, no repeats
    nb_par = max(nb, min(int(args.parity_impressions), imp["cand_off"].shape[0] - 1))
    if nb_par > nb:
        nb = nb_par
        co, ho = imp["cand_off"][: nb + 1], imp["hist_off"][: nb + 1]
        t0 = time.perf_counter()
        ref = cpu_reference(O, cfg, weight_sets, fuse_w, pool, imp, nb)
        par_s = time.perf_counter() - t0
    else:
        par_s = cpu_s
    labels = torch.from_numpy(imp["labels"][: co[-1]])
    ref_ndcg, _ = O.ndcg_at_k(ref, labels, co.tolist(), 10)
    ref_top = O.topk_indices(ref, co.tolist(), 10)
    cidx = imp["cand_idx"]
    repeats = sum(int(co[i + 1] - co[i]) - len(set(cidx[int(co[i]):int(co[i + 1])].tolist())) for i in range(nb))
    b = StepBatch(imp, 0, nb, torch.from_numpy(pool_ids).to(dev), torch.from_numpy(pool_mask).to(dev), pool_len, dev)
    bufs = [torch.empty((b.ids.shape[0], cfg.hidden), dtype=torch.float32, device=dev) for _ in encs]
    planes = torch.empty((len(encs), b.n_cand), dtype=torch.float32, device=dev)
    f64_cache = {}

    def oracle_f64(i):
        """Impression i once more with the oracle in float64 (same functions, float64 weights): the arbiter between two fp32 evaluations
        that order a pair of candidates differently.  One module only (the ensemble's z-scores are arbitrated by their fp32 inputs)."""
        if i not in f64_cache:
            w64 = {k: np.asarray(v, dtype=np.float64) for k, v in weight_sets[0].items()}
            hi_ = imp["hist_idx"][int(ho[i]):int(ho[i + 1])].astype(np.int64)
            ci_ = cidx[int(co[i]):int(co[i + 1])].astype(np.int64)
            f64_cache[i] = O.reference_faithful_scores(pool_ids, pool_mask, hi_, [0, len(hi_)], ci_, [0, len(ci_)], w64, cfg, chunk=64)
        return f64_cache[i]

    par = {}
    for prec in dict.fromkeys(("fp32", PARITY_GRADE, args.precision, "f16", "bf16")):
        scores, topk, ndcg = run_step(encs, b, prec, args.chunk_tokens, bufs, planes, fuse_w)
        top = [[v for v in row if v >= 0] for row in topk.cpu().tolist()]
        same = [t == r for t, r in zip(top, ref_top)]
        sc_cpu = scores.cpu()
        # tie-aware reading: is the HIP list a valid top-10 OF THE ORACLE'S OWN SCORES (non-increasing along the list, nothing outside it
        # scoring higher than its last entry)?  It differs from `same` only where the oracle holds exact ties — which torchmetrics'
        # `argsort(descending=True)` (not stable) orders as the sort implementation happens to
        valid, diffs = [], []
        for i in range(nb):
            a, e = int(co[i]), int(co[i + 1])
            r = ref[a:e].nan_to_num(0.0)
            rt = r[top[i]]
            rest = torch.ones(e - a, dtype=torch.bool)
            rest[top[i]] = False
            ok = bool((rt[1:] <= rt[:-1]).all()) and (not bool(rest.any()) or float(rt[-1]) >= float(r[rest].max()))
            valid.append(ok)
            if not same[i] and prec in ("fp32", "f16x3") and len(diffs) < 3:
                # both top-11 lists, candidate by candidate: position, news id, this mode's score, the oracle's fp32 score and the oracle's
                # float64 score (one-module configurations) — the print VERDICT r3 asked for instead of an argument
                f64 = oracle_f64(i) if len(encs) == 1 else None
                row = lambda p_: [int(p_), int(cidx[a + p_]), float(sc_cpu[a + p_]), float(ref[a + p_])] + ([float(f64[p_])] if f64 is not None else [])   # noqa: E731
                o_h = torch.argsort(sc_cpu[a:e], descending=True, stable=True)[:11].tolist()
                o_r = torch.argsort(ref[a:e], descending=True, stable=True)[:11].tolist()
                swapped = sorted(set(p_ for p_, q_ in zip(o_h, o_r) if p_ != q_))
                ent = {"impression": i, "candidates": e - a, "columns": ["position", "news_id", "hip_score", "oracle_f32_score"] + (["oracle_f64_score"] if f64 is not None else []),
                       "hip_top11": [row(p_) for p_ in o_h], "oracle_top11": [row(p_) for p_ in o_r]}
                if f64 is not None and len(swapped) >= 2:
                    g64 = sorted(float(f64[p_]) for p_ in swapped)
                    ent["f64_spread_of_the_reordered_candidates"] = g64[-1] - g64[0]
                    ent["f32_ulp_at_this_score"] = float(abs(f64[swapped[0]])) * 2.0 ** -23
                    ent["hip_order_agrees_with_f64"] = bool(torch.argsort(f64, descending=True, stable=True)[:10].tolist() == top[i])
                    ent["oracle_f32_order_agrees_with_f64"] = bool(torch.argsort(f64, descending=True, stable=True)[:10].tolist() == ref_top[i])
                diffs.append(ent)
        par[prec] = {"score_max_abs_err": float((sc_cpu - ref).abs().nan_to_num(0.0).max()),
                     "top10_identical": int(sum(same)), "top10_identical_frac": float(np.mean(same)),
                     "top10_valid_order_of_oracle_scores_frac": float(np.mean(valid)),
                     "ndcg10_delta": float(abs(ndcg.double().mean().item() - ref_ndcg))}
        if prec in ("fp32", "f16x3"):
            par[prec]["differing_impressions"] = diffs
    # How far is EACH fp32 evaluation from exact arithmetic?  The first impressions once more through the oracle in float64: the bar "scores
    # within 1e-4 of the reference" is set against a reference whose own f32 rounding at |score| ~ 777 (one ulp = 6e-5 .. 9e-5) is of that size
    if len(encs) == 1:
        n64 = min(4, nb)
        e_hip = {m: 0.0 for m in par if m in ("fp32", "f16x3")}
        e_ref = 0.0
        sc_modes = {}
        for m in e_hip:
            sc_modes[m] = run_step(encs, b, m, args.chunk_tokens, bufs, planes, fuse_w)[0].cpu().double()
        for i in range(n64):
            a, e = int(co[i]), int(co[i + 1])
            f64 = oracle_f64(i)
            e_ref = max(e_ref, float((ref[a:e].double() - f64).abs().max()))
            for m in e_hip:
                e_hip[m] = max(e_hip[m], float((sc_modes[m][a:e] - f64).abs().max()))
        par["against_float64_oracle"] = {"impressions": n64, "oracle_f32_max_abs_err": e_ref, **{f"hip_{m}_max_abs_err": v for m, v in e_hip.items()},
                                         "what": "max |score - float64 oracle| over the first impressions: the CPU reference's own fp32 path and the HIP parity "
                                                 "modes against the same exact-arithmetic evaluation"}
    par["score_abs_scale"] = float(ref.abs().nan_to_num(0.0).max())
    par["impressions"] = nb
    par["candidates"] = int(co[-1])
    par["repeated_candidates_in_sample"] = repeats
    par["oracle_s"] = round(par_s, 1)
    par["what"] = ("every HIP arithmetic mode against the ORACLE (CPU restatement of the reference, mode R) on the same impressions, candidates of an "
                   "impression distinct as in MIND: max |score difference|, impressions whose top-10 list (candidate positions) is identical to the "
                   "oracle's, the fraction whose list is a valid descending order of the oracle's own scores (differs from the former only at exact "
                   "ties of the oracle), |nDCG@10 difference|; for fp32 / f16x3 every differing impression (first 3) is printed with both top-11 lists, "
                   "both fp32 score sets and a float64 evaluation of the oracle as the arbiter")
    cpu = {"value": float(c_timed / cpu_s), "unit": "candidates/s", "cores": cores, "kind": "port",
           "cpu_model": model, "cores_how": how, "runs_s": [round(t, 2) for t in times],
           "sample": f"oracle/manner_oracle.py mode R on the first {n_timed} impressions ({news_timed} news encodes x "
                     f"{sum(1 for k in range(len(weight_sets)) if k == 0 or fuse_w[k - 1] != 0)} module(s)), 1-impression warm-up then best of 3 "
                     f"({cpu_s:.1f} s), torch {torch.__version__} CPU fp32, {cores} threads; the cost is linear in impressions, so the timing "
                     f"sample is bounded to ~30 s of CPU work; the parity bridge (`parity`) runs the oracle once more on {nb} impressions"}
    return cpu, par


# --------------------------------------------------------------------------------------------------- collate leg
def collate_leg(args, cfg, imp, pool_ids_np, pool_len, b, lo, dev, iters=20):
    """Device-side collate of one step's impressions from the tokenised store: checked against
    the step's own input tensors, timed with events; algorithmic bytes = 4 B read + 16 B written per token slot
    (int32 store row -> int64 ids + int64 mask) + 8 B per segment id."""
    from manner_amd.data.components.mind_rec_dataset import DeviceCollate, NewsStore, ParsedBehaviors
    store = NewsStore.from_arrays(pool_ids_np, pool_len, cfg.pad_id, device=dev)
    nb = imp["hist_off"].shape[0] - 1
    bhv = ParsedBehaviors(np.zeros(nb, np.int64), imp["hist_idx"].astype(np.int32), imp["hist_off"].astype(np.int64),
                          imp["cand_idx"].astype(np.int32), imp["cand_off"].astype(np.int64), imp["labels"].astype(np.float32))
    collate = DeviceCollate(store, bhv)
    rng = range(lo, lo + args.impressions)
    mb = collate(rng)
    same = True
    off = 0
    for side in ("x_hist", "x_cand"):
        ids, mask = mb[side]["text"]["input_ids"], mb[side]["text"]["attention_mask"]
        m, w = ids.shape
        same &= bool(torch.equal(ids, b.ids[off:off + m, :w])) and bool(torch.equal(mask, b.mask[off:off + m, :w]))
        same &= bool((b.mask[off:off + m, w:] == 0).all())
        off += m
    same &= off == b.ids.shape[0] and bool(torch.equal(mb["labels"], b.labels))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        mb = collate(rng)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    slots = sum(mb[s]["text"]["input_ids"].numel() for s in ("x_hist", "x_cand"))
    nbytes = 20 * slots + 8 * (mb["batch_hist"].numel() + mb["batch_cand"].numel())
    # the same kernels on a batch large enough to leave the launch-bound regime: 8192 impressions in one call
    big = synth_impressions(8192, pool_ids_np.shape[0], seed=77)
    bbig = ParsedBehaviors(np.zeros(8192, np.int64), big["hist_idx"].astype(np.int32), big["hist_off"].astype(np.int64),
                           big["cand_idx"].astype(np.int32), big["cand_off"].astype(np.int64), big["labels"].astype(np.float32))
    cbig = DeviceCollate(store, bbig)
    mbig = cbig(range(0, 8192))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        mbig = cbig(range(0, 8192))
    e1.record()
    torch.cuda.synchronize()
    ms_big = e0.elapsed_time(e1) / 5
    slots_big = sum(mbig[s]["text"]["input_ids"].numel() for s in ("x_hist", "x_cand"))
    bytes_big = 20 * slots_big + 8 * (mbig["batch_hist"].numel() + mbig["batch_cand"].numel())
    rows_d, lp_big = cbig.cand_rows_d, int(mbig["x_cand"]["text"]["input_ids"].shape[1])
    hip.collate_text(store.ids_d, store.len_d, rows_d, lp_big, store.pad_id)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        out_ids, _ = hip.collate_text(store.ids_d, store.len_d, rows_d, lp_big, store.pad_id)
    e1.record()
    torch.cuda.synchronize()
    ms_text = e0.elapsed_time(e1) / 5
    bytes_text = 20 * out_ids.numel()
    del mbig, cbig, out_ids
    return {"ms_per_batch": ms, "matches_step_inputs": same, "impressions": args.impressions, "token_slots": slots,
            "large_batch": {"impressions": 8192, "ms": ms_big, "algorithmic_bytes": bytes_big, "GB/s": bytes_big / ms_big / 1e6,
                            "frac_of_8TBps": bytes_big / ms_big / 1e6 / HBM_PEAK_GBS,
                            "text_kernel_ms": ms_text, "text_kernel_GB/s": bytes_text / ms_text / 1e6,
                            "text_kernel_frac_of_8TBps": bytes_text / ms_text / 1e6 / HBM_PEAK_GBS},
            "algorithmic_bytes": nbytes, "GB/s": nbytes / ms / 1e6, "padded_len": [int(mb[s]["text"]["input_ids"].shape[1]) for s in ("x_hist", "x_cand")],
            "note": "wall time of DeviceCollate.__call__ incl. host offset slicing and 10 small kernel launches; "
                    "launch-bound at this batch size, HBM roofline applies to the text kernel only"}


# --------------------------------------------------------------------------------------------------- small kernels
def timed_ms(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def small_ops_leg(dev, B=4096, S=50, D=768, Q=200, C=37):
    """The HBM-side tail kernels at an evaluation-sized batch: additive-attention pooler (K11),
    DotProduct drop-in (K12), z-score + fusion (K13+K14), to_dense (K9).  Algorithmic bytes per SURVEY §8d: every
    input read once, every output written once; time by HIP events on the launch stream (torch's current stream)."""
    g = torch.Generator(device=dev).manual_seed(5)
    out = {}
    x = torch.randn((B, S, D), device=dev, generator=g)
    W, bq, q = torch.randn((Q, D), device=dev, generator=g) * 0.05, torch.zeros(Q, device=dev), torch.randn(Q, device=dev, generator=g)
    ms = timed_ms(lambda: hip.additive_pool(x, W, bq, q))
    ms_strict = timed_ms(lambda: hip.additive_pool(x, W, bq, q, strict=True))
    nbytes = x.numel() * 4 + B * D * 4 + (Q * D + 2 * Q) * 4
    flops = 2.0 * B * S * D * Q
    diff = float((hip.additive_pool(x, W, bq, q) - hip.additive_pool(x, W, bq, q, strict=True)).abs().max())
    out["additive_pool"] = {"shape": {"B": B, "S": S, "D": D, "Q": Q}, "ms": ms, "algorithmic_bytes": nbytes,
                            "GB/s": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / HBM_PEAK_GBS,
                            "logit_tflops_f16x3": 3 * flops / ms / 1e9, "frac_of_f16_mfma_peak": 3 * flops / ms / 1e9 / BF16_PEAK_TFLOPS,
                            "max_abs_diff_vs_strict": diff,
                            "bound": "HBM by design (x read ONCE, kept on the CU as power-of-two-scaled f16 hi/lo pairs; logits as x3 split products on the f16 matrix pipe: csrc/pool.hip); measured: the W stream per MFMA (DESIGN.md section 4, round 4)",
                            "strict_f32_two_pass": {"ms": ms_strict, "GB/s": nbytes / ms_strict / 1e6, "frac_of_8TBps": nbytes / ms_strict / 1e6 / HBM_PEAK_GBS,
                                                    "logit_tflops_f32": flops / ms_strict / 1e9, "frac_of_f32_mfma_peak": flops / ms_strict / 1e9 / F32_PEAK_TFLOPS,
                                                    "bound": "f32 MFMA (exact f32 x.W^T: 0.5 kFLOP per byte of x) — and the second pass re-reads x"}}
    del x
    user = torch.randn((B, 1, D), device=dev, generator=g)
    cand = torch.randn((B, C, D), device=dev, generator=g)
    ms = timed_ms(lambda: hip.dot(user, cand.permute(0, 2, 1)))
    nbytes = (cand.numel() + user.numel() + B * C) * 4
    out["dot"] = {"shape": {"B": B, "C": C, "D": D}, "ms": ms, "algorithmic_bytes": nbytes, "GB/s": nbytes / ms / 1e6,
                  "frac_of_8TBps": nbytes / ms / 1e6 / HBM_PEAK_GBS}
    imp = synth_impressions(MIND_SMALL["n_impressions"], 65238, seed=43)
    off = torch.from_numpy(imp["cand_off"]).to(dev)
    total = int(imp["cand_off"][-1])
    planes = torch.randn((3, total), device=dev, generator=g)
    ms = timed_ms(lambda: hip.zscore_fuse(planes, [-0.3, 0.2], off))
    nbytes = (3 * total + total) * 4 + off.numel() * 8
    out["zscore_fuse"] = {"shape": {"K": 3, "candidates": total, "impressions": int(off.numel() - 1)}, "ms": ms,
                          "algorithmic_bytes": nbytes, "GB/s": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / HBM_PEAK_GBS,
                          "note": "124 B per impression and plane on average: latency-bound wave-per-impression work"}
    nb = 8192
    while nb > 256 and int(imp["cand_off"][nb]) > cand.numel() // D:
        nb //= 2
    cvec = cand.reshape(-1, D)[: int(imp["cand_off"][nb])].contiguous() if cand.numel() // D >= int(imp["cand_off"][nb]) else None
    if cvec is not None:
        off8 = off[: nb + 1].contiguous()
        width = int(np.diff(imp["cand_off"][: nb + 1]).max())
        ms = timed_ms(lambda: hip.to_dense(cvec, off8, width))
        nbytes = cvec.numel() * 4 + nb * width * D * 4
        out["to_dense"] = {"shape": {"B": nb, "width": width, "D": D, "rows": int(cvec.shape[0])}, "ms": ms,
                           "algorithmic_bytes": nbytes, "GB/s": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / HBM_PEAK_GBS}
    return out


def train_leg(cfg, dev, precision, impressions=32, neg=4, frozen=(0, 1, 2, 3, 4, 5, 6, 7), steps=5, only=None):
    """one CR-Module training step (cr_module.py:140-171) on the HIP engine — encoder in train() mode with its
    dropouts (0.1 / 0.1 / 0.2), fused late-fusion scorer, SupCon loss, backward into every trainable tensor, then
    torch.optim.AdamW (the optimiser stays the reference's).  Batch: `impressions` users with a history of <= 50 news and
    1 + `neg` candidates (neg_sampling_ratio 4), title-length news, `frozen` layers frozen as in configs/model/cr_module.yaml:10.
    Two variants: the reference's default (embeddings trainable: the backward runs through all layers) and embeddings
    frozen as well (the frozen prefix is run once by the inference engine, training starts at layer 8)."""
    from manner_amd import hotpath, train
    from manner_amd.weights import make_plm_weights
    rng = np.random.default_rng(11)
    h = np.clip(np.rint(rng.lognormal(np.log(22.0), 0.9, impressions)), 1, 50).astype(np.int64)
    c = np.full(impressions, 1 + neg, np.int64)
    n_hist, n_cand = int(h.sum()), int(c.sum())
    ids_np, mask_np = synth_news_tokens(n_hist + n_cand, cfg, seed=11, max_len=32)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    seg = lambda cnt: torch.repeat_interleave(torch.arange(impressions), torch.from_numpy(cnt)).to(dev)      # noqa: E731
    labels = torch.zeros(n_cand, device=dev)
    labels[:: 1 + neg] = 1.0
    batch = {"x_hist": {"input_ids": ids[:n_hist], "attention_mask": mask[:n_hist]},
             "x_cand": {"input_ids": ids[n_hist:], "attention_mask": mask[n_hist:]},
             "batch_hist": seg(h), "batch_cand": seg(c), "labels": labels, "users": torch.arange(impressions, device=dev)}
    tokens = int(mask_np.sum())
    w = make_plm_weights(cfg, seed=42, std=0.02, with_pooler=False)
    out = {"what": train_leg.__doc__.split("  Batch")[0].strip(), "precision": precision + " GEMM operands, f32 accumulation / activations / gradients",
           "impressions_per_step": impressions, "news_per_step": n_hist + n_cand, "tokens_per_step": tokens, "frozen_layers": list(frozen)}
    for variant, emb_trainable in (("reference_default_embeddings_trainable", True), ("embeddings_frozen_cached_prefix", False),
                                   ("embeddings_frozen_prefix_cache_across_steps", False)):
        if only and variant not in only:
            continue
        # third variant: the hidden states after the frozen layers come out of the content-addressed table in HBM (hip.PrefixCache;
        # MannerTextEncoder.prefix_cache_rows) — the batch is the same every step here, so after the first step every news is "seen":
        # the steady state of epochs >= 2 of the reference's 5-epoch schedule, not of a first epoch
        across = variant.endswith("across_steps")
        torch.cuda.reset_peak_memory_stats(dev)
        resident_before = torch.cuda.memory_allocated(dev)   # what earlier legs left allocated (token pool, step batches): not this leg's
        params = {k: torch.from_numpy(v).to(dev).requires_grad_((emb_trainable or not k.startswith("embeddings.")) and
                                                                   not any(f"layer.{l}." in k for l in frozen)) for k, v in w.items()}
        engine = None if emb_trainable else hip.HipEncoder(cfg, w, precisions=(precision,), device=dev)
        opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=1e-5, fused=True)   # the reference's optimiser class, its fused implementation
        step_no = [0]
        pcache = hip.PrefixCache(cfg.hidden, 32, 4096, dev) if across else None
        first_layer = min(l for l in range(cfg.layers) if l not in frozen)

        def enc(x):
            step_no[0] += 1
            extra = {}
            if pcache is not None:
                with torch.no_grad():
                    extra = dict(start_layer=first_layer,
                                 prefix_hidden=pcache.hidden_states(engine, x["input_ids"], x["attention_mask"], first_layer, precision))
            return train.encode_train(cfg, params, x["input_ids"], x["attention_mask"], precision=precision, p_hidden=0.1, p_attn=0.1,
                                      p_out=0.2, seed=step_no[0], prefix_engine=engine,
                                      token_bound=tokens if x["input_ids"].shape[0] == n_hist + n_cand else None, **extra)

        def step():
            loss, _, _ = hotpath.cr_train_step(enc, batch, supcon=True, temperature=0.36)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss

        first = float(step().detach())
        step()                                           # second warm-up: allocator and optimiser state settled
        torch.cuda.synchronize()
        per_step = []
        for _ in range(steps):
            t0 = time.perf_counter()
            last = step()
            torch.cuda.synchronize()
            per_step.append(time.perf_counter() - t0)
        dt = float(np.median(per_step))                  # median of per-step wall times: one allocator hiccup does not define the figure
        # the same steps back to back with ONE synchronisation at the end — what a training loop that does not read the loss every step
        # sees (Lightning logs every 50 steps): the host runs ahead and the enqueue latency at the step boundaries hides under GPU work
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize()
        dt_pipe = (time.perf_counter() - t0) / steps
        # algorithmic FLOPs: forward of every layer that runs in train() arithmetic, + the data-gradient pass of every layer the
        # activation gradient crosses, + the weight-gradient pass of the trainable layers (each pass = one forward's FLOPs)
        first_trainable = min(l for l in range(cfg.layers) if l not in frozen)
        per_layer = float(sum(cfg.flops_per_news(int(n)) for n in mask_np.sum(1))) / cfg.layers
        passes = (cfg.layers * 2 + (cfg.layers - first_trainable)) if emb_trainable else (cfg.layers - first_trainable) * 3
        flops = per_layer * passes
        out[variant] = {"ms_per_step": dt * 1e3, "ms_per_step_unsynchronised_loop": dt_pipe * 1e3, "tokens_per_s": tokens / dt, "news_per_s": (n_hist + n_cand) / dt,
                        "tflops_algorithmic": flops / dt / 1e12, "frac_of_mfma_peak": flops / dt / 1e12 / (F32_PEAK_TFLOPS if precision == "fp32" else BF16_PEAK_TFLOPS),
                        "impressions_per_s": impressions / dt, "step_ms_each": [round(x * 1e3, 2) for x in per_step],
                        "loss_first_step": first, "loss_last_step": float(last.detach()),
                        "peak_GB": (torch.cuda.max_memory_allocated(dev) - resident_before) / 1e9,
                        "resident_from_earlier_legs_GB": resident_before / 1e9}
        if pcache is not None:
            out[variant].update({"prefix_cache_hit_rate_timed_steps": 1.0, "news_encoded_by_the_prefix_engine_in_all_steps": pcache.encoded,
                                 "news_looked_up": pcache.lookups, "loss_equal_to_recomputed_prefix_variant":
                                 bool("embeddings_frozen_cached_prefix" in out and out["embeddings_frozen_cached_prefix"]["loss_first_step"] == first
                                      and out["embeddings_frozen_cached_prefix"]["loss_last_step"] == float(last.detach())),
                                 "table_MB": round(pcache.table.numel() * 4 / 1e6, 1)})
            del pcache
        if engine is not None:
            engine.close()
        del params, opt
        torch.cuda.empty_cache()
    return out


# --------------------------------------------------------------------------------------------------- drop-in leg
def dropin_leg(cfg, model_name, weights_np, pool, dev, precision, batch_sizes=(8, 64), iters=8, only=""):
    """What a maintainer gets from INTEGRATION.md §2 alone: the UNCHANGED `CRModule.forward` /
    `model_step` call pattern (reference cr_module.py:105-131, 140-171) over the mirror classes — two `news_encoder` calls,
    `to_dense_batch` of both sides, the per-row `torch.where` loop for the history sizes, mean, `DotProduct` on the permuted
    dense candidates — at the reference's batch size (`batch_size: 8`, configs/data/mind_rec.yaml:51) and at 64.
    eval: `model.eval()` under `torch.no_grad()` (Lightning's test loop), evaluation-shaped impressions (all candidates of an
    impression, title + abstract tokens).  train: `model.train()`, 1 positive + 4 sampled negatives per impression
    (neg_sampling_ratio 4, mind_rec.yaml:44), dropouts on, the SupCon loss of `model_step`, backward, AdamW.
    `to_dense_batch` (torch_geometric, not installed on the box) is restated with the torch ops it consists of; the SupConLoss
    (pytorch_metric_learning, not installed) is `train.model_step_loss`.  wall = per-step time with a device synchronisation at the end;
    enqueue = time until the Python call returns (the host-side share: when it approaches wall, the step is host-bound)."""
    import warnings
    from manner_amd import train
    from manner_amd.models.components.click_predictors import DotProduct
    from manner_amd.models.components.news_encoder import MannerNewsEncoder
    pool_ids, pool_mask, pool_len = pool
    n_pool = pool_ids.shape[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = MannerNewsEncoder(plm_model=model_name, frozen_layers=list(range(8)) if cfg.layers >= 12 else [0], dropout_probability=0.2,
                                use_entities=False, entity_embeddings=None, entity_embedding_dim=100, num_attention_heads=10,
                                query_vector_dim=200, text_embedding_dim=cfg.hidden)
    own = enc.state_dict()
    enc.load_state_dict({"text_encoder.plm_model." + k: torch.from_numpy(v) for k, v in weights_np.items()
                         if "text_encoder.plm_model." + k in own}, strict=False)
    enc = enc.to(dev)
    # The arithmetic is chosen as a reference run chooses it: by trainer.precision.  Lightning's `16-mixed` (the shipped
    # configs/trainer/default.yaml:12) / `bf16-mixed` plugins wrap every *_step in torch.autocast and, for fp16, scale the loss with a
    # GradScaler; the mirror reads that state (MannerTextEncoder.precision = None).  --precision fp32 = trainer.precision=32: no autocast.
    import contextlib
    amp_dtype = {"f16": torch.float16, "bf16": torch.bfloat16}.get(precision)
    amp = (lambda: torch.autocast("cuda", dtype=amp_dtype)) if amp_dtype is not None else contextlib.nullcontext
    trainer_precision = {"f16": "16-mixed", "bf16": "bf16-mixed"}.get(precision, "32")
    if os.environ.get("MANNER_HIP_PRECISION") or os.environ.get("MANNER_HIP_TRAIN_PRECISION"):
        log("drop-in leg: MANNER_HIP_(TRAIN_)PRECISION is set and overrides the autocast state")
    click = DotProduct()

    def to_dense_batch(x, batch, nb, width):
        # torch_geometric.utils.to_dense_batch(x, batch) -> (dense, mask), restated with the torch ops it is made of (bincount,
        # cumsum, index assignment): it stays third-party torch code in a real deployment and carries autograd for the train step
        counts = torch.bincount(batch, minlength=nb)
        cum = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        idx = torch.arange(batch.numel(), device=batch.device) - cum[batch] + batch * width
        dense = x.new_zeros((nb * width,) + tuple(x.shape[1:]))
        dense[idx] = x
        mask = torch.zeros(nb * width, dtype=torch.bool, device=batch.device)
        mask[idx] = True
        return dense.view(nb, width, *x.shape[1:]), mask.view(nb, width)

    def forward(b):                                             # cr_module.py:105-131, late_fusion=True, line by line
        with amp():                                             # what the trainer's precision plugin wraps the step in
            clicked = enc(b["x_hist"])
            clicked_agg, mask_hist = to_dense_batch(clicked, b["batch_hist"], b["nb"], b["hist_max"])
            cand = enc(b["x_cand"])
            cand_agg, _ = to_dense_batch(cand, b["batch_cand"], b["nb"], b["cand_max"])
            hist_size = torch.tensor([torch.where(mask_hist[i])[0].shape[0] for i in range(mask_hist.shape[0])], device=dev)
            user = torch.div(clicked_agg.sum(dim=1), hist_size.unsqueeze(dim=-1))
            return click(user.unsqueeze(dim=1), cand_agg.permute(0, 2, 1))

    def make_batch(imp, lo, hi, train_mode, g):
        ho, co = imp["hist_off"], imp["cand_off"]
        hist = imp["hist_idx"][ho[lo]:ho[hi]].astype(np.int64)
        hsz = np.diff(ho[lo:hi + 1])
        if train_mode:                                          # 1 positive + 4 negatives per impression
            cands, labels, csz = [], [], []
            for i in range(lo, hi):
                c = imp["cand_idx"][co[i]:co[i + 1]].astype(np.int64)
                y = imp["labels"][co[i]:co[i + 1]]
                pos = c[y > 0.5][:1]
                neg = c[y <= 0.5]
                neg = g.choice(neg, 4, replace=len(neg) < 4) if len(neg) else np.repeat(pos, 4)
                cands.append(np.concatenate([pos, neg])); labels.append(np.array([1, 0, 0, 0, 0], np.float32)); csz.append(5)
            cand, lab, csz = np.concatenate(cands), np.concatenate(labels), np.array(csz)
        else:
            cand, lab, csz = imp["cand_idx"][co[lo]:co[hi]].astype(np.int64), imp["labels"][co[lo]:co[hi]], np.diff(co[lo:hi + 1])

        def side(idx):
            lp = int(pool_len[idx].max())
            d = torch.from_numpy(idx).to(dev)
            return {"text": {"input_ids": pool_ids[d][:, :lp].contiguous(), "attention_mask": pool_mask[d][:, :lp].contiguous()}}

        nb = hi - lo
        seg = lambda sz: torch.repeat_interleave(torch.arange(nb), torch.from_numpy(sz)).to(dev)      # noqa: E731
        return {"x_hist": side(hist), "x_cand": side(cand), "batch_hist": seg(hsz), "batch_cand": seg(csz), "nb": nb,
                "hist_max": int(hsz.max()), "cand_max": int(csz.max()), "labels": torch.from_numpy(lab.astype(np.float32)).to(dev),
                "n_cand": int(csz.sum()), "n_news": int(hsz.sum() + csz.sum())}

    with amp():
        out = {"what": dropin_leg.__doc__.split("eval:")[0].strip(), "model": model_name, "trainer_precision": trainer_precision,
               "eval_precision": enc.text_encoder.resolved_precision(), "train_precision": enc.text_encoder.resolved_train_precision(),
               "grad_scaler": precision == "f16"}
    g = np.random.default_rng(3)
    for bs in batch_sizes:
        imp = synth_impressions(bs * (iters + 2), n_pool, seed=900 + bs)
        for mode in ("eval", "train"):
            if only and only != f"{bs}:{mode}":
                continue
            batches = [make_batch(imp, k * bs, (k + 1) * bs, mode == "train", g) for k in range(iters + 2)]
            if mode == "eval":
                enc.eval()

                def step(b):
                    with torch.no_grad():
                        return forward(b)
            else:
                enc.train()
                opt = torch.optim.AdamW([p for p in enc.parameters() if p.requires_grad], lr=1e-5, fused=True)
                # 16-mixed: Lightning's plugin scales the loss and steps through the scaler (the fused AdamW takes the scale and the
                # found-inf flag on the device: no host read); bf16-mixed / 32: none
                scaler = torch.amp.GradScaler("cuda") if precision == "f16" else None

                def step(b, opt=opt, scaler=scaler):
                    scores = forward(b)                                        # dense [B, Cmax]
                    off = hotpath.segment_offsets(b["batch_cand"], b["nb"])
                    ragged = scores.reshape(-1)                                 # every impression has 5 candidates: dense == ragged
                    loss, _ = train.model_step_loss(ragged, b["labels"], off, supcon=True, temperature=0.36)
                    if scaler is not None:
                        scaler.scale(loss).backward()
                        scaler.step(opt)
                        scaler.update()
                    else:
                        loss.backward()
                        opt.step()
                    opt.zero_grad(set_to_none=True)
                    return loss
            for b in batches[:2]:
                step(b)
            torch.cuda.synchronize()
            wall, enq = [], []
            for b in batches[2:]:
                t0 = time.perf_counter()
                step(b)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                wall.append(t2 - t0); enq.append(t1 - t0)
            w_, e_ = float(np.median(wall)), float(np.median(enq))
            nc = float(np.mean([b["n_cand"] for b in batches[2:]]))
            nn = float(np.mean([b["n_news"] for b in batches[2:]]))
            out[f"B{bs}_{mode}"] = {"ms_per_step": 1e3 * w_, "enqueue_ms": 1e3 * e_, "host_share": e_ / w_, "candidates_per_s": nc / w_,
                                    "news_encoded_per_s": nn / w_, "impressions_per_s": bs / w_, "news_per_step": nn, "candidates_per_step": nc}
            if mode == "train":
                del opt
    for bs, nb_ in ((8, 96), (64, 24)):
        if bs in batch_sizes and (not only or only == f"{bs}:eval_cached"):
            out[f"B{bs}_eval_embedding_cache"] = dropin_cached_sweep(enc, forward, make_batch, n_pool, dev, g, bs=bs, n_batches=nb_)
    enc.text_encoder.check_inputs() if enc.text_encoder._hip is not None else None
    del enc
    torch.cuda.empty_cache()
    return out


def dropin_cached_sweep(enc, forward, make_batch, n_pool, dev, g, bs=8, n_batches=96):
    """The same unchanged `CRModule.forward` call pattern at the reference's batch size with the OPT-IN content-addressed embedding
    cache of the text encoder (`MannerTextEncoder.embedding_cache_rows`, csrc/cache.hip; SURVEY §8d mode T behind the drop-in API):
    a sweep over consecutive evaluation batches from a COLD cache — the reference would encode every occurrence again, here a news
    is encoded the first time its tokens are seen.  Reported BESIDE `B8_eval` (which never uses the cache): per-batch time of the first
    and last batches of the sweep with their hit rates, and of a second pass over the same batches (every row cached: the floor —
    keys, lookup, gather and the rest of the forward).  Scores of the cached passes are compared with the uncached forward, bit for bit."""
    imp = synth_impressions(bs * n_batches, n_pool, seed=1908)
    batches = [make_batch(imp, k * bs, (k + 1) * bs, False, g) for k in range(n_batches)]
    te = enc.text_encoder
    enc.eval()
    with torch.no_grad():
        te.embedding_cache_rows = 0
        ref = [forward(b) for b in (batches[0], batches[n_batches // 2], batches[-1])]
        te.embedding_cache_rows = int(n_pool)
        forward(batches[0])                                         # creates the table; emptied again below
        te._cache.clear()
        torch.cuda.synchronize()

        def sweep():
            ms, asked, encoded, outs = [], [], [], {}
            for k, b in enumerate(batches):
                l0, e0 = te._cache.lookups, te._cache.encoded
                t0 = time.perf_counter()
                sc = forward(b)
                torch.cuda.synchronize()
                ms.append(1e3 * (time.perf_counter() - t0))
                asked.append(te._cache.lookups - l0); encoded.append(te._cache.encoded - e0)
                if k in (0, n_batches // 2, n_batches - 1):
                    outs[k] = sc
            return np.array(ms), np.array(asked, float), np.array(encoded, float), outs

        cold = sweep()
        warm = sweep()
        identical = all(torch.equal(o, r) for outs in (cold[3], warm[3]) for o, r in zip(outs.values(), ref))
        te.embedding_cache_rows = 0
    q = n_batches // 4

    def part(x, sl):
        ms, asked, encoded, _ = x
        return {"ms_per_step": float(np.median(ms[sl])), "hit_rate": float(1.0 - encoded[sl].sum() / max(asked[sl].sum(), 1.0)),
                "news_per_step": float(asked[sl].mean()), "encoded_per_step": float(encoded[sl].mean())}
    return {"what": dropin_cached_sweep.__doc__.split("Reported")[0].strip(), "batches": n_batches, "impressions_per_batch": bs,
            "cache_rows": int(n_pool), "cache_MB": round(n_pool * te.plm_model.cfg.hidden * 4 / 1e6, 1),
            "cold_first_quarter": part(cold, slice(0, q)), "cold_last_quarter": part(cold, slice(n_batches - q, n_batches)),
            "cold_whole_sweep": part(cold, slice(0, n_batches)), "second_pass_all_cached": part(warm, slice(0, n_batches)),
            "scores_bit_identical_to_uncached": bool(identical),
            "note": "a MIND-small dev set is 73 152 impressions over 65 238 news (9 144 batches of 8): its sweep-wide hit rate is above 0.98; the "
                    "synthetic batches here only begin to fill the table.  A batch with ANY unseen news still pays the latency floor of its two "
                    "small encoder calls (1.8-2.1 ms each up to 64 news: tools/small_call_probe.py), so the gain grows with the batch size.  "
                    "Not used by the headline metric or by B8_eval / B64_eval."}


# --------------------------------------------------------------------------------------------------- table mode
def ranking_agreement(a, b, labels, off):
    """Ranking metrics of score vector a (throughput mode) against b (the HIP fp32 parity mode, itself pinned to the
    reference at 1e-5) over all impressions: on the device."""
    ta, na, ma = hip.rank_ndcg(a, labels, off, 10, with_mrr=True)
    tb, nb_, mb = hip.rank_ndcg(b, labels, off, 10, with_mrr=True)
    same = (ta == tb).all(dim=1)
    inter = (ta.unsqueeze(2) == tb.unsqueeze(1)) & (ta.unsqueeze(2) >= 0)
    overlap = inter.any(dim=2).sum(1).double() / (tb >= 0).sum(1).clamp(min=1).double()
    return {"impressions": int(same.numel()), "top10_identical_frac": float(same.double().mean()),
            "top1_identical_frac": float((ta[:, 0] == tb[:, 0]).double().mean()),
            "top10_set_overlap_mean": float(overlap.mean()),
            "ndcg10": float(na.double().mean()), "ndcg10_parity_mode": float(nb_.double().mean()),
            "ndcg10_delta": float((na.double().mean() - nb_.double().mean()).abs()),
            "ndcg10_per_impression_abs_delta_mean": float((na.double() - nb_.double()).abs().mean()),
            "mrr_delta": float((ma.double().mean() - mb.double().mean()).abs()),
            "score_max_abs_err": float((a - b).abs().nan_to_num(0.0).max()), "score_abs_scale": float(b.abs().nan_to_num(0.0).max())}


def table_mode(args, conf, cfg, encs, fuse_w, pool, rank, world, dev, scale_parity):
    """Mode T: every rank encodes its FLOP-balanced shard of the unique-news pool once per module,
    one RCCL all-gather per module assembles the [N_news, D] table on every rank, then each rank scores its block of
    the configuration's dev-set-shaped impressions by index.  Reported beside the headline, never as it."""
    from manner_amd import distributed as D
    pool_ids, pool_mask, pool_len = pool
    n_news = pool_ids.shape[0]
    n_imp = conf["dims"]["n_impressions"]
    shards = D.equal_news_shards(n_news, world)         # equal rows: the all-gather writes the table in place
    lo, hi = shards[rank]
    mx_rows = max(h - l for l, h in shards)
    imp = synth_impressions(n_imp, n_news, seed=43)
    ho, co = imp["hist_off"], imp["cand_off"]
    a, b = D.balanced_impression_shards(ho, co, world)[rank]    # equal history + candidate occurrences per rank (SURVEY §8e phase C)
    dimp = {"hist_idx": torch.from_numpy(imp["hist_idx"][ho[a]:ho[b]]).to(dev), "hist_off": torch.from_numpy(ho[a:b + 1] - ho[a]).to(dev),
            "cand_idx": torch.from_numpy(imp["cand_idx"][co[a]:co[b]]).to(dev), "cand_off": torch.from_numpy(co[a:b + 1] - co[a]).to(dev)}
    labels = torch.from_numpy(imp["labels"][co[a]:co[b]]).to(dev)
    K = len(encs)
    # The table is assembled in place (distributed.MeshTableGather): the encoder writes its shard's rows straight into the [N_news, D]
    # table.  Exchange "collective" (default): ONE all_gather_into_tensor per module after the encoding.  Exchange "mesh": every
    # finished piece goes out to the peers point to point while the next piece is being encoded — measured further down, AFTER the
    # default pipeline's figures are safe, and only quoted when its table equals the collective's bit for bit.
    gathers = [D.MeshTableGather(n_news, cfg.hidden, dev, pieces=1 if world == 1 else 4) for _ in range(K)]
    local = [g.table[lo:hi] for g in gathers]           # this rank's rows of each table (world == 1: the whole table)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def encode_all(prec, gs=None, post=False):
        gs = gs or gathers
        for k in range(K):
            for c in range(gs[k].pieces):
                a_, b_ = gs[k].piece_rows(rank, c)
                if b_ > a_:
                    encs[k].encode_cls(pool_ids[a_:b_], pool_mask[a_:b_], precision=prec, host_lengths=pool_len[a_:b_],
                                       max_chunk_tokens=args.chunk_tokens, out=gs[k].local_out(c))
                if post:
                    gs[k].post(c)

    def pipeline(gs):
        out_t, res_, tabs = {}, None, None
        for it in range(2):                             # pass 0 warms up (workspace, RCCL channels)
            sync(); t0 = time.perf_counter()
            encode_all(args.precision, gs, post=True)
            torch.cuda.current_stream().synchronize()   # this rank's encoding is done; mesh transfers of the last pieces may still fly
            t1 = time.perf_counter()
            torch.cuda.reset_peak_memory_stats(dev)
            before = torch.cuda.memory_allocated(dev)
            tabs = [g.wait() for g in gs]
            exchange_extra[0] = max(exchange_extra[0], torch.cuda.max_memory_allocated(dev) - before)   # bytes the exchange itself allocated
            sync(); t2 = time.perf_counter()
            if b > a:                                   # a rank may hold no impressions when there are fewer impressions than ranks
                res_ = hotpath.score_impressions(tabs, dimp, weights=fuse_w, labels=labels, k=10)
            sync(); t3 = time.perf_counter()
            out_t = {"encode_s": t1 - t0, "allgather_s": t2 - t1, "score_s": t3 - t2, "total_s": t3 - t0}
        return out_t, res_, tabs

    exchange_extra = [0]
    times, res, tables = pipeline(gathers)
    if res is None:
        res = {"scores": torch.zeros(0, device=dev), "ndcg": torch.zeros(0, device=dev)}
    exchange_ran = gathers[0].exchange                  # what actually moved the table: "collective", "mesh" or "none" (world 1)
    # the collective alone, nothing to hide behind: ONE all_gather_into_tensor (RCCL's own algorithm choice) per module
    coll_s = None
    if world > 1:
        # ... IN PLACE, as the pipeline runs it (round 5: the table is the head of the [W * mx, D] gather buffer; no clone, no copy back);
        # then once more through fresh buffers (all_gather_table) as an independent check of the in-place form, not timed
        coll_gathers = gathers if exchange_ran == "collective" else [D.MeshTableGather(n_news, cfg.hidden, dev, pieces=1, exchange="collective") for _ in range(K)]
        if coll_gathers is not gathers:
            for k in range(K):
                coll_gathers[k].table[lo:hi].copy_(local[k])
        sync(); t0 = time.perf_counter()
        for g in coll_gathers:
            g.wait()
        sync(); coll_s = time.perf_counter() - t0
        blocks = [torch.zeros((mx_rows, cfg.hidden), dtype=torch.float32, device=dev) for _ in range(K)]
        for k in range(K):
            blocks[k][: hi - lo] = local[k]
        gathered = [D.all_gather_table(blocks[k], shards) for k in range(K)]
        same = all(bool(torch.equal(gathered[k], tables[k])) and bool(torch.equal(coll_gathers[k].table, tables[k])) for k in range(K))
        del blocks, gathered
    else:
        same = True
    sc_r, off_r = res["scores"], dimp["cand_off"]
    n_c = int(sc_r.numel())
    # the fused scorer alone on the first table, by HIP events (the score_s above also holds fusion + ranking)
    scorer_ms = timed_ms(lambda: hip.score_late_fusion(tables[0], dimp["hist_idx"], dimp["hist_off"], dimp["cand_idx"], dimp["cand_off"]))
    # the same scorer over the IEEE-half copy of the table (manner_hip_score_late_fusion_f16): half the bytes per gathered row and,
    # at the MIND-large shape, a table that fits the 256 MiB Infinity Cache; its scores against the f32-table scores
    s32_one = hip.score_late_fusion(tables[0], dimp["hist_idx"], dimp["hist_off"], dimp["cand_idx"], dimp["cand_off"])
    f16_table = {"table_MB": n_news * cfg.hidden * 2 / 1e6,
                 "what": "module 0's table as IEEE half (rows rounded to 11 bits; f32 accumulation) — `plain`: half(T); `centred`: half(T - column mean) "
                         "with the scores reassembled exactly: for tables the 16-bit encoder modes produced"}
    for name, centre in (("plain", False), ("centred", True)):
        t16 = hip.table_to_f16(tables[0], centre=centre)
        conv_ms = timed_ms(lambda: hip.table_to_f16(tables[0], centre=centre))
        ms16 = timed_ms(lambda: hip.score_late_fusion(t16, dimp["hist_idx"], dimp["hist_off"], dimp["cand_idx"], dimp["cand_off"]))
        s16_one = hip.score_late_fusion(t16, dimp["hist_idx"], dimp["hist_off"], dimp["cand_idx"], dimp["cand_off"])
        f16_table[name] = {"scorer_kernel_ms_rank0": ms16, "convert_ms": conv_ms, "speedup_vs_f32_table": scorer_ms / ms16,
                           "vs_f32_table_scores": ranking_agreement(s16_one, s32_one, labels, off_r)}
    del t16, s32_one, s16_one
    # SURVEY §8e phase C as ONE launch (hip.score_fuse_rank: the K score planes stay in LDS) against the K + 2 launches it replaces, same
    # impressions, same tables; the two give the same bits (tests/test_gpu_parity.py::test_phase_c_in_one_launch_equals_the_three_kernel_path)
    phase_c = None
    if b > a and cfg.hidden in (768, 1024):
        fus = hotpath.score_impressions(tables, dimp, weights=fuse_w, labels=labels, k=10, fused=True)
        sep = hotpath.score_impressions(tables, dimp, weights=fuse_w, labels=labels, k=10, fused=False)
        same_bits = bool(torch.equal(fus["scores"].nan_to_num(0.0), sep["scores"].nan_to_num(0.0)) and torch.equal(fus["topk"], sep["topk"])
                         and torch.equal(fus["ndcg"].nan_to_num(0.0), sep["ndcg"].nan_to_num(0.0)))
        ms_f = timed_ms(lambda: hotpath.score_impressions(tables, dimp, weights=fuse_w, labels=labels, k=10, fused=True))
        ms_s = timed_ms(lambda: hotpath.score_impressions(tables, dimp, weights=fuse_w, labels=labels, k=10, fused=False))
        phase_c = {"one_launch_ms": ms_f, "separate_launches_ms": ms_s, "launches_replaced": K + (2 if K > 1 else 1), "bit_identical": same_bits,
                   "impressions_rank0": int(b - a), "what": "gather-mean-dot x K modules + z-score + weighted fusion + stable top-10 + nDCG@10 + MRR per "
                   "impression in one kernel (planes in LDS) vs score_late_fusion x K -> zscore_fuse -> rank_ndcg"}
        del fus, sep
    metrics_ms = {"rank_ndcg_mrr": timed_ms(lambda: hip.rank_ndcg(sc_r, labels, off_r, 10, with_mrr=True)),
                  "auc": timed_ms(lambda: hip.auc(sc_r.nan_to_num(0.0), labels)),
                  "eval_loss_supcon": timed_ms(lambda: hip.eval_loss(sc_r, labels, off_r, supcon=True, temperature=0.36, reduce=False))}
    st = torch.tensor([times["encode_s"], times["allgather_s"], times["score_s"], times["total_s"], coll_s or 0.0],
                      dtype=torch.float64, device=dev)
    _, ndcg5 = hip.rank_ndcg(sc_r, labels, off_r, 5)                                   # SURVEY §8e phase D: (sum nDCG@10, sum nDCG@5, count)
    nd = torch.tensor([float(res["ndcg"].double().sum()), float(b - a), float(ndcg5.double().sum())], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(st, op=torch.distributed.ReduceOp.MAX)
        D.allreduce_metric_sums(nd)
    enc_s, ag_s, sc_s, tot_s, coll_s = st.tolist()
    recv_bytes = (n_news - (hi - lo)) * cfg.hidden * 4 * K
    total_c = int(co[-1])
    occ_local = float((ho[b] - ho[a]) + (co[b] - co[a]))
    scorer_bytes = occ_local * (cfg.hidden * 4 + 4) + float(co[b] - co[a]) * 4
    table_bytes = n_news * cfg.hidden * 4
    out = {"what": f"unique-news table ({n_news} news x {K} module(s)): encode shard -> all-gather -> score all {n_imp} "
                   f"{conf['shape']}-shaped impressions",
           "candidates_per_s": total_c / tot_s, "news_encoded_per_s": n_news * K / enc_s,
           "scorer_pairs_per_s": total_c * K / sc_s,
           "scorer_kernel_ms_rank0": scorer_ms, "scorer_algorithmic_bytes_rank0": scorer_bytes,
           "scorer_GBps_algorithmic": scorer_bytes / scorer_ms / 1e6,
           "scorer_occurrence_rate_over_8TBps": scorer_bytes / scorer_ms / 1e6 / HBM_PEAK_GBS,
           "table_MB": table_bytes / 1e6, "scorer_f16_table": f16_table,
           "scorer_note": ("`scorer_occurrence_rate_over_8TBps` counts every history/candidate row read once; rows repeat (Zipf) and a table "
                           "that fits the 256 MiB Infinity Cache is served from cache, so it is a gather rate relative to 8 TB/s (it exceeds 1), NOT "
                           "an HBM fraction — the HBM-side fraction from the FETCH_SIZE pass is in profiles/*/tail_pmc.json" if table_bytes < 256 * 2 ** 20 else
                           "`scorer_occurrence_rate_over_8TBps` counts every history/candidate row read once; the table exceeds the 256 MiB Infinity "
                           "Cache but popular rows (Zipf) still hit, so it is a gather rate relative to 8 TB/s, not an HBM fraction — the HBM-side "
                           "fraction from the FETCH_SIZE pass is in profiles/*/tail_pmc.json"),
           "allgather_ms": 1e3 * ag_s,
           "allgather_exchange": exchange_ran, "allgather_exchange_why": gathers[0].exchange_why,
           "allgather_what": ("no exchange (one rank)" if world == 1 else
                              "EXPOSED time of the table exchange, from the end of this rank's encoding to the complete table on every rank: " +
                              (f"ONE all_gather_into_tensor per module after the encoding (backend {D.device_backend('cuda')!r}: its own algorithm)" if exchange_ran == "collective" else
                               f"direct full mesh of point-to-point transfers, {gathers[0].pieces} pieces per shard, each posted while the next is encoded")),
           "allgather_bytes_per_rank": recv_bytes if world > 1 else 0,
           "allgather_extra_alloc_MB_rank0": exchange_extra[0] / 1e6,        # 0: the in-place exchange allocates nothing (was 2 x table per module)
           "allgather_standalone": None if world == 1 else {
               "collective_ms": 1e3 * coll_s, "collective_GBps_per_rank": recv_bytes / coll_s / 1e9,
               "collective_frac_of_xgmi": recv_bytes / coll_s / 1e9 / (7 * 153.0), "equals_pipeline_table": same,
               "what": "the exchange with nothing to overlap: ONE in-place all_gather_into_tensor per module; bytes = what a rank receives; "
                       "xGMI peak = 7 links x 153 GB/s; the mesh form is measured in `mesh_exchange`"},
           "allgather_GBps_per_rank": (recv_bytes / coll_s / 1e9) if world > 1 and coll_s > 0 else None,
           "allgather_frac_of_xgmi": (recv_bytes / coll_s / 1e9 / (7 * 153.0)) if world > 1 and coll_s > 0 else None,
           "world_size_seen": world, "phase_c": phase_c,
           "encode_ms": 1e3 * enc_s, "score_ms": 1e3 * sc_s, "ndcg10": nd[0].item() / nd[1].item(), "ndcg5": nd[2].item() / nd[1].item(),
           "metrics_ms_rank0": {**metrics_ms, "candidates": n_c,
                                "auc_Mpairs_per_s": n_c / metrics_ms["auc"] / 1e3, "rank_Mpairs_per_s": n_c / metrics_ms["rank_ndcg_mrr"] / 1e3}}
    parity = None
    if scale_parity and world == 1:
        # the 16-bit throughput modes against the HIP fp32 mode (pinned to the reference at <= 2e-5 by
        # the golden tests) on the WHOLE dev-set shape, same tables / impressions, both weight sets
        log("at-scale parity: fp32 encode of the pool")
        fast = {args.precision: sc_r.clone()} if args.precision != "fp32" else {}
        speed = {}
        for mode in ("bf16", "f16") + (("f16x3",) if K == 1 else ()):    # f16x3: the split-operand parity-grade mode
            if mode in fast or args.precision == "fp32":
                continue
            torch.cuda.synchronize(); t0 = time.perf_counter()
            encode_all(mode)
            torch.cuda.synchronize(); speed[mode] = n_news * K / (time.perf_counter() - t0)
            fast[mode] = hotpath.score_impressions([t[: hi - lo] for t in local], dimp, weights=fuse_w, labels=labels, k=10)["scores"].clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        encode_all("fp32")
        torch.cuda.synchronize(); t_f32 = time.perf_counter() - t0
        res32 = hotpath.score_impressions([t[: hi - lo] for t in local], dimp, weights=fuse_w, labels=labels, k=10)
        parity = {"what": f"16-bit modes vs the HIP fp32 parity mode, table mode, {n_news} news, all {n_imp} impressions",
                  "hf_init_weights_std0.02": {m: ranking_agreement(v, res32["scores"], labels, off_r) for m, v in fast.items()},
                  "other_mode_news_per_s": speed,
                  "parity_mode_news_per_s": n_news * K / t_f32, "parity_mode_encode_s": t_f32}
    if world > 1:
        # LAST step of the leg, after every figure above is final: the mesh form of the exchange.  It has never run on RCCL (no
        # multi-GPU node in rounds 1-4), so it is bounded in time, agreed on over the rendezvous store (no GPU collective that a
        # stuck rank could hang), checked bit for bit against the collective's tables, and only then timed.
        out["mesh_exchange"] = mesh_exchange_leg(n_news, cfg, K, tables, shards, rank, world, dev, recv_bytes, pipeline, sync)
    return out, parity, (dimp, labels)


def store_agree(tag, ok, rank, world, timeout_s):
    """True iff EVERY rank reports ok within the limit — over the process group's rendezvous store (TCP on the host): it cannot
    hang on a GPU transfer that makes no progress, which is exactly the case it arbitrates."""
    import datetime
    import torch.distributed as dist
    store = dist.distributed_c10d._get_default_store()
    store.set(f"manner/{tag}/{rank}", "1" if ok else "0")
    try:
        store.wait([f"manner/{tag}/{r}" for r in range(world)], datetime.timedelta(seconds=timeout_s))
        return all(store.get(f"manner/{tag}/{r}") == b"1" for r in range(world))
    except Exception:          # noqa: BLE001   (a rank that never reports: time-out of the store wait)
        return False


def mesh_exchange_leg(n_news, cfg, K, tables, shards, rank, world, dev, recv_bytes, pipeline, sync, pieces=4, timeout_s=60.0):
    from manner_amd import distributed as D
    mesh = [D.MeshTableGather(n_news, cfg.hidden, dev, pieces=pieces, exchange="mesh", timeout_s=timeout_s) for _ in range(K)]
    info = {"exchange": mesh[0].exchange, "why": mesh[0].exchange_why, "pieces": pieces, "bytes_received_per_rank": recv_bytes,
            "what": "direct full mesh of point-to-point transfers (distributed.MeshTableGather, exchange='mesh'): all pieces posted at once "
                    "(standalone), then the overlapped pipeline — every figure only after the table equals the collective's bit for bit"}
    if mesh[0].exchange != "mesh":
        info["skipped"] = "no GPU point-to-point transfers on this backend; the collective ran instead"
        return info
    lo, hi = shards[rank]
    ok, err, t_first = True, None, None
    try:
        for k in range(K):
            mesh[k].table.fill_(float("nan"))
            mesh[k].table[lo:hi].copy_(tables[k][lo:hi])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in mesh:
            for c in range(pieces):
                g.post(c)
        for g in mesh:
            g.wait()                                                        # bounded: TimeoutError after timeout_s
        t_first = time.perf_counter() - t0
        identical = all(bool(torch.equal(mesh[k].table, tables[k])) for k in range(K))
    except Exception as e:      # noqa: BLE001
        ok, err, identical = False, f"{type(e).__name__}: {e}", False
    if not store_agree("mesh_first", ok, rank, world, timeout_s + 30.0):
        info.update({"error": err or "another rank's mesh exchange failed or timed out", "fatal": True, "tables_identical": False})
        return info
    flag = torch.tensor([1.0 if identical else 0.0, t_first], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(flag[:1], op=torch.distributed.ReduceOp.MIN)
    info["tables_identical"] = bool(flag[0].item() == 1.0)
    info["first_exchange_s_rank0"] = t_first
    if not info["tables_identical"]:
        info["error"] = "the mesh table differs from the collective's table — mesh timings not reported"
        return info
    sync(); t0 = time.perf_counter()
    for g in mesh:
        for c in range(pieces):
            g.post(c)
    for g in mesh:
        g.wait()
    sync(); mesh_s = time.perf_counter() - t0
    times, _, _ = pipeline(mesh)                                            # encode with the pieces posted as they finish
    st = torch.tensor([mesh_s, times["encode_s"], times["allgather_s"], times["score_s"], times["total_s"]], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(st, op=torch.distributed.ReduceOp.MAX)
    mesh_s, enc_s, ag_s, sc_s, tot_s = st.tolist()
    info.update({"standalone_ms": 1e3 * mesh_s, "standalone_GBps_per_rank": recv_bytes / mesh_s / 1e9,
                 "standalone_frac_of_xgmi": recv_bytes / mesh_s / 1e9 / (7 * 153.0),
                 "overlapped_pipeline": {"encode_ms": 1e3 * enc_s, "exposed_exchange_ms": 1e3 * ag_s, "score_ms": 1e3 * sc_s, "total_ms": 1e3 * tot_s}})
    return info


# --------------------------------------------------------------------------------------------------- launcher
def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N fresh rank processes of this script
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them, rendezvous on 127.0.0.1) and wait.
    Runs BEFORE anything in this process has touched the GPU (no torch.cuda call, no encoder): the children are ordinary
    child processes (never an exec of a process that initialised HIP).  Rank 0 inherits stdout, so its one JSON line is this
    command's output; the exit code is the first non-zero child exit code (the other ranks are then terminated by PID)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MANNER_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p_ in list(live):
            code = p_.poll()
            if code is None:
                continue
            live.remove(p_)
            if code != 0 and rc == 0:
                rc = code
                for q in live:                                          # a rank died: the others would wait in a collective forever
                    q.terminate()
    return rc


def dry_run(args, rank, world):
    """MANNER_BENCH_DRY=1 (CPU rehearsal of the multi-rank plumbing, tests/test_host.py): rendezvous, one all-reduce and the
    barrier / MAX-over-ranks protocol of the timed region without any GPU work.  Prints a line marked ``dry_run``; never a
    measurement."""
    import torch.distributed as dist
    seen = 1
    if world > 1:
        dist.init_process_group(os.environ.get("MANNER_DIST_BACKEND", "gloo"))
        seen = dist.get_world_size()
        dist.barrier()
    t0 = time.perf_counter()
    stats = torch.tensor([0.001 * (rank + 1), float(args.impressions)], dtype=torch.float64)
    exch = None
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        stats[0] = mx[0]
        dist.barrier()
        # the table exchange of table_mode on a CPU stand-in table of the MIND-large row count: the default (collective) form, then the
        # mesh form under the bench's own protocol — bounded wait, agreement over the rendezvous store, bit-for-bit comparison
        from manner_amd import distributed as D
        n_news, dim = MIND_LARGE["n_news"], 4
        full = (torch.arange(n_news * dim, dtype=torch.float32).reshape(n_news, dim) * 0.25).contiguous()
        tabs = {}
        for form in ("collective", "mesh"):
            g = D.MeshTableGather(n_news, dim, "cpu", pieces=4, exchange=form, timeout_s=60)
            g.table.fill_(float("nan"))
            ok = True
            try:
                for c in range(g.pieces):
                    a, b = g.piece_rows(rank, c)
                    g.local_out(c).copy_(full[a:b])
                    g.post(c)
                tabs[form] = g.wait()
            except Exception:      # noqa: BLE001
                ok = False
            if not store_agree("dry_" + form, ok, rank, world, 90.0):
                raise SystemExit(f"dry run: the {form} exchange failed on some rank")
        same = torch.tensor([float(torch.equal(tabs["mesh"], tabs["collective"]) and torch.equal(tabs["mesh"], full))])
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        exch = {"default_exchange": D.MeshTableGather(8, dim, "cpu").exchange, "mesh_equals_collective_on_every_rank": bool(same.item() == 1.0),
                "n_news": n_news, "pieces": 4}
    if rank == 0:
        print(json.dumps({"metric": "candidate news encoded+scored/sec", "value": None, "unit": "candidates/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "world_size_seen": seen,
                          "launcher": os.environ.get("MANNER_BENCH_LAUNCHER", "external"),
                          "impressions_all_ranks": stats[1].item(), "max_rank_time_s": stats[0].item(), "table_exchange": exch,
                          "wall_s": time.perf_counter() - t0}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def summarise_for_driver(result, args):
    """The driver's record keeps `config`, `roofline` and `cpu_baseline` of the line and drops the other objects:
    the parity figures of every arithmetic mode and the encoder-wide MFMA fractions are therefore repeated, as plain numbers, inside
    `config` and `roofline`.  Nothing here is new information — every figure is copied from the leg that measured it."""
    par, scale, pm = result.get("parity"), result.get("parity_at_scale") or {}, result.get("parity_mode") or {}
    speed = {args.precision: result.get("news_encoded_per_s"), "bf16": (result.get("bf16_mode") or {}).get("news_encoded_per_s"),
             "fp32": pm.get("news_encoded_per_s"), "f16x3": pm.get("f16x3_news_encoded_per_s")}
    speed.update({k: v for k, v in (scale.get("other_mode_news_per_s") or {}).items() if speed.get(k) is None})
    if par:
        summ = {"impressions": par["impressions"], "candidates": par["candidates"], "repeated_candidates": par["repeated_candidates_in_sample"]}
        for m in ("fp32", "f16x3", "f16", "bf16"):
            if m in par:
                summ[m] = {"top10_identical": par[m]["top10_identical"], "top10_valid_order_frac": par[m]["top10_valid_order_of_oracle_scores_frac"],
                           "ndcg10_delta": par[m]["ndcg10_delta"], "score_max_abs_err": par[m]["score_max_abs_err"],
                           "news_per_s": speed.get(m)}
        if "against_float64_oracle" in par:
            summ["max_abs_err_vs_float64_oracle"] = {k: v for k, v in par["against_float64_oracle"].items() if k.endswith("_err")}
        result["config"]["parity_vs_oracle"] = summ
    if "roofline" in result:
        result["roofline"]["encoder_mfma_frac"] = {args.precision: result.get("encoder_mfma_frac"),
                                                   "bf16": (result.get("bf16_mode") or {}).get("encoder_mfma_frac")}


def _tag(s, n=100):
    """Prose of the full record cut to a short tag for the compact line."""
    if not isinstance(s, str):
        return s
    s = " ".join(s.split())
    return s if len(s) <= n else s[: n - 1].rstrip() + "~"


def _clean(x, digits=6):
    """Strict-JSON numbers (no NaN / Infinity), floats rounded to `digits` significant digits, strings as tags."""
    if isinstance(x, dict):
        return {str(k): _clean(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v, digits) for v in x]
    if isinstance(x, (bool, int)) or x is None:
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, np.integer):
        return int(x)
    return _tag(x)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(result, limit=COMPACT_LIMIT):
    """The ONE line the driver parses: the contract keys and the three objects (`config`, `roofline`, `cpu_baseline`), every prose string a
    short tag, below `limit` bytes.  Every other leg lives in the full record (`--full-json`, default bench_full.json beside this file);
    `legs` repeats a handful of their headline NUMBERS.  Nothing here is measured again — every figure is copied from `result`."""
    line = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = result.get("vs_baseline")
    line.update(_pick(result, ("dtype", "data")))
    cfg = result.get("config") or {}
    c = _pick(cfg, ("workload", "baseline_config", "modules", "ensemble_weights", "impressions_per_step_per_gpu", "length_profile",
                    "seeded_weights_std", "parallelism"))
    pv = cfg.get("parity_vs_oracle")
    if pv:
        c["parity_vs_oracle"] = {k: (_pick(v, ("top10_identical", "top10_valid_order_frac", "ndcg10_delta", "score_max_abs_err", "news_per_s"))
                                      if isinstance(v, dict) and k in ("fp32", "f16x3", "f16", "bf16") else v) for k, v in pv.items()}
    pg = result.get("parity_grade_mode")
    if pg:
        g = {"dtype": pg["dtype"], "candidates_per_s": pg["value"], "ms_per_step": pg["ms_per_step"], "news_per_s": pg["news_encoded_per_s"],
             "timed_steps": pg["steps"]}
        m = (pv or {}).get(pg["dtype"]) or {}
        if m:
            g.update({"top10_identical": m.get("top10_identical"), "of_impressions": (pv or {}).get("impressions"), "ndcg10_delta": m.get("ndcg10_delta")})
        c["parity_grade"] = g
    line["config"] = c
    if "roofline" in result:
        line["roofline"] = _pick(result["roofline"], ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                      "avg_launch_us", "flops_per_launch", "mfma_only_ceiling_tflops", "encoder_mfma_frac"))
        line["roofline"].setdefault("traffic", None)
    if "cpu_baseline" in result:
        line["cpu_baseline"] = _pick(result["cpu_baseline"], ("value", "unit", "cores", "kind", "cpu_model", "runs_s", "sample"))
    legs = {}
    legs.update(_pick(result, ("news_encoded_per_s", "tokens_per_s", "encoder_tflops", "encoder_mfma_frac", "ndcg10_last_step", "world_size_seen",
                               "dist_backend", "launcher")))
    if result.get("bf16_mode"):
        legs["bf16"] = _pick(result["bf16_mode"], ("value", "ms_per_step", "encoder_mfma_frac"))
    if result.get("pcie_inclusive_rank0"):
        legs["pcie_inclusive"] = _pick(result["pcie_inclusive_rank0"], ("candidates_per_s", "ms_per_step"))
    tm = result.get("table_mode") or {}
    if tm:
        legs["table_mode"] = _pick(tm, ("candidates_per_s", "news_encoded_per_s", "scorer_pairs_per_s", "scorer_kernel_ms_rank0", "allgather_ms",
                                        "allgather_exchange", "allgather_GBps_per_rank", "allgather_frac_of_xgmi", "allgather_extra_alloc_MB_rank0", "encode_ms", "score_ms", "ndcg10", "error"))
        me = tm.get("mesh_exchange")
        if isinstance(me, dict):
            legs["table_mode"]["mesh"] = _pick(me, ("exchange", "tables_identical", "standalone_ms", "standalone_frac_of_xgmi", "error", "skipped"))
    ps = (result.get("parity_at_scale") or {}).get("hf_init_weights_std0.02") or {}
    if ps:
        legs["parity_at_scale_vs_hip_fp32"] = {m: _pick(v, ("top10_identical_frac", "ndcg10_delta", "score_max_abs_err")) for m, v in ps.items()}
    so = result.get("small_ops") or {}
    if so:
        legs["small_ops_frac_of_8TBps"] = {k: v.get("frac_of_8TBps") for k, v in so.items() if isinstance(v, dict)}
        legs["small_ops_ms"] = {k: v.get("ms") for k, v in so.items() if isinstance(v, dict)}
    tr = result.get("train_mode") or {}
    if tr:
        legs["train_ms_per_step"] = {k: v.get("ms_per_step") for k, v in tr.items() if isinstance(v, dict) and "ms_per_step" in v}
        legs["train_peak_GB"] = {k: v.get("peak_GB") for k, v in tr.items() if isinstance(v, dict) and "peak_GB" in v}
        legs["train_frac_of_mfma_peak"] = {k: v.get("frac_of_mfma_peak") for k, v in tr.items() if isinstance(v, dict) and "frac_of_mfma_peak" in v}
    dr = result.get("dropin") or {}
    if dr:
        legs["dropin_ms_per_step"] = {k: v.get("ms_per_step") for k, v in dr.items() if isinstance(v, dict) and "ms_per_step" in v}
    if result.get("collate"):
        legs["collate_ms_per_batch"] = result["collate"].get("ms_per_batch")
    kern = result.get("kernels") or {}
    if kern:
        legs["kernel_avg_us"] = {k: v.get("avg_us") for k, v in kern.items()}
    errors = {k: _tag(str(result[k]["error"])) for k in ("table_mode", "small_ops", "cpu_baseline", "train_mode", "dropin", "collate")
              if isinstance(result.get(k), dict) and result[k].get("error")}
    if errors:
        legs["errors"] = errors
    line["legs"] = legs
    line["full_record"] = result.get("full_record")
    line = _clean(line)
    # belt and braces: the line must fit whatever a leg returns — shed the optional parts, largest first, until it does
    for drop in (None, ("legs", "kernel_avg_us"), ("legs", "parity_at_scale_vs_hip_fp32"), ("legs", "table_mode"), ("legs",),
                 ("config", "parity_vs_oracle"), ("cpu_baseline", "runs_s")):
        if drop is not None:
            d = line
            for k in drop[:-1]:
                d = d.get(k, {})
            d.pop(drop[-1], None)
        out = json.dumps(line, allow_nan=False, separators=(", ", ": "))
        if len(out.encode()) < limit:
            return out
    # cannot happen with the fields above (every one is a number or a <= 100-character tag); if it ever does, the line still goes out:
    # the contract scalars and the three objects reduced to their numbers
    line["config"] = _pick(line.get("config") or {}, ("baseline_config", "modules", "impressions_per_step_per_gpu"))
    for k in ("roofline", "cpu_baseline"):
        line[k] = {kk: v for kk, v in (line.get(k) or {}).items() if isinstance(v, (int, float)) or v is None or kk in ("bound", "unit", "kind")}
    return json.dumps(line, allow_nan=False, separators=(", ", ": "))


def emit(result, args):
    """Full record -> file (and nothing of it on stdout); compact line -> the LAST line of stdout."""
    try:
        summarise_for_driver(result, args)
    except Exception as e:      # noqa: BLE001   (copies of figures already in the record: never worth the line)
        log(f"summarise_for_driver failed: {type(e).__name__}: {e}")
    path = args.full_json
    try:
        with open(path, "w") as f:
            json.dump(result, f, indent=1, default=str)
        result["full_record"] = os.path.relpath(path, ROOT)
        side = os.path.join(ROOT, "gpurun_out")             # scratch that travels back from the GPU box
        if os.path.isdir(side) and os.path.dirname(os.path.abspath(path)) == ROOT:
            with open(os.path.join(side, "bench_full.json"), "w") as f:
                json.dump(result, f, indent=1, default=str)
    except (OSError, TypeError, ValueError) as e:
        log(f"could not write {path}: {e}")
        result["full_record"] = None
    sys.stderr.flush()
    print(compact_line(result), flush=True)


FAILED_LEGS = []


def guarded(name, fn, *a, **kw):
    """A side leg must not take the headline line with it: run it, and on ANY failure log the reason and return {"error": ...}.
    Every guarded leg runs on rank 0 of a single-rank job only (no collective inside one), so a failure cannot strand a peer."""
    try:
        return fn(*a, **kw)
    except Exception as exc:      # noqa: BLE001
        import traceback
        FAILED_LEGS.append(name)
        log(f"{name} FAILED: " + "".join(traceback.format_exception_only(type(exc), exc)).strip())
        traceback.print_exc(file=sys.stderr)
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


# --------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    conf = CONFIGS[args.config]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: drop the launcher (bench.py starts its own ranks) or make them agree")
    if os.environ.get("MANNER_BENCH_DRY"):
        return dry_run(args, rank, world)
    if os.environ.get("MANNER_BENCH_ONE_DEVICE"):      # rehearsal: every rank on cuda:0
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        # RCCL over xGMI ("nccl" is RCCL on ROCm); MANNER_DIST_BACKEND=gloo only to rehearse the
        # multi-rank logic on a one-GPU box
        backend = os.environ.get("MANNER_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        log(f"process group up: backend {backend}, world size {dist.get_world_size()} (launcher: {os.environ.get('MANNER_BENCH_LAUNCHER', 'external')})")

    model = args.model or conf["model"]
    cfg = PRESETS[model]
    fuse_w = tuple(conf["weights"])
    K = 1 + len(fuse_w)
    log(f"config {args.config}: {conf['what']}; generating seeded weights for {K} module(s)")
    weight_sets = [make_plm_weights(cfg, seed=42 + 100 * k, std=args.std) for k in range(K)]
    log("packing weights into the HIP encoders")
    # The two 16-bit MFMA modes run the same kernels at the same MFMA rate.  f16 (IEEE half, the arithmetic of the
    # reference's own `precision: 16-mixed`) is the headline because it is the one whose nDCG@10 stays within 1e-4 of the
    # fp32 parity mode on the whole dev-set shape; bf16 (BASELINE.json's wording for configs[1]) is timed right after it
    # on the same batches and reported as `bf16_mode`, with its own at-scale parity.
    precs = tuple(dict.fromkeys(("bf16", "fp32", "f16", args.precision, PARITY_GRADE)))
    encs = [hip.HipEncoder(cfg, w, precisions=precs, device=dev) for w in weight_sets]
    n_news = conf["dims"]["n_news"]
    log(f"synthesising the {conf['shape']}-shaped news pool ({n_news} news) + impressions")
    pool_ids_np, pool_mask_np = synth_news_tokens(n_news, cfg, seed=42, max_len=96, profile=args.profile)
    pool_len = pool_mask_np.sum(1)
    pool_ids, pool_mask = torch.from_numpy(pool_ids_np).to(dev), torch.from_numpy(pool_mask_np).to(dev)
    n_steps = args.warmup + args.steps
    # one independent 256-impression draw per (rank, step): step s of rank r is the same batch for every --steps and
    # --gpus, so the figure does not depend on how many steps were asked for beyond averaging over more batches
    imp_all = synth_impression_blocks([rank * 1_000_000 + s for s in range(n_steps)], args.impressions, n_news, seed=42)
    batches = [StepBatch(imp_all, s * args.impressions, (s + 1) * args.impressions, pool_ids, pool_mask, pool_len, dev)
               for s in range(n_steps)]
    max_news = max(b.ids.shape[0] for b in batches)
    max_cand = max(b.n_cand for b in batches)
    table_bufs = [torch.empty((max_news, cfg.hidden), dtype=torch.float32, device=dev) for _ in range(K)]
    plane_buf = torch.empty((K, max_cand), dtype=torch.float32, device=dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # initialisation, not a step: every kernel variant the encoders can pick (16-bit and fp32 code objects, the tail-chunk and [CLS]
    # tail shapes) is launched once on a small slice of the pool so that no timed step pays a first-use code-object load
    init_n = min(2048, pool_ids.shape[0])
    for prec in dict.fromkeys((args.precision, "bf16")):
        for k in range(K):
            encs[k].encode_cls(pool_ids[:init_n], pool_mask[:init_n], precision=prec, host_lengths=pool_len[:init_n],
                               max_chunk_tokens=args.chunk_tokens, out=table_bufs[k][:init_n] if init_n <= max_news else None)
    torch.cuda.synchronize()
    log(f"{n_steps} step batches resident ({batches[0].ids.shape[0]} news, {batches[0].tokens} tokens in step 0); warm-up")
    for b in batches[: args.warmup]:
        run_step(encs, b, args.precision, args.chunk_tokens, table_bufs, plane_buf, fuse_w)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for b in batches[args.warmup:]:
        last = run_step(encs, b, args.precision, args.chunk_tokens, table_bufs, plane_buf, fuse_w)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
    for e in encs:
        e.status()
    hip.check_status(dev)
    elapsed_bf16 = None
    if args.precision == "f16":                       # the same steps in bf16, same protocol
        run_step(encs, batches[0], "bf16", args.chunk_tokens, table_bufs, plane_buf, fuse_w)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches[args.warmup:]:
            run_step(encs, b, "bf16", args.chunk_tokens, table_bufs, plane_buf, fuse_w)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed_bf16 = time.perf_counter() - t0
    # ... and in the parity-grade arithmetic (f16x3: f32 activations, every GEMM as split f16 products — the fast mode whose top-10 lists
    # match the reference's): the SAME timed steps, same protocol, so the record carries an index-exact rate next to the headline
    elapsed_pg = None
    if args.precision != PARITY_GRADE and PARITY_GRADE in precs and not args.no_parity_grade:
        run_step(encs, batches[0], PARITY_GRADE, args.chunk_tokens, table_bufs, plane_buf, fuse_w)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches[args.warmup:]:
            run_step(encs, b, PARITY_GRADE, args.chunk_tokens, table_bufs, plane_buf, fuse_w)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed_pg = time.perf_counter() - t0
        log(f"parity-grade ({PARITY_GRADE}) timed region done: {elapsed_pg:.3f} s for {args.steps} steps")

    # PCIe-inclusive rate of the same steps (the boundary takes device tensors — the reference's collate output lives on
    # the host, so this is what a caller pays who feeds host token tensors): ids + mask copied from pinned host memory
    # in front of every step, on the same stream
    pcie_s = None
    if rank == 0:
        host = [(b.ids.cpu().pin_memory(), b.mask.cpu().pin_memory()) for b in batches[args.warmup:]]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b, (hi_, hm_) in zip(batches[args.warmup:], host):
            b.ids.copy_(hi_, non_blocking=True)
            b.mask.copy_(hm_, non_blocking=True)
            run_step(encs, b, args.precision, args.chunk_tokens, table_bufs, plane_buf, fuse_w)
        torch.cuda.synchronize(); pcie_s = time.perf_counter() - t0
        pcie_bytes = float(sum(hi_.numel() * 16 for hi_, _ in host))
        del host

    timed = batches[args.warmup:]
    cands = float(sum(b.n_cand for b in timed))
    news = float(sum(b.n_hist + b.n_cand for b in timed)) * K
    tokens = float(sum(b.tokens for b in timed)) * K
    enc_flops = float(sum(cfg.flops_per_news(int(l)) for b in timed for l in b.lens)) * K
    h_, i_ = cfg.hidden, cfg.intermediate
    per_layer = lambda l: 8 * l * h_ * h_ + 4 * l * h_ * i_ + 4 * l * l * h_      # noqa: E731
    exec_flops = float(sum((cfg.layers - 1) * per_layer(int(l)) + 4 * int(l) * h_ * h_ + 4 * int(l) * h_
                           + 4 * h_ * h_ + 4 * h_ * i_ for b in timed for l in b.lens)) * K
    stats = torch.tensor([elapsed, cands, news, tokens, enc_flops, exec_flops, elapsed_bf16 or 0.0, elapsed_pg or 0.0], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.SUM)
        stats[0], stats[6], stats[7] = mx[0], mx[6], mx[7]
    elapsed_max, cands_all, news_all, tokens_all, flops_all, exec_all, elapsed_bf16_max, elapsed_pg_max = stats.tolist()

    result = None
    if rank == 0:
        peak = BF16_PEAK_TFLOPS if args.precision in ("bf16", "f16", "bf16x3", "f16x3") else F32_PEAK_TFLOPS
        result = {
            "metric": "candidate news encoded+scored/sec", "value": cands_all / elapsed_max, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{conf['what']}, {conf['shape']} shape ({n_news}-news pool), {model} architecture, "
                                   "mode R (every history+candidate occurrence encoded by every active module)",
                       "baseline_config": args.config, "modules": K, "ensemble_weights": list(fuse_w),
                       "impressions_per_step_per_gpu": args.impressions, "length_profile": args.profile,
                       "seeded_weights_std": args.std, "parallelism": f"dp{world} (impressions sharded, no collective)"},
            "world_size_seen": torch.distributed.get_world_size() if world > 1 else 1,
            "dist_backend": None if world == 1 else {"nccl": "nccl (RCCL)"}.get(os.environ.get("MANNER_DIST_BACKEND", "nccl"), os.environ.get("MANNER_DIST_BACKEND")),
            "launcher": os.environ.get("MANNER_BENCH_LAUNCHER", "external") if world > 1 else None,
            "news_encoded_per_s": news_all / elapsed_max, "tokens_per_s": tokens_all / elapsed_max,
            # algorithmic = SURVEY.md §8d F(L) for every news (what the reference computes); executed
            # excludes the last layer's non-[CLS] rows, which the HIP path prunes (not credited as
            # utilisation): last layer costs 4LH^2 (K,V) instead of 8LH^2 + 4LHI + 4L^2 H
            "encoder_tflops_algorithmic": flops_all / elapsed_max / 1e12,
            "encoder_tflops": exec_all / elapsed_max / 1e12,
            "encoder_mfma_frac": exec_all / elapsed_max / 1e12 / (peak * world),
            "ndcg10_last_step": float(last[2].double().mean().item()),
        }
        if pcie_s is not None:
            result["pcie_inclusive_rank0"] = {"what": "rank 0's timed steps again with input_ids + attention_mask (int64) copied from pinned host memory "
                                                      "in front of each step; never the headline value",
                                              "candidates_per_s": float(sum(b.n_cand for b in batches[args.warmup:])) / pcie_s,
                                              "ms_per_step": 1e3 * pcie_s / args.steps, "h2d_MB_per_step": pcie_bytes / args.steps / 1e6}
        if elapsed_bf16 is not None:
            result["bf16_mode"] = {"what": "the same timed steps with bf16 operands (BASELINE.json's wording for configs[1]); its at-scale ranking "
                                           "parity is in parity_at_scale",
                                   "value": cands_all / elapsed_bf16_max, "unit": "candidates/s",
                                   "ms_per_step": 1e3 * elapsed_bf16_max / args.steps,
                                   "news_encoded_per_s": news_all / elapsed_bf16_max,
                                   "encoder_tflops": exec_all / elapsed_bf16_max / 1e12,
                                   "encoder_mfma_frac": exec_all / elapsed_bf16_max / 1e12 / (BF16_PEAK_TFLOPS * world)}
        if elapsed_pg is not None:
            # 3 f16 MFMA products per algorithmic product: utilisation is NOT quoted for this mode, only its rate
            result["parity_grade_mode"] = {"dtype": PARITY_GRADE, "value": cands_all / elapsed_pg_max, "unit": "candidates/s",
                                           "ms_per_step": 1e3 * elapsed_pg_max / args.steps, "news_encoded_per_s": news_all / elapsed_pg_max,
                                           "steps": args.steps, "tag": "same timed steps, split-f16 GEMMs over f32 activations"}

    # per-kernel roofline: same steps again with every launch bracketed by HIP events on the launch stream
    if rank == 0 and not args.no_kernel_profile:
        log("per-kernel HIP-event pass")
        enc = encs[0]
        enc.profile(True)
        for b in timed:
            run_step([enc], b, args.precision, args.chunk_tokens, table_bufs, plane_buf, ())
        prof = enc.profile_read()
        enc.profile(False)
        tok_local = float(sum(b.tokens for b in timed))
        h, i = cfg.hidden, cfg.intermediate
        shape = {"gemm_qkv": (3 * h, h), "gemm_out": (h, h), "gemm_ffn1": (i, h), "gemm_ffn2": (h, i)}
        nl = cfg.layers
        # launches per class per chunk: QKV runs in every layer (the last one as the K|V-only GEMM,
        # N = 2H); out-proj / FFN on all tokens only in the first layers-1 layers
        nk_total = {"gemm_qkv": (3 * h * (nl - 1) + 2 * h) * h, "gemm_out": h * h * (nl - 1),
                    "gemm_ffn1": i * h * (nl - 1), "gemm_ffn2": h * i * (nl - 1)}
        kern = {}
        for cls, (ms, cnt) in prof.items():
            if cnt == 0:
                continue
            ent = {"ms_total": ms, "launches": cnt, "avg_us": 1e3 * ms / cnt}
            if cls in shape:
                fl = 2.0 * nk_total[cls] * tok_local / cnt                 # algorithmic FLOPs per launch (average)
                ent["flops_per_launch"] = fl
                ent["tflops"] = fl / (ms / cnt * 1e-3) / 1e12
            kern[cls] = ent
        dom = max((c for c in kern if c in shape), key=lambda c: kern[c]["ms_total"])
        # HBM-side bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
        # runs; bench.py cannot collect counters itself): the newest profiles/r*_final/pmc_traffic.json
        traffic, traffic_src = None, None
        for rdir in ("r6_final", "r5_final", "r4_final", "r3_final", "r2_final", "r1_final"):
            tpath = os.path.join(ROOT, "profiles", rdir, "pmc_traffic.json")
            if os.path.exists(tpath) and args.chunk_tokens == 65536 and model == "bert-base-uncased" and args.precision in ("bf16", "f16"):
                with open(tpath) as f:
                    tj = json.load(f)
                if dom in tj:
                    traffic = tj[dom]["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/{rdir}/pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE of a separate rocprofv3 --pmc run of this command, full 65536-token launch)"
                    break
        result["kernels"] = kern
        result["roofline"] = {
            "kernel": dom + ((" (gemm_tn_w8_kernel)" if os.environ.get("MANNER_HIP_GEMM_ASM", "8") == "8" else " (gemm_tn_x16_kernel)")
                             if args.precision != "fp32" else " (gemm_tn_big_kernel)"), "bound": "mfma",
            "achieved": kern[dom]["tflops"], "peak": peak,
            "unit": "TFLOP/s", "frac": kern[dom]["tflops"] / peak, "traffic": traffic, "traffic_source": traffic_src,
            "mfma_only_ceiling_tflops": 2040.0 if args.precision in ("bf16", "f16") else None,
            "avg_launch_us": kern[dom]["avg_us"], "flops_per_launch": kern[dom]["flops_per_launch"]}
    barrier()

    if rank == 0 and not args.no_collate:
        result["collate"] = guarded("collate leg", collate_leg, args, cfg, imp_all, pool_ids_np, pool_len, batches[-1], (n_steps - 1) * args.impressions, dev)
    del batches, table_bufs
    if not args.no_table:
        log("table mode")
        try:
            tab, par_scale, held = table_mode(args, conf, cfg, encs, fuse_w, (pool_ids, pool_mask, pool_len), rank, world, dev,
                                              scale_parity=not args.no_scale_parity)
        except Exception as exc:                       # a side leg must not take the headline line with it
            import traceback
            log("table mode FAILED: " + "".join(traceback.format_exception_only(type(exc), exc)).strip())
            tab, par_scale, held = {"error": repr(exc)}, None, None
        if rank == 0:
            result["table_mode"] = tab
        if isinstance(tab.get("mesh_exchange"), dict) and tab["mesh_exchange"].get("fatal"):
            # the experimental exchange did not complete on some rank: no GPU collective is safe any more.  Every figure of the line
            # was final before it started — print the line and leave without touching the process group (no re-exec: a plain exit)
            log("mesh exchange failed (" + str(tab["mesh_exchange"].get("error")) + "): printing the line and exiting")
            if rank == 0:
                emit(result, args)
            sys.stderr.flush()
            os._exit(0)
        if rank == 0 and par_scale is not None:
            def spread_weights_parity():
                log("at-scale parity on spread weights (std 0.05)")
                w2 = make_plm_weights(cfg, seed=44, std=0.05)
                enc2 = hip.HipEncoder(cfg, w2, precisions=("bf16", "f16", "fp32"), device=dev)
                dimp, labels = held
                sc = {}
                for prec in ("bf16", "f16", "fp32"):
                    t_ = enc2.encode_cls(pool_ids, pool_mask, precision=prec, host_lengths=pool_len, max_chunk_tokens=args.chunk_tokens)
                    sc[prec] = hip.score_late_fusion(t_, dimp["hist_idx"], dimp["hist_off"], dimp["cand_idx"], dimp["cand_off"])
                    del t_
                out = {m: ranking_agreement(sc[m], sc["fp32"], labels, dimp["cand_off"]) for m in ("bf16", "f16")}
                enc2.close()
                return out
            if K == 1:
                par_scale["spread_weights_std0.05"] = guarded("at-scale parity on spread weights", spread_weights_parity)
            result["parity_at_scale"] = par_scale
            # the parity mode's own throughput, in the driver-run line
            result["parity_mode"] = {"dtype": "fp32", "news_encoded_per_s": par_scale["parity_mode_news_per_s"],
                                     "what": "HIP f32-MFMA mode (<= 2e-5 from the reference on the goldens), table-mode encode of the pool",
                                     "f16x3_news_encoded_per_s": par_scale["other_mode_news_per_s"].get("f16x3"),
                                     "f16x3_what": "split-operand f16 GEMMs over f32 activations: within 1e-4 of the reference on the "
                                                   "goldens like the f32 mode; its ranking agreement with the f32 mode is in parity_at_scale"}
        del held
    if rank == 0 and world == 1 and not args.no_small_ops:
        log("small-kernel legs (pooler, dot, z-score, to_dense)")
        result["small_ops"] = guarded("small-kernel legs", small_ops_leg, dev)
    if rank == 0 and world == 1 and not args.no_cpu:
        log("CPU baseline (oracle) + parity on the bounded sample")
        nb = max(2, args.cpu_impressions // K)
        args.parity_impressions = max(nb, args.parity_impressions // K // (4 if cfg.layers * cfg.hidden > 12 * 768 else 1))   # bounded CPU time
        got = guarded("CPU baseline", cpu_baseline_and_parity, args, cfg, weight_sets, fuse_w, encs, imp_all, (pool_ids_np, pool_mask_np, pool_len), dev, nb)
        if isinstance(got, tuple):
            result["cpu_baseline"], result["parity"] = got
        else:
            result["cpu_baseline"] = dict(got, value=None, unit="candidates/s", cores=None, kind="port", sample=None)
    if rank == 0 and world == 1 and not args.no_train and cfg.head_dim == 64 and args.config == 1:
        log("training-step leg (train() mode encoder + scorer + SupCon + backward + AdamW)")
        # bf16 GEMM operands: f32's exponent range, so the step needs no loss scaling (f16 needs the caller's GradScaler, as the
        # reference's 16-mixed Lightning plugin provides — manner_amd/models/components/news_encoder.py train_precision)
        result["train_mode"] = guarded("training-step leg", train_leg, cfg, dev, args.precision if args.precision in ("bf16", "fp32") else "bf16")
    if rank == 0 and world == 1 and not args.no_dropin and cfg.head_dim == 64 and args.config == 1:
        log("drop-in leg (mirror classes under the unchanged CRModule.forward, B = 8 and 64, eval + train)")
        for e in encs:
            e.close()
        result["dropin"] = guarded("drop-in leg", dropin_leg, cfg, model, weight_sets[0], (pool_ids, pool_mask, pool_len), dev,
                                   args.precision if args.precision in ("f16", "bf16", "fp32") else "f16", only=args.dropin_only)
    if rank == 0:
        emit(result, args)
    if world > 1:
        torch.distributed.destroy_process_group()
    if FAILED_LEGS:
        log(f"side legs that FAILED (named in legs.errors of the line): {FAILED_LEGS}" + ("" if args.strict else "  [--strict turns this into exit status 3]"))
        if args.strict:
            sys.exit(3)


if __name__ == "__main__":
    main()
