#!/usr/bin/env python3
"""Headline bench: candidate news encoded+scored per second (BASELINE.json metric), configs[1]:
CR-Module, MIND-small shape, bert-base-uncased architecture, bf16, HIP NewsEncoder + scorer.

One STEP = one pass of the hot path over one batch of synthetic impressions in reference-faithful
mode R: EVERY history and candidate occurrence of the batch is encoded by the PLM
(as reference cr_module.py:107,113 does — nothing is cached or deduplicated), then late-fusion
mean + dot product per candidate, stable top-10 ranking and nDCG@10.  Token tensors, index lists
and labels of every step are resident in HBM before the timed region; each step uses different
impressions.  value = candidates scored by all ranks / max-over-ranks wall time of the K steps.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: impressions are independent, so ranks take disjoint impression batches (weak scaling,
no data-path collective in mode R; barrier + MAX over ranks for the clock).  The table
architecture with the RCCL all-gather of the news-embedding table is measured by --table.
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
from manner_amd.synth import (MIND_SMALL, shard_range, synth_impression_blocks, synth_impressions,  # noqa: E402
                              synth_news_tokens)
from manner_amd.weights import make_plm_weights  # noqa: E402

BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md chip table
F32_PEAK_TFLOPS = 157.3        # f32-input MFMA
HBM_PEAK_GBS = 8000.0


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--impressions", type=int, default=256, help="impressions per step per GPU")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3"])
    p.add_argument("--profile", default="title_abstract", choices=["title", "title_abstract"],
                   help="token-length profile of the news pool (SURVEY.md §8d)")
    p.add_argument("--model", default="bert-base-uncased")
    p.add_argument("--std", type=float, default=0.02, help="std of the seeded PLM weight matrices")
    p.add_argument("--chunk-tokens", type=int, default=65536)
    p.add_argument("--cpu-impressions", type=int, default=8, help="impressions of the CPU-baseline sample")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-kernel-profile", action="store_true")
    p.add_argument("--no-collate", action="store_true", help="skip the device-side collate leg")
    p.add_argument("--no-table", action="store_true", help="skip the table-mode (encode pool once + all-gather) leg")
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
        self.flops = None


def run_step(enc, b, precision, chunk_tokens, table_buf):
    table = enc.encode_cls(b.ids, b.mask, precision=precision, host_lengths=b.lens, max_chunk_tokens=chunk_tokens,
                           out=table_buf[: b.ids.shape[0]])
    scores = hip.score_late_fusion(table, b.hist_idx, b.hist_off, b.cand_idx, b.cand_off, total_cand=b.n_cand)
    topk, ndcg = hip.rank_ndcg(scores, b.labels, b.cand_off, 10)
    return scores, topk, ndcg


def cpu_baseline_and_parity(args, cfg, weights, enc, imp, pool, dev, nb=None):
    """Oracle (CPU port of the reference path) on a bounded sample + parity of the HIP path on it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import manner_oracle as O
    pool_ids, pool_mask, pool_len = pool
    nb = nb or args.cpu_impressions
    ho, co = imp["hist_off"][: nb + 1], imp["cand_off"][: nb + 1]
    hi, ci = imp["hist_idx"][: ho[-1]], imp["cand_idx"][: co[-1]]
    # the GPU box grants a CPU share of 16 cores per GPU; os.cpu_count() reports the whole host
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    ref = O.reference_faithful_scores(pool_ids, pool_mask, hi.astype(np.int64), ho.tolist(), ci.astype(np.int64),
                                      co.tolist(), weights, cfg, chunk=64)
    cpu_s = time.perf_counter() - t0
    labels = torch.from_numpy(imp["labels"][: co[-1]])
    ref_ndcg, _ = O.ndcg_at_k(ref, labels, co.tolist(), 10)
    ref_top = O.topk_indices(ref, co.tolist(), 10)
    b = StepBatch(imp, 0, nb, torch.from_numpy(pool_ids).to(dev), torch.from_numpy(pool_mask).to(dev), pool_len, dev)
    buf = torch.empty((b.ids.shape[0], cfg.hidden), dtype=torch.float32, device=dev)
    par = {}
    for prec in ("fp32", "bf16"):
        scores, topk, ndcg = run_step(enc, b, prec, args.chunk_tokens, buf)
        top = [[v for v in row if v >= 0] for row in topk.cpu().tolist()]
        agree = float(np.mean([t == r for t, r in zip(top, ref_top)]))
        par[prec] = {"score_max_abs_err": float((scores.cpu() - ref).abs().max()),
                     "top10_identical_frac": agree,
                     "ndcg10_delta": float(abs(ndcg.double().mean().item() - ref_ndcg))}
    par["score_abs_scale"] = float(ref.abs().max())
    cpu = {"value": float(co[-1] / cpu_s), "unit": "candidates/s", "cores": cores, "kind": "port",
           "sample": f"oracle/manner_oracle.py mode R on the first {nb} impressions "
                     f"({int(ho[-1] + co[-1])} news encodes, {cpu_s:.1f} s, torch {torch.__version__} CPU fp32)"}
    return cpu, par


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
    lp = b.ids.shape[1]
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
    # the text kernel alone (no host-side offset slicing, no allocation of the small tensors)
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


def table_mode(args, cfg, enc, pool, rank, world, dev):
    """Mode T: every rank encodes its FLOP-balanced shard of the unique-news pool once,
    one RCCL all-gather assembles the [N_news, D] table on every rank, then each rank scores its block of
    MIND-small-shaped impressions (73 152 in total) by index.  Reported beside the headline, never as it."""
    from manner_amd import distributed as D
    pool_ids, pool_mask, pool_len = pool
    n_news = pool_ids.shape[0]
    shards = D.balanced_news_shards(pool_len, world, cfg.flops_per_news)
    lo, hi = shards[rank]
    imp = synth_impressions(MIND_SMALL["n_impressions"], n_news, seed=43)
    a, b = shard_range(MIND_SMALL["n_impressions"], rank, world)
    ho, co = imp["hist_off"], imp["cand_off"]
    dimp = {"hist_idx": torch.from_numpy(imp["hist_idx"][ho[a]:ho[b]]).to(dev), "hist_off": torch.from_numpy(ho[a:b + 1] - ho[a]).to(dev),
            "cand_idx": torch.from_numpy(imp["cand_idx"][co[a]:co[b]]).to(dev), "cand_off": torch.from_numpy(co[a:b + 1] - co[a]).to(dev)}
    labels = torch.from_numpy(imp["labels"][co[a]:co[b]]).to(dev)
    local = torch.empty((hi - lo, cfg.hidden), dtype=torch.float32, device=dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    times = {}
    for it in range(2):                                 # pass 0 warms up (workspace, RCCL channels)
        sync(); t0 = time.perf_counter()
        enc.encode_cls(pool_ids[lo:hi], pool_mask[lo:hi], precision=args.precision, host_lengths=pool_len[lo:hi],
                       max_chunk_tokens=args.chunk_tokens, out=local)
        sync(); t1 = time.perf_counter()
        table = D.all_gather_table(local, shards)
        sync(); t2 = time.perf_counter()
        res = hotpath.score_impressions([table], dimp, labels=labels, k=10)
        sync(); t3 = time.perf_counter()
        times = {"encode_s": t1 - t0, "allgather_s": t2 - t1, "score_s": t3 - t2, "total_s": t3 - t0}
    # epoch-end metrics of this rank's block on the device: timed separately, not part of total_s
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
    sc_r, off_r = res["scores"], dimp["cand_off"]
    n_c = int(sc_r.numel())
    metrics_ms = {"rank_ndcg_mrr": timed_ms(lambda: hip.rank_ndcg(sc_r, labels, off_r, 10, with_mrr=True)),
                  "auc": timed_ms(lambda: hip.auc(sc_r, labels)),
                  "eval_loss_supcon": timed_ms(lambda: hip.eval_loss(sc_r, labels, off_r, supcon=True, temperature=0.36, reduce=False))}
    st = torch.tensor([times["encode_s"], times["allgather_s"], times["score_s"], times["total_s"]], dtype=torch.float64, device=dev)
    nd = torch.tensor([float(res["ndcg"].double().sum()), float(b - a)], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(st, op=torch.distributed.ReduceOp.MAX)
        D.allreduce_metric_sums(nd)
    enc_s, ag_s, sc_s, tot_s = st.tolist()
    total_c = int(co[-1])
    hbm_bytes = float(ho[-1] + co[-1]) * (cfg.hidden * 4 + 4) + float(co[-1]) * 4
    return {"what": "unique-news table: encode shard -> all-gather -> score all 73152 MIND-small-shaped impressions",
            "candidates_per_s": total_c / tot_s, "news_encoded_per_s": n_news / enc_s,
            "scorer_pairs_per_s": total_c / sc_s, "scorer_GBps_algorithmic": hbm_bytes / sc_s / 1e9 / world,
            "scorer_frac_of_8TBps": hbm_bytes / sc_s / 1e9 / world / HBM_PEAK_GBS,
            "allgather_ms": 1e3 * ag_s, "allgather_bytes_per_rank": (n_news - (hi - lo)) * cfg.hidden * 4 if world > 1 else 0,
            # bytes every rank RECEIVES over xGMI / time, against 7 links x 153 GB/s per GPU
            "allgather_GBps_per_rank": ((n_news - (hi - lo)) * cfg.hidden * 4 / ag_s / 1e9) if world > 1 and ag_s > 0 else None,
            "allgather_frac_of_xgmi": ((n_news - (hi - lo)) * cfg.hidden * 4 / ag_s / 1e9 / (7 * 153.0)) if world > 1 and ag_s > 0 else None,
            "encode_ms": 1e3 * enc_s, "score_ms": 1e3 * sc_s, "ndcg10": nd[0].item() / nd[1].item(),
            "metrics_ms_rank0": {**metrics_ms, "candidates": n_c,
                                 "auc_Mpairs_per_s": n_c / metrics_ms["auc"] / 1e3, "rank_Mpairs_per_s": n_c / metrics_ms["rank_ndcg_mrr"] / 1e3}}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run (see docstring)")
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

    cfg = PRESETS[args.model]
    log("generating seeded weights")
    weights = make_plm_weights(cfg, seed=42, std=args.std)
    log("packing weights into the HIP encoder")
    enc = hip.HipEncoder(cfg, weights, precisions=("bf16", "fp32") + (("bf16x3",) if args.precision == "bf16x3" else ()), device=dev)
    n_news = MIND_SMALL["n_news"]
    log("synthesising news pool + impressions")
    pool_ids_np, pool_mask_np = synth_news_tokens(n_news, cfg, seed=42, max_len=96, profile=args.profile)
    pool_len = pool_mask_np.sum(1)
    pool_ids, pool_mask = torch.from_numpy(pool_ids_np).to(dev), torch.from_numpy(pool_mask_np).to(dev)
    n_steps = args.warmup + args.steps
    # one independent 256-impression draw per (rank, step): step s of rank r is the same batch for every --steps and
    # --gpus, so the figure does not depend on how many steps were asked for beyond averaging over more batches
    imp_all = synth_impression_blocks([rank * 1_000_000 + s for s in range(n_steps)], args.impressions, n_news, seed=42)
    lo_r = 0
    batches = [StepBatch(imp_all, lo_r + s * args.impressions, lo_r + (s + 1) * args.impressions, pool_ids, pool_mask,
                         pool_len, dev) for s in range(n_steps)]
    max_news = max(b.ids.shape[0] for b in batches)
    table_buf = torch.empty((max_news, cfg.hidden), dtype=torch.float32, device=dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    log(f"{n_steps} step batches resident ({batches[0].ids.shape[0]} news, {batches[0].tokens} tokens in step 0); warm-up")
    for b in batches[: args.warmup]:
        run_step(enc, b, args.precision, args.chunk_tokens, table_buf)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for b in batches[args.warmup:]:
        last = run_step(enc, b, args.precision, args.chunk_tokens, table_buf)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
    enc.status()

    timed = batches[args.warmup:]
    cands = float(sum(b.n_cand for b in timed))
    news = float(sum(b.n_hist + b.n_cand for b in timed))
    tokens = float(sum(b.tokens for b in timed))
    enc_flops = float(sum(cfg.flops_per_news(int(l)) for b in timed for l in b.lens))
    h_, i_ = cfg.hidden, cfg.intermediate
    per_layer = lambda l: 8 * l * h_ * h_ + 4 * l * h_ * i_ + 4 * l * l * h_      # noqa: E731
    exec_flops = float(sum((cfg.layers - 1) * per_layer(int(l)) + 4 * int(l) * h_ * h_ + 4 * int(l) * h_
                           + 4 * h_ * h_ + 4 * h_ * i_ for b in timed for l in b.lens))
    stats = torch.tensor([elapsed, cands, news, tokens, enc_flops, exec_flops], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.SUM)
        stats[0] = mx[0]
    elapsed_max, cands_all, news_all, tokens_all, flops_all, exec_all = stats.tolist()

    result = None
    if rank == 0:
        peak = BF16_PEAK_TFLOPS if args.precision == "bf16" else F32_PEAK_TFLOPS
        result = {
            "metric": "candidate news encoded+scored/sec", "value": cands_all / elapsed_max, "unit": "candidates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "configs[1]: CR-Module late fusion, MIND-small shape (65238-news pool), "
                                   f"{args.model} architecture, mode R (every history+candidate occurrence encoded)",
                       "impressions_per_step_per_gpu": args.impressions, "length_profile": args.profile,
                       "seeded_weights_std": args.std, "parallelism": f"dp{world} (impressions sharded, no collective)"},
            "news_encoded_per_s": news_all / elapsed_max, "tokens_per_s": tokens_all / elapsed_max,
            # algorithmic = SURVEY.md §8d F(L) for every news (what the reference computes); executed
            # excludes the last layer's non-[CLS] rows, which the HIP path prunes (not credited as
            # utilisation): last layer costs 4LH^2 (K,V) instead of 8LH^2 + 4LHI + 4L^2 H
            "encoder_tflops_algorithmic": flops_all / elapsed_max / 1e12,
            "encoder_tflops": exec_all / elapsed_max / 1e12,
            "encoder_mfma_frac": exec_all / elapsed_max / 1e12 / (peak * world),
            "ndcg10_last_step": float(last[2].double().mean().item()),
        }

    # per-kernel roofline: same steps again with every launch bracketed by HIP events on the launch stream
    if rank == 0 and not args.no_kernel_profile:
        log("per-kernel HIP-event pass")
        enc.profile(True)
        for b in timed:
            run_step(enc, b, args.precision, args.chunk_tokens, table_buf)
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
                n_, k_ = shape[cls]
                fl = 2.0 * nk_total[cls] * tok_local / cnt                 # algorithmic FLOPs per launch (average)
                ent["flops_per_launch"] = fl
                ent["tflops"] = fl / (ms / cnt * 1e-3) / 1e12
            kern[cls] = ent
        dom = max((c for c in kern if c in shape), key=lambda c: kern[c]["ms_total"])
        # HBM-side bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
        # runs; bench.py cannot collect counters itself): profiles/r1_final/pmc_traffic.json
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r1_final", "pmc_traffic.json")
        if os.path.exists(tpath) and args.chunk_tokens == 65536 and args.model == "bert-base-uncased":
            with open(tpath) as f:
                tj = json.load(f)
            if dom in tj:
                traffic = tj[dom]["hbm_bytes_per_launch"]
                traffic_src = "profiles/r1_final/pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE, full 65536-token launch)"
        result["kernels"] = kern
        result["roofline"] = {
            "kernel": dom + " (gemm_tn_x16_kernel)", "bound": "mfma", "achieved": kern[dom]["tflops"], "peak": peak,
            "unit": "TFLOP/s", "frac": kern[dom]["tflops"] / peak, "traffic": traffic, "traffic_source": traffic_src,
            "mfma_only_ceiling_tflops": 2040.0 if args.precision == "bf16" else None,
            "avg_launch_us": kern[dom]["avg_us"], "flops_per_launch": kern[dom]["flops_per_launch"]}
    barrier()

    if rank == 0 and not args.no_collate:
        result["collate"] = collate_leg(args, cfg, imp_all, pool_ids_np, pool_len, batches[-1], lo_r + (n_steps - 1) * args.impressions, dev)
    if not args.no_table:
        tab = table_mode(args, cfg, enc, (pool_ids, pool_mask, pool_len), rank, world, dev)
        if rank == 0:
            result["table_mode"] = tab
    if rank == 0 and world == 1 and not args.no_cpu:
        log("CPU baseline (oracle) + parity on the bounded sample")
        cpu, par = cpu_baseline_and_parity(args, cfg, weights, enc, imp_all, (pool_ids_np, pool_mask_np, pool_len), dev)
        result["cpu_baseline"] = cpu
        result["parity"] = par
        # the same check on "trained-like" weights (matrices ~N(0, 0.05^2)): with HF-init weights the CLS
        # vectors of different news are almost collinear, so bf16 rounding reorders near-tied scores
        log("parity on spread weights (std 0.05)")
        w2 = make_plm_weights(cfg, seed=44, std=0.05)
        enc2 = hip.HipEncoder(cfg, w2, precisions=("bf16", "fp32"), device=dev)
        _, par2 = cpu_baseline_and_parity(args, cfg, w2, enc2, imp_all, (pool_ids_np, pool_mask_np, pool_len), dev, nb=4)
        enc2.close()
        result["parity_spread_weights"] = par2
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
