#!/usr/bin/env python3
"""Launches the HBM-side tail kernels at evaluation scale for a rocprofv3 pass (tools/profile_r2.sh):

* the fused late-fusion scorer over a MIND-LARGE-shaped news-embedding table — 161 013 rows x 768 f32 = 495 MB, larger
  than the 256 MiB Infinity Cache, so FETCH_SIZE is an HBM-side figure — on the first 131 072
  MIND-large-shaped impressions;
* additive-attention pooler, DotProduct, z-score fusion, to_dense at B = 4096, S = 50, D = 768.

Prints one JSON line with the algorithmic bytes of every launch so that tools/pmc_tail.py can set counter bytes
against them.  Not part of the product path."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manner_amd import hip  # noqa: E402
from manner_amd.synth import MIND_LARGE, synth_impressions  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
D = 768
n_news, n_imp = MIND_LARGE["n_news"], 131072
table = torch.randn((n_news, D), device=dev, generator=g)
imp = synth_impressions(n_imp, n_news, seed=43)
dimp = {k: torch.from_numpy(v).to(dev) for k, v in imp.items() if k != "labels"}
occ = int(imp["hist_off"][-1] + imp["cand_off"][-1])
uniq = int(np.unique(np.concatenate([imp["hist_idx"], imp["cand_idx"]])).size)
out = {"score_late_fusion": {"launches": 3, "table_MB": n_news * D * 4 / 1e6, "impressions": n_imp, "row_reads": occ,
                             "distinct_rows": uniq,
                             "algorithmic_bytes_per_launch": occ * (D * 4 + 4) + int(imp["cand_off"][-1]) * 4,
                             "compulsory_bytes_per_launch": uniq * D * 4 + occ * 4 + int(imp["cand_off"][-1]) * 4}}
for _ in range(3):
    sc = hip.score_late_fusion(table, dimp["hist_idx"], dimp["hist_off"], dimp["cand_idx"], dimp["cand_off"])
torch.cuda.synchronize()
# the same impressions over the IEEE-half copy of the table (247 MB: Infinity-Cache resident)
t16 = hip.table_to_f16(table, centre=True)
out["score_late_fusion_f16"] = {"launches": 3, "table_MB": n_news * D * 2 / 1e6, "impressions": n_imp, "row_reads": occ, "distinct_rows": uniq,
                                "algorithmic_bytes_per_launch": occ * (D * 2 + 4) + int(imp["cand_off"][-1]) * 4,
                                "compulsory_bytes_per_launch": uniq * D * 2 + occ * 4 + int(imp["cand_off"][-1]) * 4}
for _ in range(3):
    sc16 = hip.score_late_fusion(t16, dimp["hist_idx"], dimp["hist_off"], dimp["cand_idx"], dimp["cand_off"])
torch.cuda.synchronize()
out["score_late_fusion_f16"]["max_abs_score_diff_vs_f32_table"] = float((sc16 - sc).abs().max())
out["score_late_fusion_f16"]["score_abs_scale"] = float(sc.abs().max())
del table, t16
B, S, Q, C = 4096, 50, 200, 37
x = torch.randn((B, S, D), device=dev, generator=g)
W, bq, q = torch.randn((Q, D), device=dev, generator=g) * 0.05, torch.zeros(Q, device=dev), torch.randn(Q, device=dev, generator=g)
for _ in range(3):
    hip.additive_pool(x, W, bq, q)
out["additive_pool"] = {"launches": 3, "algorithmic_bytes_per_launch": x.numel() * 4 + B * D * 4 + (Q * D + 2 * Q) * 4,
                        "kernels": "round 4: pool_fused_kernel (x read ONCE, resident as scaled f16 hi/lo pairs) + pool_w_max / pool_pack_w (W, a few hundred KB)"}
for _ in range(3):
    hip.additive_pool(x, W, bq, q, strict=True)
out["additive_pool_strict"] = {"launches": 3, "algorithmic_bytes_per_launch": x.numel() * 4 + B * D * 4 + (Q * D + 2 * Q) * 4,
                               "kernels": "pool_logits_kernel (reads x once, f32 matrix pipe) + pool_apply_kernel (reads x again)"}
del x
user, cand = torch.randn((B, 1, D), device=dev, generator=g), torch.randn((B, C, D), device=dev, generator=g)
for _ in range(3):
    hip.dot(user, cand.permute(0, 2, 1))
out["dot"] = {"launches": 3, "algorithmic_bytes_per_launch": (cand.numel() + user.numel() + B * C) * 4}
off = dimp["cand_off"]
total = int(imp["cand_off"][-1])
planes = torch.randn((3, total), device=dev, generator=g)
for _ in range(3):
    hip.zscore_fuse(planes, [-0.3, 0.2], off)
out["zscore_fuse"] = {"launches": 3, "algorithmic_bytes_per_launch": 4 * total * 4 + off.numel() * 8}
nb = 2048
rows = int(imp["cand_off"][nb])
cv = cand.reshape(-1, D)[:rows].contiguous()
width = int(np.diff(imp["cand_off"][: nb + 1]).max())
for _ in range(3):
    hip.to_dense(cv, off[: nb + 1].contiguous(), width)
out["to_dense"] = {"launches": 3, "algorithmic_bytes_per_launch": cv.numel() * 4 + nb * width * D * 4}
del cand, cv, planes
torch.cuda.synchronize()
# SURVEY 8e phase C: one launch (planes in LDS) vs K + 2 launches, K = 3 MIND-large-shaped tables (3 x 495 MB), 131 072 impressions
from manner_amd import hotpath  # noqa: E402
tabs = [torch.randn((n_news, D), device=dev, generator=g) for _ in range(3)]
labels = torch.from_numpy(imp["labels"]).to(dev)
k_bytes = 3 * (occ * (D * 4 + 4)) + total * 4 + total * 4 + n_imp * (10 * 4 + 8)
for _ in range(3):
    fus = hotpath.score_impressions(tabs, dimp, weights=(-0.3, 0.2), labels=labels, k=10, fused=True)
out["phase_c_one_launch"] = {"launches": 3, "tables": 3, "impressions": n_imp, "algorithmic_bytes_per_launch": k_bytes,
                             "kernels": "score_fuse_rank_kernel"}
for _ in range(3):
    sep = hotpath.score_impressions(tabs, dimp, weights=(-0.3, 0.2), labels=labels, k=10, fused=False)
out["phase_c_separate"] = {"launches": 3, "tables": 3, "impressions": n_imp,
                           "algorithmic_bytes_per_launch": k_bytes + 2 * 3 * total * 4 + 2 * total * 4,
                           "kernels": "3 x score_late_fusion_rows_kernel + zscore_fuse_kernel + rank_ndcg_kernel (the planes and the fused scores round-trip HBM)"}
out["phase_c_one_launch"]["bit_identical_to_separate"] = bool(torch.equal(fus["scores"].nan_to_num(0.0), sep["scores"].nan_to_num(0.0)) and torch.equal(fus["topk"], sep["topk"]))
torch.cuda.synchronize()
hip.check_status(dev)
print(json.dumps(out))
