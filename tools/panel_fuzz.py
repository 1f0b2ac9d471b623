#!/usr/bin/env python3
"""One-off soak (round 5): random encoder calls with the persistent GEMM's row panels pinned to 256 rows, pinned to 192 rows and chosen by
the kernel (csrc/gemm.hip panel_rows) must give the same bits — [CLS] rows and layer-1 hidden states, f16 / bf16 / f16x3, with and without
host lengths (exact and loose row bounds: the 192-row tiles' clamped DMA pieces).  python tools/panel_fuzz.py [n_cases]"""
import dataclasses
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manner_amd import hip  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = np.random.default_rng(11)
dev = "cuda:0"
os.environ["MANNER_HIP_GEMM_SMALL_TILES"] = "0"          # the persistent kernel for every shape
encs = {}
for arch in ("bert-base-uncased", "mini-roberta-large"):
    cfg = PRESETS[arch] if arch.startswith("mini") else dataclasses.replace(PRESETS[arch], layers=2)
    encs[arch] = (cfg, hip.HipEncoder(cfg, make_plm_weights(cfg, seed=6, std=0.03), precisions=("f16", "bf16", "f16x3"), device=dev))
bad = 0
for case in range(n_cases):
    arch = ("bert-base-uncased", "mini-roberta-large")[case % 2]
    cfg, enc = encs[arch]
    n = int(g.integers(1, 700))
    ml = int(g.integers(4, 97))
    prec = ("f16", "bf16", "f16x3")[int(g.integers(0, 3))]
    profile = ("title", "title_abstract")[int(g.integers(0, 2))]          # title_abstract: ~75 tokens per news, up to ~50 k tokens per call
    ids_np, mask_np = synth_news_tokens(n, cfg, seed=2000 + case, max_len=ml, profile=profile)
    lens = mask_np.sum(1) if g.integers(0, 2) else None
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    got = {}
    for mode in ("256", "192", None):
        if mode is None:
            os.environ.pop("MANNER_HIP_GEMM_PANEL", None)
        else:
            os.environ["MANNER_HIP_GEMM_PANEL"] = mode
        got[mode] = (enc.encode_cls(ids, mask, precision=prec, host_lengths=lens), enc.encode_hidden(ids, mask, 1, precision=prec, host_lengths=lens))
    ok = all(torch.equal(got[m][0], got["256"][0]) and torch.equal(got[m][1], got["256"][1]) for m in ("192", None)) and bool(torch.isfinite(got["256"][0]).all())
    bad += not ok
    print(case, arch, n, ml, profile, prec, int(mask_np.sum()), "lengths" if lens is not None else "bound", "ok" if ok else "MISMATCH", flush=True)
os.environ.pop("MANNER_HIP_GEMM_PANEL", None)
for _, e in encs.values():
    e.status()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
