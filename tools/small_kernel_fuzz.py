#!/usr/bin/env python3
"""One-off soak (round 4): random small encoder calls through the 128x128 kernel (default) and the persistent 256x256 kernel
(MANNER_HIP_GEMM_SMALL_TILES=0) must give the same bits.  python tools/small_kernel_fuzz.py [n_cases]"""
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
g = np.random.default_rng(7)
dev = "cuda:0"
bad = 0
encs = {}
for arch in ("bert-base-uncased", "mini-roberta-large"):
    cfg = PRESETS[arch] if arch.startswith("mini") else dataclasses.replace(PRESETS[arch], layers=2)
    encs[arch] = (cfg, hip.HipEncoder(cfg, make_plm_weights(cfg, seed=5, std=0.03), precisions=("f16", "bf16"), device=dev))
for case in range(n_cases):
    arch = ("bert-base-uncased", "mini-roberta-large")[case % 2]
    cfg, enc = encs[arch]
    n = int(g.integers(1, 420))
    ml = int(g.integers(4, 97))
    prec = ("f16", "bf16")[int(g.integers(0, 2))]
    ids_np, mask_np = synth_news_tokens(n, cfg, seed=1000 + case, max_len=ml)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    os.environ["MANNER_HIP_GEMM_SMALL_TILES"] = "0"
    a = enc.encode_cls(ids, mask, precision=prec)
    h = enc.encode_hidden(ids, mask, 1, precision=prec)
    del os.environ["MANNER_HIP_GEMM_SMALL_TILES"]
    b = enc.encode_cls(ids, mask, precision=prec)
    k = enc.encode_hidden(ids, mask, 1, precision=prec)
    ok = torch.equal(a, b) and torch.equal(h, k) and bool(torch.isfinite(b).all())
    bad += not ok
    print(case, arch, n, ml, prec, int(mask_np.sum()), "ok" if ok else "MISMATCH", flush=True)
for _, e in encs.values():
    e.status()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
