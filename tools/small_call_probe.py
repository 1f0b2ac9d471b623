#!/usr/bin/env python3
"""Latency of SMALL encoder calls (what the embedding cache leaves to encode: a handful of unseen news per batch).  Development aid.

    python tools/small_call_probe.py [n_news ...]     # wall time per encode_cls call, f16, bert-base, title+abstract tokens"""
import json
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manner_amd import hip  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

cfg = PRESETS["bert-base-uncased"]
dev = torch.device("cuda", 0)
enc = hip.HipEncoder(cfg, make_plm_weights(cfg, seed=1, std=0.02), precisions=("f16",), device=dev)
ids_np, mask_np = synth_news_tokens(4096, cfg, seed=3, max_len=96, profile="title_abstract")
out = {}
for n in [int(a) for a in sys.argv[1:]] or [4, 16, 64, 256, 512]:
    ids, mask = torch.from_numpy(ids_np[:n]).to(dev), torch.from_numpy(mask_np[:n]).to(dev)
    for _ in range(3):
        enc.encode_cls(ids, mask, precision="f16")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        enc.encode_cls(ids, mask, precision="f16")
    torch.cuda.synchronize()
    out[n] = {"ms_per_call": round(1e3 * (time.perf_counter() - t0) / 10, 3), "tokens": int(mask_np[:n].sum())}
print(json.dumps(out))
