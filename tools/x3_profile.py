#!/usr/bin/env python3
"""Per-kernel-class HIP-event times of one table-mode encode of 65 536-token chunks in a given arithmetic mode (default f16x3),
next to f16 — where the split-operand parity mode spends its time.  Development aid, not product code.

    python tools/x3_profile.py [f16x3|bf16x3|fp32] [n_news]"""
import json
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manner_amd import hip  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
cfg = PRESETS["bert-base-uncased"]
dev = torch.device("cuda", 0)
ids, mask = synth_news_tokens(n, cfg, seed=42, max_len=96, profile="title_abstract")
lens = mask.sum(1)                                  # host lengths: full 65 536-token chunks, as in bench.py's table mode
ids, mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
w = make_plm_weights(cfg, seed=42, std=0.02, with_pooler=False)
os.environ["MANNER_HIP_STREAMS"] = "1"
out = {}
for prec in ("f16", mode):
    enc = hip.HipEncoder(cfg, w, precisions=(prec,), device=dev)
    enc.encode_cls(ids, mask, precision=prec, host_lengths=lens)
    torch.cuda.synchronize()
    enc.profile(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    enc.encode_cls(ids, mask, precision=prec, host_lengths=lens)
    ev[1].record()
    torch.cuda.synchronize()
    prof = enc.profile_read()
    enc.profile(False)
    out[prec] = {"wall_ms": ev[0].elapsed_time(ev[1]), "news_per_s": n / ev[0].elapsed_time(ev[1]) * 1e3,
                 "classes": {k: {"ms": round(v[0], 3), "launches": v[1]} for k, v in prof.items() if v[1]}}
    enc.close()
print(json.dumps(out, indent=1))
