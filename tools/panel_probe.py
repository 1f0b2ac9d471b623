#!/usr/bin/env python3
"""Per-class GEMM launch times of ONE encoder call at a given token count with 256-row, 192-row and auto-selected panels
(MANNER_HIP_GEMM_PANEL; csrc/gemm.hip panel_rows).  Development aid for the wave-quantisation work of round 5.

    python tools/panel_probe.py [tokens ...]      # f16, bert-base, 96-token news; HIP events per launch class (encoder profile)"""
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

cfg = PRESETS["bert-base-uncased"]
dev = torch.device("cuda", 0)
prec = os.environ.get("PROBE_PRECISION", "f16")
enc = hip.HipEncoder(cfg, make_plm_weights(cfg, seed=1, std=0.02), precisions=(prec,), device=dev)
ids_np, mask_np = synth_news_tokens(4096, cfg, seed=3, max_len=96, profile="title_abstract")
lens = mask_np.sum(1)
cum = lens.cumsum()
os.environ["MANNER_HIP_GEMM_SMALL_TILES"] = "0"
out = {}
for tokens in [int(a) for a in sys.argv[1:]] or [10000, 13000, 15616, 19000, 25000, 31000, 40000]:
    n = int((cum <= tokens).sum())
    ids, mask = torch.from_numpy(ids_np[:n]).to(dev), torch.from_numpy(mask_np[:n]).to(dev)
    row = {"news": n, "tokens": int(cum[n - 1])}
    for mode in ("256", "192", "auto"):
        if mode == "auto":
            os.environ.pop("MANNER_HIP_GEMM_PANEL", None)
        else:
            os.environ["MANNER_HIP_GEMM_PANEL"] = mode
        for _ in range(3):
            enc.encode_cls(ids, mask, precision=prec, host_lengths=lens[:n])
        torch.cuda.synchronize()
        enc.profile(True)
        for _ in range(10):
            enc.encode_cls(ids, mask, precision=prec, host_lengths=lens[:n])
        prof = enc.profile_read()
        enc.profile(False)
        row[mode] = {k: round(1e3 * ms / cnt, 1) for k, (ms, cnt) in prof.items() if cnt and k.startswith("gemm")}
        row[mode]["all_ms_per_call"] = round(sum(ms for ms, _ in prof.values()) / 10, 3)
    out[tokens] = row
    print(json.dumps({tokens: row}), flush=True)
