#!/usr/bin/env python3
"""Per-class GEMM launch times of one encoder call under different tile orders of the persistent GEMMs (csrc/gemm.hip tile_walk:
MANNER_HIP_XCD_RANGES, MANNER_HIP_COL_GROUP — both read per launch), interleaved in ONE process so that box and clock state are shared.
Development aid of round 6.

    python tools/tile_order_probe.py [tokens ...]      # PROBE_PRECISION=f16|bf16, bert-base, title+abstract news"""
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

cfg = PRESETS[os.environ.get("PROBE_PRESET", "bert-base-uncased")]
dev = torch.device("cuda", 0)
prec = os.environ.get("PROBE_PRECISION", "f16")
enc = hip.HipEncoder(cfg, make_plm_weights(cfg, seed=1, std=0.02), precisions=(prec,), device=dev)
ids_np, mask_np = synth_news_tokens(4096, cfg, seed=3, max_len=96, profile="title_abstract")
lens = mask_np.sum(1)
cum = lens.cumsum()
os.environ["MANNER_HIP_GEMM_SMALL_TILES"] = "0"
MODES = [m.split(":") for m in os.environ.get("PROBE_MODES", "0:0,1:0,1:-,1:6,1:4,1:3,1:2").split(",")]   # ranges:col_group ("-" = chosen per launch)
ROUNDS, CALLS = int(os.environ.get("PROBE_ROUNDS", "4")), int(os.environ.get("PROBE_CALLS", "5"))


def set_mode(ranges, group):
    os.environ["MANNER_HIP_XCD_RANGES"] = ranges
    if group == "-":
        os.environ.pop("MANNER_HIP_COL_GROUP", None)
    else:
        os.environ["MANNER_HIP_COL_GROUP"] = group


for tokens in [int(a) for a in sys.argv[1:]] or [65536, 24000]:
    n = int((cum <= tokens).sum())
    ids, mask = torch.from_numpy(ids_np[:n]).to(dev), torch.from_numpy(mask_np[:n]).to(dev)
    acc = {}
    for r in range(ROUNDS + 1):                                   # round 0 = warm-up
        for ranges, group in MODES:
            set_mode(ranges, group)
            enc.encode_cls(ids, mask, precision=prec, host_lengths=lens[:n])
            torch.cuda.synchronize()
            enc.profile(True)
            for _ in range(CALLS):
                enc.encode_cls(ids, mask, precision=prec, host_lengths=lens[:n])
            prof = enc.profile_read()
            enc.profile(False)
            if r == 0:
                continue
            a = acc.setdefault(f"{ranges}:{group}", {})
            for k, (ms, cnt) in prof.items():
                if cnt and (k.startswith("gemm") or k == "attention"):
                    s = a.setdefault(k, [0.0, 0])
                    s[0] += ms; s[1] += cnt
    print(json.dumps({"tokens": int(cum[n - 1]), "news": n, "precision": prec}), flush=True)
    for mode, a in acc.items():
        row = {k: round(1e3 * ms / cnt, 1) for k, (ms, cnt) in a.items()}
        row["layer_us"] = round(sum(row.values()), 1)
        print(f"  ranges:group {mode:>5}  " + "  ".join(f"{k} {v:7.1f}" for k, v in row.items()), flush=True)
