#!/usr/bin/env python3
"""Per-class GEMM launch times of one encoder call under different tile orders of the persistent GEMMs (csrc/gemm.hip tile_walk:
MANNER_HIP_XCD_RANGES, MANNER_HIP_COL_GROUP — read per launch; profiles/r6_final/tile_rev_probe.txt came from a lab build with two more
switches, back-to-front tile / news orders, that changed nothing and were not kept), interleaved in ONE process so that box and clock state are shared.
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
# settings to compare: comma-separated, each "VAR=value+VAR=value" ("base" = the library's defaults); every variable named anywhere is
# unset for the settings that do not name it
MODES = os.environ.get("PROBE_ENVS", "MANNER_HIP_XCD_RANGES=0+MANNER_HIP_COL_GROUP=0,MANNER_HIP_COL_GROUP=0,base,MANNER_HIP_COL_GROUP=6,"
                       "MANNER_HIP_COL_GROUP=4,MANNER_HIP_COL_GROUP=3,MANNER_HIP_COL_GROUP=2").split(",")
VARS = sorted({kv.split("=")[0] for m in MODES if m != "base" for kv in m.split("+")})
ROUNDS, CALLS = int(os.environ.get("PROBE_ROUNDS", "4")), int(os.environ.get("PROBE_CALLS", "5"))


def set_mode(mode):
    for v in VARS:
        os.environ.pop(v, None)
    if mode != "base":
        for kv in mode.split("+"):
            k, v = kv.split("=")
            os.environ[k] = v


for tokens in [int(a) for a in sys.argv[1:]] or [65536, 24000]:
    n = int((cum <= tokens).sum())
    ids, mask = torch.from_numpy(ids_np[:n]).to(dev), torch.from_numpy(mask_np[:n]).to(dev)
    acc = {}
    for r in range(ROUNDS + 1):                                   # round 0 = warm-up
        for mode in MODES:
            set_mode(mode)
            enc.encode_cls(ids, mask, precision=prec, host_lengths=lens[:n])
            torch.cuda.synchronize()
            enc.profile(True)
            for _ in range(CALLS):
                enc.encode_cls(ids, mask, precision=prec, host_lengths=lens[:n])
            prof = enc.profile_read()
            enc.profile(False)
            if r == 0:
                continue
            a = acc.setdefault(mode, {})
            for k, (ms, cnt) in prof.items():
                if cnt and (k.startswith("gemm") or k == "attention"):
                    s = a.setdefault(k, [0.0, 0])
                    s[0] += ms; s[1] += cnt
    print(json.dumps({"tokens": int(cum[n - 1]), "news": n, "precision": prec}), flush=True)
    for mode, a in acc.items():
        row = {k: round(1e3 * ms / cnt, 1) for k, (ms, cnt) in a.items()}
        row["layer_us"] = round(sum(row.values()), 1)
        print("  " + "  ".join(f"{k} {v:7.1f}" for k, v in row.items()) + "   " + mode.replace("MANNER_HIP_", ""), flush=True)
