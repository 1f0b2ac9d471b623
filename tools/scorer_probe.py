#!/usr/bin/env python3
"""Times the fused late-fusion scorer (f32 table, centred f16 table) with HIP events at the MIND-large and MIND-small table
shapes and prints the scores' checksum, so that two builds / switches (MANNER_HIP_SCORER_GENERIC=1: the column-block kernel)
can be compared bit for bit.  Development aid, not product code.

    python tools/scorer_probe.py [large|small] [out.pt]"""
import json
import sys
import os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manner_amd import hip  # noqa: E402
from manner_amd.synth import MIND_LARGE, MIND_SMALL, synth_impressions  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "large"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
D = 768
n_news = (MIND_LARGE if which == "large" else MIND_SMALL)["n_news"]
n_imp = 131072 if which == "large" else 73152
table = torch.randn((n_news, D), device=dev, generator=g)
imp = synth_impressions(n_imp, n_news, seed=43)
d = {k: torch.from_numpy(v).to(dev) for k, v in imp.items() if k != "labels"}
occ = int(imp["hist_off"][-1] + imp["cand_off"][-1])


def timed(fn, n=10):
    for _ in range(2):
        out = fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        out = fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return out, float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]))


sc, ms = timed(lambda: hip.score_late_fusion(table, d["hist_idx"], d["hist_off"], d["cand_idx"], d["cand_off"]))
t16 = hip.table_to_f16(table, centre=True)
sc16, ms16 = timed(lambda: hip.score_late_fusion(t16, d["hist_idx"], d["hist_off"], d["cand_idx"], d["cand_off"]))
res = {"shape": which, "table_MB": n_news * D * 4 / 1e6, "impressions": n_imp, "row_reads": occ,
       "f32_ms": ms, "f32_occurrence_TBps": occ * D * 4 / ms / 1e9, "f16_ms": ms16, "f16_occurrence_TBps": occ * D * 2 / ms16 / 1e9,
       "f32_sum_bits": int(sc.view(torch.int32).to(torch.int64).sum()), "f16_sum_bits": int(sc16.view(torch.int32).to(torch.int64).sum()),
       "generic": os.environ.get("MANNER_HIP_SCORER_GENERIC", "0")}
if len(sys.argv) > 2:
    torch.save({"f32": sc.cpu(), "f16": sc16.cpu()}, sys.argv[2])
hip.check_status(dev)
print(json.dumps(res))
