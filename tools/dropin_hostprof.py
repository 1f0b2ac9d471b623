#!/usr/bin/env python3
"""Where the HOST time of the B = 8 drop-in eval step goes: cProfile around bench.dropin_leg's 8:eval case (development aid, round 6).
    python tools/dropin_hostprof.py [top_n]"""
import cProfile
import io
import os
import pstats
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

model = "bert-base-uncased"
cfg = PRESETS[model]
dev = torch.device("cuda", 0)
w = make_plm_weights(cfg, seed=42, std=0.02)
ids, mask = synth_news_tokens(65238, cfg, seed=42, max_len=96, profile="title_abstract")
pool = (torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), mask.sum(1))
bench.dropin_leg(cfg, model, w, pool, dev, "f16", only="8:eval")          # warm: handle built, kernels loaded
pr = cProfile.Profile()
pr.enable()
out = bench.dropin_leg(cfg, model, w, pool, dev, "f16", only="8:eval")
pr.disable()
print({k: v for k, v in out.get("B8_eval", {}).items() if k in ("ms_per_step", "enqueue_ms", "steps")})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
print(s.getvalue()[:9000])
