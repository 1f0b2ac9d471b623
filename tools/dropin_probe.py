#!/usr/bin/env python3
"""bench.py's drop-in leg alone (one case), for a rocprofv3 pass that sees nothing else.  Development aid.

    python tools/dropin_probe.py 8:train"""
import json
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "8:train"
model = "bert-base-uncased"
cfg = PRESETS[model]
dev = torch.device("cuda", 0)
w = make_plm_weights(cfg, seed=42, std=0.02)
ids, mask = synth_news_tokens(65238, cfg, seed=42, max_len=96, profile="title_abstract")
pool = (torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev), mask.sum(1))
print(json.dumps(bench.dropin_leg(cfg, model, w, pool, dev, "f16", only=case), indent=0)[-2600:])
