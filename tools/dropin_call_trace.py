#!/usr/bin/env python3
"""One eval() forward of the module mirror, repeated: what does a call launch besides the encoder's own kernels?  (rocprofv3 --kernel-trace
--stats -- python3 tools/dropin_call_trace.py; development aid for the drop-in path's launch overhead.)"""
import os, sys, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manner_amd.config import PRESETS
from manner_amd.synth import synth_news_tokens
from manner_amd.models.components.news_encoder import MannerTextEncoder
cfg = PRESETS["bert-base-uncased"]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    enc = MannerTextEncoder("bert-base-uncased", [], 0.2).to("cuda:0").eval()
ids, mask = synth_news_tokens(200, cfg, seed=1, max_len=96, profile="title_abstract")
x = {"input_ids": torch.from_numpy(ids).to("cuda:0"), "attention_mask": torch.from_numpy(mask).to("cuda:0")}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    enc(x); enc(x)
    torch.cuda.synchronize()
    for _ in range(n):
        enc(x)
    torch.cuda.synchronize()
print("calls:", n + 2)
