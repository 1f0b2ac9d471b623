#!/usr/bin/env python3
"""bench.py's training-step leg alone, for a rocprofv3 pass that sees nothing else.  Development aid.

    python tools/train_probe.py [bf16|f16|fp32]"""
import json
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
if os.environ.get("TRAIN_PROBE_NODROP"):            # what the counter-based dropout bits cost: the same step with every p = 0
    from manner_amd import train as _T
    _orig = _T.encode_train

    def _nodrop(*a, **k):
        k.update(p_hidden=0.0, p_attn=0.0, p_out=0.0)
        return _orig(*a, **k)
    _T.encode_train = _nodrop
only = os.environ.get("TRAIN_PROBE_VARIANT")          # e.g. reference_default_embeddings_trainable: one variant, for a profile of it alone
out = bench.train_leg(PRESETS["bert-base-uncased"], torch.device("cuda", 0), prec, only=(only,) if only else None,
                      steps=int(os.environ.get("TRAIN_PROBE_STEPS", "5")))
print(json.dumps({k: ({kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "ms_per_step_unsynchronised_loop", "step_ms_each", "peak_GB")} if isinstance(v, dict) else v)
                  for k, v in out.items() if k != "what"}))
