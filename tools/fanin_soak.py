#!/usr/bin/env python3
"""Soak of the in-launch row-statistics finalize (csrc/gemm.hip nres_fan_in): which workgroup reduces a row panel depends on timing,
the result must not.  Random call sizes (one tile per CU ... several chunks on two streams), both 16-bit types, both kernels, both
panel heights, for `seconds`: every output is held to the bits of the separate finalize launch (MANNER_HIP_DLN_FANIN=0).
    python tools/fanin_soak.py [seconds=240]"""
import dataclasses
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manner_amd import hip  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
dev = torch.device("cuda", 0)
cfg = dataclasses.replace(PRESETS["bert-base-uncased"], layers=4)
w = make_plm_weights(cfg, seed=5, std=0.03)
ids_np, mask_np = synth_news_tokens(4000, cfg, seed=5, max_len=96, profile="title_abstract")
lens_all = mask_np.sum(1)
ids_all, mask_all = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
enc = hip.HipEncoder(cfg, w, precisions=("f16", "bf16"), device=dev)
os.environ["MANNER_HIP_GEMM_SMALL_TILES"] = "0"
g = np.random.default_rng(11)
t0, calls, cases = time.time(), 0, 0
while time.time() - t0 < seconds:
    n = int(g.choice([int(g.integers(40, 400)), int(g.integers(400, 1200)), int(g.integers(1200, 4000))]))
    a = int(g.integers(0, 4000 - n + 1))
    ids, mask, lens = ids_all[a:a + n], mask_all[a:a + n], lens_all[a:a + n]
    prec = "f16" if g.integers(2) else "bf16"
    os.environ["MANNER_HIP_GEMM_ASM"] = "8" if g.integers(4) else "0"
    panel = g.integers(3)
    if panel == 0:
        os.environ.pop("MANNER_HIP_GEMM_PANEL", None)
    else:
        os.environ["MANNER_HIP_GEMM_PANEL"] = "256" if panel == 1 else "192"
    os.environ["MANNER_HIP_DLN_FANIN"] = "0"
    ref = enc.encode_cls(ids, mask, precision=prec, host_lengths=lens).clone()
    os.environ["MANNER_HIP_DLN_FANIN"] = "1"
    for rep in range(4):
        got = enc.encode_cls(ids, mask, precision=prec, host_lengths=lens)
        calls += 1
        if not torch.equal(got, ref):
            print(f"MISMATCH: n={n} a={a} {prec} asm={os.environ['MANNER_HIP_GEMM_ASM']} panel={panel} rep={rep} max|d|={float((got - ref).abs().max()):.3e}", flush=True)
            sys.exit(1)
    cases += 1
    if cases % 25 == 0:
        print(f"{cases} cases, {calls} fan-in calls, {time.time() - t0:.0f} s: all equal", flush=True)
enc.status()
print(f"done: {cases} cases, {calls} fan-in calls in {time.time() - t0:.0f} s, every output bit-identical to the finalize-kernel path")
