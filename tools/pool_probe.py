#!/usr/bin/env python3
"""Round 4: the K11 pooler alone at the evaluation shape (B = 4096, S = 50, D = 768, Q = 200) — the one-pass kernel (csrc/pool.hip)
and the strict two-pass path, by HIP events; prints one JSON line.  Also the program `tools/profile_pool_r4.sh` puts under rocprofv3."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manner_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
B, S, D, Q = [int(v) for v in (sys.argv[1:5] + [4096, 50, 768, 200][len(sys.argv[1:5]):])]
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn((B, S, D), device=dev, generator=g)
W, bq, q = torch.randn((Q, D), device=dev, generator=g) * 0.05, torch.randn(Q, device=dev, generator=g) * 0.1, torch.randn(Q, device=dev, generator=g)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


nbytes = x.numel() * 4 + B * D * 4 + (Q * D + 2 * Q) * 4
res = {"shape": [B, S, D, Q], "algorithmic_bytes": nbytes}
for name, strict in (("one_pass", False), ("strict_two_pass", True)):
    ms = timed(lambda: hip.additive_pool(x, W, bq, q, strict=strict))
    res[name] = {"ms": ms, "GB/s": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000.0}
a, b = hip.additive_pool(x, W, bq, q), hip.additive_pool(x, W, bq, q, strict=True)
ref = torch.softmax(torch.tanh(x[:64] @ W.T + bq) @ q, dim=1).unsqueeze(1).bmm(x[:64]).squeeze(1)
res["max_abs_diff_one_pass_vs_strict"] = float((a - b).abs().max())
res["max_abs_err_vs_torch_f32_first_64"] = {"one_pass": float((a[:64] - ref).abs().max()), "strict": float((b[:64] - ref).abs().max())}
hip.check_status(dev)
print(json.dumps(res))
