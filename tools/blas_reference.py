#!/usr/bin/env python3
"""Development aid (not part of the product path): what the vendor GEMM library reaches on the encoder's GEMM shapes,
as a yardstick for gemm_tn_x16_kernel.  torch.nn.functional.linear in bf16 dispatches to hipBLASLt / rocBLAS."""
import torch

M = 65536
shapes = {"qkv": (2304, 768), "out": (768, 768), "ffn1": (3072, 768), "ffn2": (768, 3072),
          "rl-ffn1": (4096, 1024), "rl-ffn2": (1024, 4096)}
dev = "cuda"
for name, (n, k) in shapes.items():
    x = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        y = torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = torch.nn.functional.linear(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:8s} M={M} N={n} K={k}: {ms * 1e3:7.1f} us  {2.0 * M * n * k / ms / 1e9:6.0f} TFLOP/s (library, bias only)")
