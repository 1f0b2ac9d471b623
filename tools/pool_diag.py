#!/usr/bin/env python3
"""Development aid: per-phase s_memtime sums of the one-pass pooler's pass loop (library built with -DMANNER_POOL_DIAG by tools/pool_diag.sh;
never part of a measured run)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
nw = int(os.environ.get("MANNER_HIP_POOL_NW", "4"))
dbg = torch.zeros(64 * 8 * 6, dtype=torch.int64, device=dev)
os.environ["MANNER_HIP_POOL_DIAG"] = str(dbg.data_ptr())
from manner_amd import hip  # noqa: E402
B, S, D, Q = 4096, 50, 768, 200
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn((B, S, D), device=dev, generator=g)
W, bq, q = torch.randn((Q, D), device=dev, generator=g) * 0.05, torch.randn(Q, device=dev, generator=g) * 0.1, torch.randn(Q, device=dev, generator=g)
for _ in range(5):
    hip.additive_pool(x, W, bq, q)
torch.cuda.synchronize()
st = dbg.cpu().view(64, 8, 6)[:, :nw].double()
ok = st[..., 3] > 0
names = ["s_waitcnt vmcnt (own DMA pieces)", "s_barrier", "early DMA issue (waves 4-7 of 8)", "fragment reads + MFMAs", "late DMA issue", "-"]
tot = st[..., :5].sum(-1)[ok].mean()
for i, n in enumerate(names[:5]):
    v = st[..., i][ok]
    print(f"{n:36s} mean {v.mean():9.0f} cycles per wave over the 3 passes ({100 * v.mean() / tot:4.1f} %)   per stage {v.mean() / 36:7.0f}")
print("sum", float(tot), "stages per wave: 36 (3 passes x 12 stages of 2 k-steps); own MFMAs per stage: 2 x 15 x 16 = 480 cycles (13 tiles: 416 avg)")
