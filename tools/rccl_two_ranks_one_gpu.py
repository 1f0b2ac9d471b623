#!/usr/bin/env python3
"""Probe: do two RCCL ranks on ONE GPU work on this box?  (If they did, the mesh exchange could be rehearsed on the real backend.)
RESULT (round 4, RCCL 2.26.6 of torch 2.10+rocm7.0): no — `ncclInvalidUsage: Duplicate GPU detected : rank 0 and rank 1 both on CUDA
device d000` at the first collective.  The world-size-1 nccl test (tests/test_gpu_rccl.py) is what one GPU allows.
    python tools/rccl_two_ranks_one_gpu.py          # spawns 2 children, prints what happened; bounded by its own timeouts"""
import os
import subprocess
import sys

CHILD = r"""
import os, sys, torch, torch.distributed as dist, datetime
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=40))
x = torch.full((1024, 768), float(rank + 1), device="cuda")
out = torch.empty((2048, 768), device="cuda")
dist.all_gather_into_tensor(out, x)
torch.cuda.synchronize()
print("rank", rank, "allgather ok", float(out[0, 0]), float(out[1024, 0]), flush=True)
peer = 1 - rank
recv = torch.empty_like(x)
works = dist.batch_isend_irecv([dist.P2POp(dist.isend, x, peer), dist.P2POp(dist.irecv, recv, peer)])
for w in works: w.wait()
torch.cuda.synchronize()
print("rank", rank, "p2p ok", float(recv[0, 0]), flush=True)
dist.destroy_process_group()
"""
if __name__ == "__main__":
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, "-c", CHILD], env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
          for r in range(2)]
    for r, p in enumerate(ps):
        try:
            out, _ = p.communicate(timeout=75)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[killed after 75 s]"
        print(f"--- rank {r} rc={p.returncode}\n{out[-1200:]}")
