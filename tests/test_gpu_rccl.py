"""The RCCL path of torch.distributed on the GPU box — as far as one GPU allows.  Run on the MI355X box: ``pytest -m gpu``."""
import pytest

pytestmark = pytest.mark.gpu

def test_rccl_backend_initialises_and_runs_the_exchange_collectives_on_this_box():
    """SURVEY §8e on the REAL backend, as far as one GPU allows: a world-size-1 `nccl` (= RCCL) process group in a child process —
    rendezvous on 127.0.0.1, `all_gather_into_tensor` of a table block (the default exchange of `MeshTableGather` / `all_gather_table`),
    the metric `all_reduce`, a barrier — on GPU tensors of the shapes the table exchange uses.  The multi-rank LOGIC is covered by the
    gloo tests of tests/test_host.py (worlds 2 - 8); this one only shows that the RCCL path of torch.distributed loads and executes
    here, so that the first 8-GPU run is not also the first RCCL call of the build."""
    import os
    import subprocess
    import sys
    code = r"""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MANNER_TEST_PORT", "29617"), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
from manner_amd import distributed as D
assert D.device_backend("cuda") == "nccl"
send = torch.arange(20127 * 768, dtype=torch.float32, device="cuda").view(20127, 768)     # one rank's block of the MIND-large table
recv = torch.empty_like(send)
dist.all_gather_into_tensor(recv, send)
v = torch.tensor([1.5, 2.0, 3.0], device="cuda", dtype=torch.float64)
dist.all_reduce(v, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(recv, send) and v.tolist() == [1.5, 2.0, 3.0]
g = D.MeshTableGather(161013, 768, "cuda", pieces=4)
assert g.exchange == "none" and g.wait().shape == (161013, 768)
assert D.allreduce_metric_sums(v).tolist() == [1.5, 2.0, 3.0]
dist.destroy_process_group()
print("rccl ok")
"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] +
                                                                                      [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
