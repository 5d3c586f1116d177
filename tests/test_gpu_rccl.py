"""The RCCL branch of dist.py on real hardware: a one-rank "nccl" process group (all a one-GPU box allows) through the same calls
an N > 1 run makes -- the device-tensor all-gather of token ids (`_gather_into`'s non-gloo branch), the MAX all-reduce of the
elapsed time and the barrier of bench.py.  Run in a child process so that the test session keeps no process group."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from patchioner_amd import dist as pdist
os.environ["RANK"] = "0"; os.environ["LOCAL_RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29541"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
ids = torch.arange(16 * 30, dtype=torch.int32, device="cuda").view(16, 30)
out = torch.empty_like(ids)
pdist._gather_into(out, ids)                      # all_gather_into_tensor on device tensors: RCCL
t = torch.tensor([1.25], device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py: the slowest rank's time
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(out, ids) and float(t) == 1.25
dist.destroy_process_group()
print("rccl ok")
''' % ROOT


@pytest.mark.gpu
def test_rccl_branch_runs_on_the_device():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
