"""Ten 32-query projection passes over the full-size bank and nothing else (the program rocprofv3 --pmc collects counters on):
    [PIO_PROJECT_EXACT=1] rocprofv3 --pmc ... -- python3 tools/microbench/project_run.py [N]      (tools/microbench/pmc_project.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchioner_amd.engine import Engine
torch.set_grad_enabled(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, max_prefixes=128, vit_dtype="fp16")
g = torch.Generator(device="cuda").manual_seed(6)
e.set_memory_bank(torch.randn(591753, 768, device="cuda", generator=g))
q = torch.randn(N, 768, device="cuda", generator=g)
for _ in range(10):
    e.project(q.clone(), normalize=True)
torch.cuda.synchronize()
e.close()
