"""A few pio_mem_project calls (16 and 32 queries) over the full-size bank, for rocprofv3 --pmc runs (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchioner_amd.engine import Engine
torch.set_grad_enabled(False)
e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, max_prefixes=128, vit_dtype="fp16")
g = torch.Generator(device="cuda").manual_seed(6)
bank = torch.randn(591753, 768, device="cuda", generator=g)
e.set_memory_bank(bank)
for N in (16, 32):
    q = torch.randn(N, 768, device="cuda", generator=g)
    for _ in range(3):
        e.project(q.clone(), normalize=True)
torch.cuda.synchronize()
