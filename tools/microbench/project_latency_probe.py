"""Is the 32-query projection pass bound by the latency of its tile loads?  Time per 16-row tile and CU as a function of the
bank size (a small bank stays in the Infinity Cache / L2 between calls, the full one streams from HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchioner_amd.engine import Engine
torch.set_grad_enabled(False)
g = torch.Generator(device="cuda").manual_seed(6)
for M in (16384, 32768, 65536, 131072, 262144, 591753):
    bank = torch.randn(M, 768, device="cuda", generator=g)
    q = torch.randn(32, 768, device="cuda", generator=g)
    for exact in ("0", "1"):                      # PIO_PROJECT_EXACT is read when the bank is set
        if exact == "1":
            os.environ["PIO_PROJECT_EXACT"] = "1"
        e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, max_prefixes=128, vit_dtype="fp16")
        e.set_memory_bank(bank)
        os.environ.pop("PIO_PROJECT_EXACT", None)
        for _ in range(5): e.project(q.clone(), normalize=True)
        torch.cuda.synchronize(); t = time.perf_counter()
        R = 30
        for _ in range(R): e.project(q.clone(), normalize=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / R
        tiles_per_cu = M / 16 / 256
        print("M=%7d (%6.1f MB) %s: %.3f ms per call, %.2f us per tile and CU (%.1f tiles per CU)"
              % (M, M * 3072 / 1e6, "exact fp32" if exact == "1" else "split fp16", dt * 1e3, (dt * 1e6 - 60) / tiles_per_cu, tiles_per_cu), flush=True)
        e.close()
    del bank
