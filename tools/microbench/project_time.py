"""pio_mem_project over the full-size bank: ms per call for 16 / 32 / 48 / 64 / 128 queries, and a spot check against a
torch fp64 evaluation on a sub-bank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchioner_amd.engine import Engine
torch.set_grad_enabled(False)
e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, max_prefixes=128, vit_dtype="fp16")
g = torch.Generator(device="cuda").manual_seed(6)
M = 591753
bank = torch.randn(M, 768, device="cuda", generator=g)
e.set_memory_bank(bank)
e_exact = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, max_prefixes=128, vit_dtype="fp16")
os.environ["PIO_PROJECT_EXACT"] = "1"          # read when the bank is set
e_exact.set_memory_bank(bank)
del os.environ["PIO_PROJECT_EXACT"]
for N in ([int(a) for a in sys.argv[1:]] or (16, 32, 48, 64, 128)):
    q = torch.randn(N, 768, device="cuda", generator=g)
    for _ in range(3): out = e.project(q.clone(), normalize=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): out = e.project(q.clone(), normalize=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    # reference in fp64 on the device (plumbing only: a check, not the product path)
    qn = (q / q.norm(dim=-1, keepdim=True)).double()
    ref = torch.zeros(N, 768, dtype=torch.float64, device="cuda")
    mx = torch.full((N,), -1e30, dtype=torch.float64, device="cuda"); den = torch.zeros(N, dtype=torch.float64, device="cuda")
    for s in range(0, M, 65536):
        b = bank[s:s + 65536].double()
        sim = qn @ (b / b.norm(dim=-1, keepdim=True)).T / 0.01
        m2 = torch.maximum(mx, sim.max(dim=1).values)
        w = torch.exp(sim - m2[:, None]); sc = torch.exp(mx - m2)
        ref = ref * sc[:, None] + w @ b; den = den * sc + w.sum(1); mx = m2
    ref = ref / den[:, None]; ref = ref / ref.norm(dim=-1, keepdim=True)
    err = (out.double() - ref).abs().max().item()
    # the exact form (GEMM2 on the fp32 matrix pipe) on the same input: time, error against fp64, distance between the two
    for _ in range(3): old = e_exact.project(q.clone(), normalize=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): old = e_exact.project(q.clone(), normalize=True)
    torch.cuda.synchronize(); dt1 = (time.perf_counter() - t) / 10
    err1 = (old.double() - ref).abs().max().item()
    print("N=%3d: %.3f ms per call (%.3f ms per 16 queries), max |err| vs fp64 %.2e | exact fp32 GEMM2: %.3f ms, max |err| %.2e | max |split - exact| %.2e"
          % (N, dt * 1e3, dt * 1e3 * 16 / N, err, dt1 * 1e3, err1, (out - old).abs().max().item()), flush=True)
