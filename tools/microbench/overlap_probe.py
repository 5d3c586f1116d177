"""Do independent decode graphs (and ViT forwards) on different streams overlap on this GPU?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from patchioner_amd.engine import Engine
from patchioner_amd import weights as W
torch.set_grad_enabled(False)
P = 4
engs = []
dec_sd, vit_sd = W.synth_decap(3), W.synth_dinov2(1)
for _ in range(P):
    e = Engine(embed_dim=768, depth=12, num_heads=12, num_registers=4, crop_dim=224, max_batch=16, max_prefixes=64)
    e.load_state_dict(vit_sd); e.load_state_dict(dec_sd); e.finalize(); engs.append(e)
streams = [torch.cuda.Stream() for _ in range(P)]
pre = torch.randn(16, 768, device="cuda")
imgs = torch.randn(16, 3, 224, 224, device="cuda")
for e in engs:
    e.decode_greedy(pre); e.vit_forward(imgs)
torch.cuda.synchronize()

def run(fn, p, reps=8):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        for i in range(p):
            with torch.cuda.stream(streams[i]):
                fn(engs[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3

for p in (1, 2, 4):
    print("decode x%d concurrently: %.2f ms per round (%.2f ms per decode)" % (p, run(lambda e: e.decode_greedy(pre), p), run(lambda e: e.decode_greedy(pre), p) / p))
for p in (1, 2, 4):
    print("vit    x%d concurrently: %.2f ms per round" % (p, run(lambda e: e.vit_forward(imgs, want_qkv=False), p)))
def mixed(e):
    e.vit_forward(imgs, want_qkv=False); e.decode_greedy(pre)
for p in (1, 2, 4):
    r = run(mixed, p)
    print("vit+decode x%d: %.2f ms per round (%.2f per item)" % (p, r, r / p))
