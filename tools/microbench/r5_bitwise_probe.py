"""Where do an 80-image ViT launch and five 16-image launches differ?  (round 5 debugging aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from patchioner_amd import weights as W
from patchioner_amd.engine import Engine
torch.set_grad_enabled(False)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 80
sd = W.synth_dinov2(17, depth=depth)
big = Engine(embed_dim=768, depth=depth, num_heads=12, num_registers=4, crop_dim=224, max_batch=nb, vit_dtype="fp16")
small = Engine(embed_dim=768, depth=depth, num_heads=12, num_registers=4, crop_dim=224, max_batch=16, vit_dtype="fp16")
for e in (big, small):
    e.load_state_dict(sd); e.finalize()
imgs = W.synth_images(5, nb, 224).cuda()
tb, qb = big.vit_forward(imgs)
ts, qs = small.vit_forward(imgs)
torch.cuda.synchronize()
d = (tb != ts)
print("depth", depth, "tokens differ:", int(d.sum()), "of", d.numel(), " qkv differ:", int((qb != qs).sum()))
if d.any():
    idx = d.nonzero()
    print("images with differences:", sorted(set(idx[:, 0].tolist()))[:20], "...")
    print("token rows:", sorted(set(idx[:, 1].tolist()))[:40])
    cols = idx[:, 2]
    print("columns by 256-tile:", [int(((cols >= 256 * t) & (cols < 256 * (t + 1))).sum()) for t in range(3)])
    print("max abs diff", float((tb - ts).abs().max()), "max abs", float(ts.abs().max()))
    # global row index in the launch: b * 264 + t
    g = idx[:, 0] * 264 + idx[:, 1]
    print("global rows mod 256 histogram (16 bins):", torch.histc((g % 256).float(), 16, 0, 256).int().tolist())
    print("row tiles touched:", sorted(set((g // 256).tolist()))[:50])
