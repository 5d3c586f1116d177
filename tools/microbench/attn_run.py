"""Three ViT forwards of B images (default 64) and nothing else -- the program rocprofv3 --pmc collects the k_vit_attention
counters on:  rocprofv3 --pmc ... -- python3 tools/microbench/attn_run.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from patchioner_amd import weights as W
from patchioner_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.cuda.set_device(0)
e = Engine(embed_dim=768, depth=12, num_heads=12, num_registers=4, crop_dim=224, max_batch=B, vit_dtype="fp16")
e.load_state_dict(W.synth_dinov2(1))
e.finalize()
imgs = W.synth_images(7, B, 224).cuda()
for _ in range(3):
    e.vit_forward(imgs)
torch.cuda.synchronize()
e.close()
