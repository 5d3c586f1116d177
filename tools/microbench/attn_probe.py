"""k_vit_attention per-launch time inside a ViT forward (HIP-event brackets of the PIO_PROF_VIT_ATTN class), and a checksum of
the tokens so that two builds / variants can be compared:  [PIO_LIB_PATH=<variant .so>] python tools/microbench/attn_probe.py [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from patchioner_amd import weights as W
from patchioner_amd.engine import Engine


def main():
    torch.cuda.set_device(0)
    for crop, sizes in ((224, [int(a) for a in sys.argv[1:]] or [16, 64]), (518, [8])):
        e = Engine(embed_dim=768, depth=12, num_heads=12, num_registers=4, crop_dim=crop, max_batch=max(sizes), vit_dtype="fp16")
        e.load_state_dict(W.synth_dinov2(1))
        e.finalize()
        for B in sizes:
            imgs = W.synth_images(7, B, crop).cuda()
            for _ in range(3):
                tok = e.vit_forward(imgs)[0]
            e.profile_enable(True)
            for _ in range(5):
                e.vit_forward(imgs)
            torch.cuda.synchronize()
            p = e.profile_read()
            e.profile_enable(False)
            a, g, ln = p["vit_attention"], p["vit_gemm"], p["vit_layernorm"]
            print("crop %d B=%3d  attention %.1f us/launch (%.0f TF)   gemm %.1f us/launch   layernorm %.1f us/launch   tokens sum %.6f absmax %.4f"
                  % (crop, B, a["ms"] * 1e3 / a["launches"], a["flops"] / a["ms"] / 1e9, g["ms"] * 1e3 / g["launches"],
                     ln["ms"] * 1e3 / ln["launches"], float(tok.double().sum()), float(tok.abs().max())), flush=True)
        e.close()


main()
