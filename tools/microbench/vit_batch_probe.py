"""ViT forward time per image as a function of the images per launch (diagnostic):  python tools/microbench/vit_batch_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from patchioner_amd import Patchioner, weights as W

def main():
    torch.cuda.set_device(0)
    cfg = {"decap_weights": W.synth_decap(3), "dino_weights": W.synth_dinov2(1), "memory_bank": W.synth_bank(6, 4096).cuda(),
           "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 4096, "dino_model": "dinov2_vitb14_reg",
           "normalize": True, "resize_dim": bench.CROP, "crop_dim": bench.CROP, "max_batch": 128, "max_prefixes": 64}
    m = Patchioner.from_config(cfg, device="cuda:0")
    eng = m.engine
    big = W.synth_images(7, 64, bench.CROP).cuda()
    t64 = eng.vit_forward(big)[0]
    t16 = torch.cat([eng.vit_forward(big[i:i + 16])[0] for i in range(0, 64, 16)])
    print("64 images in one launch == 4 launches of 16, bitwise:", bool(torch.equal(t64, t16)), flush=True)
    for B in ([int(a) for a in sys.argv[1:]] or (16, 32, 48, 64, 128)):
        imgs = W.synth_images(1, B, bench.CROP).cuda()
        for _ in range(3): eng.vit_forward(imgs)
        torch.cuda.synchronize()
        t = time.perf_counter()
        R = 20
        for _ in range(R): eng.vit_forward(imgs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / R
        print("B=%3d  %.3f ms per launch   %.1f us per image" % (B, dt * 1e3, dt * 1e6 / B), flush=True)

main()
