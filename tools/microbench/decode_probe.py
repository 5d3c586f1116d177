"""Greedy decode of N random prefixes, a few times (diagnostic; run under rocprofv3 for per-kernel times):
python tools/microbench/decode_probe.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from patchioner_amd import Patchioner, weights as W

def main():
    torch.cuda.set_device(0)
    cfg = {"decap_weights": W.synth_decap(3), "dino_weights": W.synth_dinov2(1, depth=1), "memory_bank": W.synth_bank(6, 4096).cuda(),
           "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 4096, "dino_model": "dinov2_vitb14_reg",
           "normalize": True, "resize_dim": 224, "crop_dim": 224, "max_batch": 16, "max_prefixes": 256}
    m = Patchioner.from_config(cfg, device="cuda:0")
    g = torch.Generator(device="cuda").manual_seed(5)
    for N in ([int(a) for a in sys.argv[1:]] or (16, 64, 128, 256)):
        pre = torch.randn(N, 768, device="cuda", generator=g)
        pre = pre / pre.norm(dim=-1, keepdim=True)
        for _ in range(2): m.engine.decode_greedy(pre, steps=30)
        torch.cuda.synchronize()
        t = time.perf_counter()
        R = 5
        for _ in range(R): m.engine.decode_greedy(pre, steps=30)
        torch.cuda.synchronize()
        print("N=%3d  %.3f ms per 30-step decode" % (N, (time.perf_counter() - t) / R * 1e3), flush=True)

main()
