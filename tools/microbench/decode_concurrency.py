"""How many greedy-decode chains can the chip carry side by side when NOTHING else runs (diagnostic)?  k decoders (the engine's own +
clones on the same weights), each on a stream of its own that the pipeline's probe found to be concurrent, each decoding N prefixes;
wall time per round and captions/s.  Beside it: one ViT launch of 80 images with its projections, alone.
    python tools/microbench/decode_concurrency.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from patchioner_amd import Patchioner, weights as W
from patchioner_amd.pipeline import TraceCaptionPipeline


def main():
    torch.cuda.set_device(0)
    cfg = {"decap_weights": W.synth_decap(3), "dino_weights": W.synth_dinov2(1), "memory_bank": W.synth_bank(6, 4096).cuda(),
           "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 4096, "dino_model": "dinov2_vitb14_reg",
           "normalize": True, "resize_dim": 224, "crop_dim": 224, "max_batch": 80, "max_prefixes": 256}
    m = Patchioner.from_config(cfg, device="cuda:0")
    KMAX = 5
    pipe = TraceCaptionPipeline(m, group_batches=8, vit_batches=5, decode_clones=KMAX - 1)
    engines, streams = pipe.decode_engines, pipe.decode_streams
    print("stream probe:", getattr(pipe, "stream_probe", None), flush=True)
    g = torch.Generator(device="cuda").manual_seed(5)
    for N in ([int(a) for a in sys.argv[1:]] or (64, 128, 256)):
        pres = [torch.nn.functional.normalize(torch.randn(N, 768, device="cuda", generator=g), dim=-1) for _ in range(KMAX)]
        for e, p in zip(engines, pres):
            e.decode_greedy(p, steps=30)
        torch.cuda.synchronize()
        for k in range(1, KMAX + 1):
            def round_():
                for e, st, p in zip(engines[:k], streams[:k], pres):
                    with torch.cuda.stream(st):
                        e.decode_greedy(p, steps=30)
            round_(); torch.cuda.synchronize()
            t = time.perf_counter()
            R = 4
            for _ in range(R):
                round_()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / R
            print("N=%3d  %d chains in flight: %.2f ms per round = %.1f k captions/s" % (N, k, dt * 1e3, k * N / dt / 1e3), flush=True)
    imgs = W.synth_images(7, 80, 224).cuda()
    for _ in range(2):
        m.engine.vit_forward(imgs)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        m.engine.vit_forward(imgs)
    torch.cuda.synchronize()
    print("ViT forward of 80 images alone: %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
    pipe.close()


main()
