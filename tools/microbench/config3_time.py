"""BASELINE config 3 shape: talk2dino_decap at 518^2 (T = 1374), batch 8, 16 Gaussian-weighted boxes per image, full bank:
one synchronous forward, ms per batch and per stage (HIP-event brackets)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from patchioner_amd import Patchioner, weights as W
torch.set_grad_enabled(False)
B, NB, crop = 8, 16, 518
bank = W.synth_bank(2, 591753)
cfg = {"decap_weights": W.synth_decap(3), "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 591753,
       "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": crop, "crop_dim": crop,
       "dino_weights": W.synth_dinov2(1, "dinov2_vitb14_reg"), "memory_bank": bank, "max_batch": B, "max_prefixes": 256}
m = Patchioner.from_config(cfg, device="cuda")
imgs = W.synth_images(5, B, crop).cuda()
rng = np.random.RandomState(4)
xy = rng.randint(0, 30, size=(B, NB, 2)) * 14.0
wh = rng.randint(1, 8, size=(B, NB, 2)) * 14.0
boxes = torch.tensor(np.concatenate([xy, wh], -1), dtype=torch.float32)
kw = dict(get_cls_capt=False, gaussian_avg=True, gaussian_bbox_variance=1.0)
for _ in range(3): m(imgs, bboxes=boxes.clone(), **kw)
torch.cuda.synchronize(); t = time.perf_counter()
n = 10
for _ in range(n): out = m(imgs, bboxes=boxes.clone(), **kw)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print("config 3 shape: %.2f ms per batch of %d images x %d boxes = %.0f box captions/s" % (dt * 1e3, B, NB, B * NB / dt))
m.engine.profile_enable(True)
for _ in range(3): m(imgs, bboxes=boxes.clone(), **kw)
torch.cuda.synchronize()
for k, v in m.engine.profile_read().items():
    if v["launches"]: print("  %-14s %.3f ms/forward  %s" % (k, v["ms"] / 3, ("%.0f TFLOP/s" % (v["flops"] / v["ms"] / 1e9)) if v["flops"] else ""))

# the same shape through the pipeline (RegionCaptionPipeline + BoxRegions): stage 1 of batch i+1 under the decode of batch i
from patchioner_amd.pipeline import RegionCaptionPipeline, BoxRegions
for clones, gb in ((1, 1), (2, 1), (1, 2), (2, 2)):          # gb = 2: one greedy decode per 256 boxes
    pipe = RegionCaptionPipeline(m, group_batches=gb, decode_clones=clones)
    list(pipe.run((imgs, BoxRegions(boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=1.0)) for _ in range(4)))
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 24
    for _ in pipe.run((imgs, BoxRegions(boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=1.0)) for _ in range(n)):
        pass
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("config 3 shape, pipelined (%d decodes in flight, %d boxes per decode): %.2f ms per batch = %.0f box captions/s" % (1 + clones, gb * B * NB, dt * 1e3, B * NB / dt))
    pipe.close()
