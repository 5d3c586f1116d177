import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import golden_cases as gc
from patchioner_amd import Patchioner, weights as W
from patchioner_amd.pipeline import TraceCaptionPipeline
torch.set_grad_enabled(False)
for dt, crop, mb in (("bf16", 224, 8), ("fp16", 518, 4)):
    cfg = {"decap_weights": W.synth_decap(3), "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 4096,
           "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": crop, "crop_dim": crop,
           "dino_weights": W.synth_dinov2(91, "dinov2_vitb14_reg", depth=3), "memory_bank": W.synth_bank(61, 4096),
           "max_batch": mb, "vit_dtype": dt, "max_prefixes": 128}
    m = Patchioner.from_config(cfg, device="cuda")
    n = crop // 14
    batches = []
    for i in range(5):
        imgs = W.synth_images(10 + i, mb, crop).cuda()
        traces = [[dict(x=(j + 0.5) / n, y=(k + 0.5) / n) for j in range(3 + b) for k in range(2)] for b in range(mb)]
        batches.append((imgs, traces))
    want = [m(b, get_cls_capt=False, traces=t)["trace_capts"] for b, t in batches]
    got = list(TraceCaptionPipeline(m, group_batches=4).run(batches))
    print(dt, crop, "pipeline == sync:", got == want, "| sample:", want[0][0][:40])
    out = m(batches[0][0], get_cls_capt=True, get_avg_self_attn_capt=True, get_avg_patch_capt=True, compute_scores=True,
            bboxes=torch.tensor([[[14.0, 14.0, 60.0, 60.0], [0.0, 0.0, 1.0, 1.0]]] * mb), gaussian_avg=True)
    print("  keys:", sorted(out))
