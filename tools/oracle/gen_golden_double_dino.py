"""tests/golden/double_dino.npz: the REFERENCE's extract_bboxes_feats_double_dino (P/src/bbox_utils.py:300-403, through
refshim) on seeded final tokens and boxes, with the DINOv2-shaped stand-in module carrying seeded weights as
``dino_model`` (only ``.blocks[-1]`` and ``.parameters()`` are used by the reference function).
    python tools/oracle/gen_golden_double_dino.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_cases as gc  # noqa: E402
import refshim  # noqa: E402
from dino_standin import DinoStandIn  # noqa: E402
from patchioner_amd import weights as W  # noqa: E402

torch.set_grad_enabled(False)
ref = refshim.load()
c = gc.DDINO
sd = W.synth_dinov2(c["seed_w"], "dinov2_vitb14_reg", depth=c["depth"])
net = DinoStandIn(768, c["depth"], 12)
net.load_state_dict(sd, strict=True)
net.eval()
tokens = gc.ddino_tokens()
out = {}
for name, boxes in (("regular", gc.boxes_regular()), ("dummies", gc.boxes_with_dummies())):
    for use_cls in (True, False):
        for rt in ("cls", "avg", "gaussian_avg"):
            if rt == "cls" and not use_cls:
                continue
            feats = ref.bbox_utils.extract_bboxes_feats_double_dino(
                net, tokens[:, 5:].clone(), boxes.clone(), tokens[:, 0].clone() if use_cls else None,
                tokens[:, 1:5].clone() if use_cls else None, 14, return_type=rt, gaussian_bbox_variance=c["variance"])
            out["%s_%s_%s" % (name, "cls" if use_cls else "nocls", rt)] = feats.numpy()
path = os.path.join(ROOT, "tests", "golden", "double_dino.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KB", sorted(out))
