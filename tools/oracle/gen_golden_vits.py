"""tests/golden/attn_readout_vits.npz: the reference's CLS-attention read-out with the ViT-S settings (6 heads of 64
channels, scale 0.125; P/src/model.py:336-337, dino_extraction.py:24-34, model.py:869-872) on seeded inputs.
    python tools/oracle/gen_golden_vits.py"""
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

torch.set_grad_enabled(False)
ref = refshim.load()
c = gc.ATTN_VITS
qkv, patches = gc.attn_vits_inputs()
T = c["G"] + c["n"] ** 2
self_attn, maps = ref.dino_extraction.process_self_attention(qkv, c["B"], T, c["heads"], c["D"], c["scale"], c["G"],
                                                             ret_self_attn_maps=True)
avg = (self_attn.unsqueeze(-1) * patches).mean(dim=1)
dis = (patches.unsqueeze(1) * maps.softmax(dim=-1).unsqueeze(-1)).mean(dim=2)
path = os.path.join(ROOT, "tests", "golden", "attn_readout_vits.npz")
np.savez_compressed(path, self_attn=self_attn.numpy(), maps=maps.numpy(), avg_self_attn_token=avg.numpy(),
                    disentangled=dis.numpy())
print("wrote", path, os.path.getsize(path) // 1024, "KB")
