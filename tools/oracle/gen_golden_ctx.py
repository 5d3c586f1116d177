"""tests/golden/ctx_cleaner.npz: the reference's Patchioner.ctx_cleaner (P/src/model.py:1425-1436) on seeded inputs.
    python tools/oracle/gen_golden_ctx.py"""
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
dirty, ctx = gc.ctx_inputs()
out = {}
for ct in ("orthogonal_projection", "contrastive_mask"):
    for alpha in (1.0, 0.35):
        out["%s_%g" % (ct, alpha)] = ref.model.Patchioner.ctx_cleaner(None, dirty.clone(), ctx.clone(), cleaning_type=ct,
                                                                      alpha=alpha).numpy()
path = os.path.join(ROOT, "tests", "golden", "ctx_cleaner.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KB", sorted(out))
