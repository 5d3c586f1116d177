"""Golden vectors for the image transforms, made with Pillow itself (the third-party code behind the reference's
torchvision Resize; P/src/model.py:347-357): small seeded images, PIL.Image.resize(..., BICUBIC) outputs, and the full
transform through the size rules of torchvision restated in patchioner_amd/preprocess.py.
    python tools/oracle/gen_golden_preprocess.py   ->  tests/golden/preprocess.npz"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402

out = {"pil_version": np.array(Image.__version__ if hasattr(Image, "__version__") else "?")}
import PIL  # noqa: E402
out["pil_version"] = np.array(PIL.__version__)
for i, (w, h, nw, nh) in enumerate(gc.PREP_RESIZE_CASES):
    arr = gc.prep_image(i, w, h)
    out["resize_%d" % i] = np.asarray(Image.fromarray(arr).resize((nw, nh), Image.BICUBIC))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "preprocess.npz"), **out)
print("wrote", {k: v.shape for k, v in out.items()})
