"""Order effects on Engine.preprocess: forwards (small trace staging slots) first, then the image transforms; per-call host
times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, numpy as np
import bench, golden_cases as gc
from PIL import Image
torch.set_grad_enabled(False)
model = bench.build_models(0, 1, max_prefixes=128)[0]
imgs, traces = bench.make_inputs()
for _ in range(20): model(imgs, get_cls_capt=False, traces=traces)
raw = [gc.prep_image(300 + i, 640, 480) for i in range(16)]
for _ in range(9): model.preprocess_images(raw)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); model.preprocess_images(raw); ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("per-call host ms:", " ".join("%.2f" % t for t in ts))
pil = [Image.fromarray(a) for a in raw]
t0 = time.perf_counter(); host = torch.stack([model.image_transforms(im) for im in pil]); print("host PIL %.1f ms" % ((time.perf_counter() - t0) * 1e3))
ts = []
for _ in range(10):
    t0 = time.perf_counter(); model.preprocess_images(raw); ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("after PIL, per-call host ms:", " ".join("%.2f" % t for t in ts))
