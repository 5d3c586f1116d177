"""Where the device-side image transform's time goes (host packing, PCIe, table building, kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import golden_cases as gc
from patchioner_amd.engine import Engine
e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, vit_dtype="fp16")
raw = [gc.prep_image(300 + i, 640, 480) for i in range(16)]
for _ in range(9): e.preprocess(raw, 224, 224)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): out = e.preprocess(raw, 224, 224)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host side of 20 calls %.2f ms/call, + drain %.2f ms" % ((t1 - t) / 20 * 1e3, (t2 - t1) * 1e3))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); out = e.preprocess(raw, 224, 224); b.record(); torch.cuda.synchronize()
print("one call, GPU span (copy + kernels, includes host gaps) %.3f ms" % a.elapsed_time(b))
# pieces
pin = torch.empty(16 * 640 * 480 * 3, dtype=torch.uint8).pin_memory(); dev = torch.empty_like(pin, device="cuda")
t = time.perf_counter()
for _ in range(20):
    h = pin.numpy(); o = 0
    for r in raw:
        h[o:o + r.size] = r.reshape(-1); o += r.size
print("pack into pinned: %.2f ms" % ((time.perf_counter() - t) / 20 * 1e3))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); print("H2D 14.7 MB: %.2f ms" % ((time.perf_counter() - t) / 20 * 1e3))
