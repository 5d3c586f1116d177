"""How long does the host take to ENQUEUE one forward (forward_async returns) vs the GPU to run it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
torch.set_grad_enabled(False)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
models = bench.build_models(0, P)
streams = [torch.cuda.Stream() for _ in range(P)]
imgs, traces = bench.make_inputs()
for m in models:
    m(imgs, get_cls_capt=False, traces=traces)
torch.cuda.synchronize()
sub, res = [], []
from collections import deque
pend = deque()
t_all = time.perf_counter()
K = 40
for i in range(K):
    if len(pend) == P:
        t = time.perf_counter(); pend.popleft().result(); res.append(time.perf_counter() - t)
    t = time.perf_counter()
    pend.append(models[i % P].forward_async(imgs, stream=streams[i % P], get_cls_capt=False, traces=traces))
    sub.append(time.perf_counter() - t)
while pend:
    pend.popleft().result()
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print("P=%d  total %.2f ms/step | submit avg %.3f ms (min %.3f) | result-wait avg %.3f ms" %
      (P, tot / K * 1e3, sum(sub) / len(sub) * 1e3, min(sub) * 1e3, sum(res) / max(len(res), 1) * 1e3))
# host-only cost of the pieces
m = models[0]
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    tok, qkv = m.engine.vit_forward(imgs)
host_vit = (time.perf_counter() - t) / 20
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    g = m.engine.trace_grids(traces)
host_tr = (time.perf_counter() - t) / 20
torch.cuda.synchronize()
pre = torch.randn(16, 768, device="cuda")
t = time.perf_counter()
for _ in range(20):
    ids, _ = m.engine.decode_greedy(pre)
host_dec = (time.perf_counter() - t) / 20
torch.cuda.synchronize()
print("host enqueue: vit_forward %.3f ms, trace_grids %.3f ms, decode %.3f ms" % (host_vit * 1e3, host_tr * 1e3, host_dec * 1e3))
