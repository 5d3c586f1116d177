"""What slows stage 1 when a decode runs beside it?  Stage 1 x4 on stream A with, on stream B:
   nothing | a graph of 660 tiny dependent kernels (kernel boundaries only) | the real decode(64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from patchioner_amd.pipeline import TraceCaptionPipeline
torch.set_grad_enabled(False)
model = bench.build_models(0, 1)[0]
imgs, traces = bench.make_inputs()
pipe = TraceCaptionPipeline(model, group_batches=4)
g = pipe.groups[0]
sa, sb = pipe.stage_streams[0], pipe.sb
ev = lambda: torch.cuda.Event(enable_timing=True)

# a graph of 660 tiny dependent kernels
tiny = torch.zeros(64, device="cuda")
gr = torch.cuda.CUDAGraph()
with torch.cuda.stream(sb):
    for _ in range(3): tiny.add_(1.0)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=sb):
        for _ in range(660): tiny.add_(1.0)
prefix = torch.randn(64, 768, device="cuda")
model.engine.decode_greedy(prefix); torch.cuda.synchronize()

def run(kind):
    g.rows, g.counts, g.staged = 0, [], []
    torch.cuda.synchronize()
    a0, a1, b0, b1 = ev(), ev(), ev(), ev()
    a0.record(sa)
    with torch.cuda.stream(sb):
        b0.record(sb)
        if kind == "tiny":
            gr.replay(); gr.replay()
        elif kind == "decode":
            model.engine.decode_greedy(prefix)
        b1.record(sb)
    for _ in range(4):
        pipe._stage(g, imgs, traces)
    a1.record(sa)
    torch.cuda.synchronize()
    return a0.elapsed_time(a1), b0.elapsed_time(b1)

for kind in ("none", "tiny", "decode", "none", "tiny", "decode"):
    a, b = run(kind)
    print("%-7s stage1 x4 %.3f ms   stream-B work %.3f ms" % (kind, a, b))
# B alone
for kind in ("tiny", "decode"):
    torch.cuda.synchronize()
    b0, b1 = ev(), ev()
    with torch.cuda.stream(sb):
        b0.record(sb)
        if kind == "tiny":
            gr.replay(); gr.replay()
        else:
            model.engine.decode_greedy(prefix)
        b1.record(sb)
    torch.cuda.synchronize()
    print("%-7s alone on B %.3f ms" % (kind, b0.elapsed_time(b1)))
