"""Where the grouped-decode pipeline's time goes: per group, stage-1 span on stream A and decode span on stream B
(events), against the same spans run alone.  python tools/microbench/pipeline_probe.py [group_batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from patchioner_amd.pipeline import TraceCaptionPipeline

torch.set_grad_enabled(False)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = bench.build_models(0, 1, max_prefixes=min(128, max(64, 16 * G)))[0]
imgs, traces = bench.make_inputs()
pipe = TraceCaptionPipeline(model, group_batches=G)
ev = lambda: torch.cuda.Event(enable_timing=True)

# alone: stage 1 of G batches, then one decode of 16*G prefixes
g = pipe.groups[0]
for rep in range(3):
    g.rows, g.counts, g.staged = 0, [], []
    torch.cuda.synchronize()
    a0, a1, b1 = ev(), ev(), ev()
    a0.record(pipe.stage_streams[0])
    for _ in range(G):
        pipe._stage(g, imgs, traces)
    a1.record(pipe.stage_streams[0])
    pipe._decode(g)
    b1.record(pipe.sb)
    torch.cuda.synchronize()
    print("alone: stage1 x%d %.3f ms, decode(%d) %.3f ms" % (G, a0.elapsed_time(a1), g.rows, a1.elapsed_time(b1)))
g.rows, g.counts, g.staged, g.busy = 0, [], [], False

# pipelined: instrument _stage / _decode
spans = []
orig_stage, orig_decode = pipe._stage, pipe._decode
def stage(g, imgs, traces):
    if not g.counts:
        g.t0 = ev(); g.t0.record(pipe.stage_streams[0])
    orig_stage(g, imgs, traces)
def decode(g):
    g.t1 = ev(); g.t1.record(pipe.stage_streams[0])
    with torch.cuda.stream(pipe.sb):
        for e in g.staged:
            pipe.sb.wait_event(e)
        g.d0 = ev(); g.d0.record(pipe.sb)
    orig_decode(g)
    g.d1 = ev(); g.d1.record(pipe.sb)
    spans.append((g.t0, g.t1, g.d0, g.d1))
pipe._stage, pipe._decode = stage, decode
n = 12 * G
torch.cuda.synchronize()
t = time.perf_counter()
for _ in pipe.run((imgs, traces) for _ in range(n)):
    pass
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("pipelined: %d batches in %.2f ms = %.3f ms/batch, %.0f captions/s" % (n, dt * 1e3, dt * 1e3 / n, 16 * n / dt))
base = spans[0][0]
for i, (t0, t1, d0, d1) in enumerate(spans):
    print("group %2d: stage1 [%7.2f .. %7.2f] = %.2f ms   decode [%7.2f .. %7.2f] = %.2f ms" % (
        i, base.elapsed_time(t0), base.elapsed_time(t1), t0.elapsed_time(t1), base.elapsed_time(d0), base.elapsed_time(d1), d0.elapsed_time(d1)))
