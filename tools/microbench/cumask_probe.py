"""Does a CU-masked stream (pio_stream_create) restrict a kernel?  Times stage 1 alone on streams of 256/128/64/32 CUs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from patchioner_amd.pipeline import TraceCaptionPipeline
torch.set_grad_enabled(False)
model = bench.build_models(0, 1)[0]
imgs, traces = bench.make_inputs()
for cus in (0, 128, 64, 32):
    pipe = TraceCaptionPipeline(model, group_batches=4, stage_cus=cus or None)
    g = pipe.groups[0]
    for rep in range(2):
        g.rows, g.counts, g.staged = 0, [], []
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(pipe.stage_streams[0])
        for _ in range(4):
            pipe._stage(g, imgs, traces)
        a1.record(pipe.stage_streams[0])
        torch.cuda.synchronize()
    print("stage_cus %3d: stage1 x4 alone %.3f ms" % (cus, a0.elapsed_time(a1)))
    pipe.close()
