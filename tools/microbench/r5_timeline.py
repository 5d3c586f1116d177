"""Where the K timed steps of bench.py go (diagnostic): the pipeline of bench.py (8 batches per decode, VB per ViT launch, 3
decodes in flight) over K batches, with events at the end of every ViT launch's staging and around every decode.
usage: r5_timeline.py K [VB] [total-aware 0/1 | plan a,b,c] [repeats] [decode clones] [quiet]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    VB = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    arg3 = sys.argv[3] if len(sys.argv) > 3 else "1"
    plan = [int(x) for x in arg3.split(",")] if "," in arg3 else None
    aware = arg3 == "1"
    clones = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    quiet = len(sys.argv) > 6
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    torch.cuda.set_device(0)
    torch.set_grad_enabled(False)
    from patchioner_amd.pipeline import TraceCaptionPipeline
    GB = int(os.environ.get("PIO_TL_GROUP", "8"))
    model = bench.build_models(0, 1, max_prefixes=16 * GB, max_batch=16 * max(VB, 10))[0]
    imgs, traces = bench.make_inputs()
    pipe = TraceCaptionPipeline(model, group_batches=GB, vit_batches=VB, decode_clones=clones)
    prime = torch.zeros(16 * GB, 768, device="cuda")
    for eng in pipe.decode_engines:
        for k in range(1, GB + 1):
            eng.decode_greedy(prime[:16 * k], steps=pipe.steps)
    marks = []
    stage0, decode0 = pipe._stage, pipe._decode

    def stage(held):
        k = pipe._nstaged % len(pipe.stage_models)
        s = pipe.stage_streams[k]
        e0 = torch.cuda.Event(enable_timing=True); e0.record(s)
        stage0(held)
        e1 = torch.cuda.Event(enable_timing=True); e1.record(s)
        marks.append(("vit x%d" % len(held), e0, e1, time.perf_counter()))

    def decode(g):
        k = pipe._ndecoded % len(pipe.decode_engines)
        s = pipe.decode_streams[k]
        rows = g.rows
        pipe._flush(g)
        with torch.cuda.stream(s):
            for ev in g.staged:
                s.wait_event(ev)
        e0 = torch.cuda.Event(enable_timing=True); e0.record(s)
        decode0(g)
        e1 = torch.cuda.Event(enable_timing=True); e1.record(s)
        marks.append(("decode %d rows on %d" % (rows, k), e0, e1, time.perf_counter()))

    pipe._stage, pipe._decode = stage, decode
    for r in range(reps + 1):
        del marks[:]
        torch.cuda.synchronize()
        base = torch.cuda.Event(enable_timing=True)
        base.record(torch.cuda.current_stream())
        pipe.stage_streams[0].wait_stream(torch.cuda.current_stream())
        t0 = time.perf_counter()
        for _ in pipe.run(((imgs, traces) for _ in range(K)), total=K if aware else None, plan=plan):
            pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r == 0:
            continue
        print("K=%d VB=%d %s clones=%d: %.2f ms = %.0f captions/s" % (K, VB, "plan " + arg3 if plan else "total-aware=%d" % aware, clones, dt * 1e3, 16 * K / dt), flush=True)
        for name, e0, e1, th in ([] if quiet else marks):
            print("   %-24s gpu %7.2f -> %7.2f ms   (host enqueued by %6.2f ms)" % (name, base.elapsed_time(e0), base.elapsed_time(e1), (th - t0) * 1e3))
    pipe.close()


if __name__ == "__main__":
    main()
