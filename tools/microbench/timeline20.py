"""The bench's 20-step timed region alone, between two marker kernels (a fill of 7777 / 8888 floats), for a rocprofv3
--kernel-trace: where the ViT launches, the projections and the decodes of a short stream sit in time.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/microbench/timeline20.py [steps]
    python3 tools/microbench/timeline20.py --summarise OUT/.../*_kernel_trace.csv"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def summarise(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the timed region is fenced by 60 ms of host sleep on both sides: the LAST burst of pio kernels between two gaps > 30 ms
    segs, cur = [], []
    for r in rows:
        if cur and int(r["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in cur[-4:]) > 30e6:
            segs.append(cur); cur = []
        cur.append(r)
    segs.append(cur)
    segs = [g for g in segs if sum("k_vit_gemm" in r["Kernel_Name"] for r in g) > 40]
    if not segs:
        print("no timed region found"); return
    seg = segs[-1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    print("timed region: %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
    cls = lambda n: ("vit" if "k_vit_" in n or "k_layernorm" in n or "k_im2col" in n or "k_token" in n else "project" if "k_project" in n or "k_l2norm" in n
                     else "decode" if "k_dec_" in n or "lmhead" in n or "k_lm_" in n else "other")
    spans, busy = {}, {}
    for r in seg:
        s_, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        c = cls(r["Kernel_Name"])
        busy[c] = busy.get(c, 0) + (e - s_)
        spans.setdefault(c, []).append((s_ - t0, e - t0))
    for c, v in busy.items():
        print("  %-8s kernel time %6.2f ms, first start %6.2f ms, last end %6.2f ms" % (c, v / 1e6, min(a for a, _ in spans[c]) / 1e6, max(b for _, b in spans[c]) / 1e6))
    T = int((t1 - t0) / 1e6) + 1
    print("  per millisecond, the share of it a class had a kernel running (' ' none .. '@' all; kernels of one class overlap, so '@' = >= 1 ms)")
    for c in ("vit", "project", "decode"):
        line = ""
        for ms in range(T):
            lo, hi = ms * 1e6, (ms + 1) * 1e6
            cov = sum(max(0, min(b, hi) - max(a, lo)) for a, b in spans.get(c, ()))
            line += " .:-=+*#%@"[min(9, int(cov / 1e6 * 9.99))]
        print("  %-8s |%s|" % (c, line))
    # the ViT launches: runs of vit kernels separated by > 0.2 ms without one
    v = sorted(spans.get("vit", []))
    runs, st, en = [], None, None
    for a, b in v:
        if st is None: st, en = a, b
        elif a - en > 0.2e6: runs.append((st, en)); st, en = a, b
        else: en = max(en, b)
    if st is not None: runs.append((st, en))
    print("  vit bursts (ms): " + ", ".join("%.1f-%.1f" % (a / 1e6, b / 1e6) for a, b in runs))


MARK0, MARK1 = (7777, 7936, 8192), (8888, 8960, 9216)
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    summarise(sys.argv[2]); sys.exit(0)

import torch
import bench as B
from patchioner_amd.pipeline import TraceCaptionPipeline
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.cuda.set_device(0); torch.set_grad_enabled(False)
models = B.build_models(0, 1, max_prefixes=128, max_batch=B.BATCH * 5)
m = models[0]
imgs, traces = B.make_inputs()
pipe = TraceCaptionPipeline(m, group_batches=8, vit_batches=5, decode_clones=2)
for _ in pipe.run((imgs, traces) for _ in range(8)): pass
prime = torch.zeros(128, 768, device="cuda")
for eng in pipe.decode_engines:
    for rows in (128, 16 * (steps % 8)):
        if rows: eng.decode_greedy(prime[:rows], steps=30)
torch.cuda.synchronize()
# un-traced timeline from HIP events on the pipeline's own streams (PIO_TL_EVENTS=1): stage 1 of every ViT launch and every decode
marks = []
if os.environ.get("PIO_TL_EVENTS") == "1":
    base = torch.cuda.Event(enable_timing=True)
    _stage, _decode = pipe._stage, pipe._decode
    def stage(held):
        k = pipe._nstaged % len(pipe.stage_models)
        st = pipe.stage_streams[k]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); _stage(held); b.record(st)
        marks.append(("vit+proj x%d" % len(held), a, b))
    def decode(g, *rest):
        k = pipe._ndecoded % len(pipe.decode_engines)
        sb = pipe.decode_streams[k]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _decode(g, *rest)
        # the decode waits for its group's staging events inside: bracket with an event after the fact on the same stream
        b.record(sb)
        marks.append(("decode %d rows" % g.rows, None, b))
    pipe._stage, pipe._decode = stage, decode
import time
time.sleep(0.06)
if marks is not None and os.environ.get("PIO_TL_EVENTS") == "1":
    base.record(torch.cuda.current_stream())
t = time.perf_counter()
for _ in pipe.run((imgs, traces) for _ in range(steps)): pass
torch.cuda.synchronize()
dt = time.perf_counter() - t
time.sleep(0.06)
b = torch.empty(8888, device="cuda"); b.fill_(2.0)
torch.cuda.synchronize()
print("%d steps: %.2f ms = %.0f captions/s" % (steps, dt * 1e3, steps * 16 / dt))
for name, a, b in marks:
    print("  %-16s %s -> %6.2f ms" % (name, ("%6.2f" % base.elapsed_time(a)) if a is not None else "      ", base.elapsed_time(b)))
pipe.close()
