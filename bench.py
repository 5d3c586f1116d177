"""Headline benchmark: captions/s of the Patch-ioner hot path on MI355X (BASELINE.json config 2).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: started without a torch.distributed.run environment, ``bench.py --gpus N`` spawns its N ranks
itself (fresh child processes through ``python -m torch.distributed.run``, before anything in the parent touches the GPU)
and exits with their code.

One "step" = one Patchioner.forward over one batch of 16 synthetic 224x224 images with one 16-patch trace
region per image (caption_from=patches): ViT-B/14-reg (12 layers) -> CLS-attention read-out -> trace grids
+ weighted mean -> memory projection against the 591 753 x 768 bank -> 30-step greedy decode -> id->string.
Inputs are resident in HBM before the timed region.  Each rank runs the same per-GPU batch (weak scaling)
and the ranks all-gather the token ids once per step (RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# Kernel arguments in device memory (a ROCm launch-latency setting, read when the HIP runtime initialises; never over the user's):
# +1.2 % captions/s through the pipeline on one box, two alternations (8.58 against 8.47 k), nothing on the synchronous forward.
# The same default as patchioner_amd/__init__.py and tests/conftest.py; the effective value is reported in `config.env`.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# Eight hardware queues instead of HIP's four: the pipeline runs five streams side by side (caller, stage 1, three decodes), and whether two of them
# shared a queue depended on creation order and on a timing probe (patchioner_amd/__init__.py); same throughput when nothing is aliased.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

BATCH = 16
CROP = 224
BANK_ROWS = 591753
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
GEMM_KERNEL = "k_vit_gemm_roll / k_vit_gemm256 / k_vit_gemm"    # persistent rolling-epilogue kernel for qkv / fc1 from 704 tiles on, 256 x 256 tiles from 144, 128-row tiles below (vit_gemm.hip)
HBM_PEAK_GBS = 8000.0


def build_models(device_index: int, count: int, max_prefixes: int = 64, max_batch: int = BATCH):
    """`count` identical model instances (own workspaces / decode graphs / bank copy each) for `count` batches in flight."""
    from patchioner_amd import Patchioner, weights as W
    g = torch.Generator(device="cuda").manual_seed(6)
    bank = torch.empty(BANK_ROWS, 768, device="cuda", dtype=torch.float32)
    for s in range(0, BANK_ROWS, 65536):            # N(0,1) bank generated on the device (1.8 GB)
        e = min(BANK_ROWS, s + 65536)
        bank[s:e] = torch.randn(e - s, 768, device="cuda", generator=g)
    cfg = {"decap_weights": W.synth_decap(3), "dino_weights": W.synth_dinov2(1), "memory_bank": bank,
           "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": BANK_ROWS,
           "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": CROP, "crop_dim": CROP,
           "max_batch": max_batch, "max_prefixes": max_prefixes}
    models = [Patchioner.from_config(cfg, device="cuda:%d" % device_index) for _ in range(count)]
    del bank
    torch.cuda.empty_cache()
    return models


def make_inputs():
    import numpy as np
    import golden_cases as gc
    from patchioner_amd import weights as W
    imgs = W.synth_images(1, BATCH, CROP).cuda()
    rng = np.random.RandomState(2)
    traces = [gc.block_trace(int(rng.randint(0, 13)), int(rng.randint(0, 13))) for _ in range(BATCH)]
    return imgs, traces


def _stats(xs):
    xs = sorted(xs)
    return {"n": len(xs), "median": xs[len(xs) // 2], "p95": xs[min(len(xs) - 1, int(round(0.95 * (len(xs) - 1))))],
            "min": xs[0], "max": xs[-1]}


def cpu_baseline():
    """The oracle (a port of the reference's algorithm as executed: no KV cache, 3-pass projection, python loops) timed on
    this box's host cores: the bench workload (config 2: batch 16, traces) on all cores, 3 full-batch warm-up passes and
    10 timed steps; the same on ONE thread on a bounded sample (1 image); and BASELINE config 1 (4 images, caption_from=cls), 10 timed steps."""
    import golden_cases as gc
    import numpy as np
    from oracle import patchioner_oracle as O
    from patchioner_amd import weights as W
    from patchioner_amd.tokenizer import ClipDetokenizer
    torch.set_grad_enabled(False)
    # the threads this process can really run: a GPU box shows 256 CPUs but its cgroup grants 16 -- torch's default of 128
    # threads is then throttled (the same step takes 2-6x longer) and "cores: 128" would be a fiction
    cores = torch.get_num_threads()
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (AttributeError, OSError, ValueError):
        pass
    torch.set_num_threads(cores)
    vit = O.DinoV2Oracle(W.synth_dinov2(1), num_heads=12)
    dec = O.DeCapOracle(W.synth_decap(3))
    bank = W.synth_bank(6, BANK_ROWS)
    m = O.PatchionerOracle(vit, dec, bank, ClipDetokenizer().decode, crop_dim=CROP)
    imgs = W.synth_images(1, BATCH, CROP)
    rng = np.random.RandomState(2)
    traces = [gc.block_trace(int(rng.randint(0, 13)), int(rng.randint(0, 13))) for _ in range(BATCH)]
    for _ in range(3):                                              # >= 3 full-batch warm-up passes: thread pools, allocations
        m.forward(imgs, get_cls_capt=False, traces=traces)

    def timed(fn, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return ts

    t_all = timed(lambda: m.forward(imgs, get_cls_capt=False, traces=traces), 10)   # >= 10 timed steps: ~27 s
    t_c1 = timed(lambda: m.forward(imgs[:4], get_cls_capt=True), 10)       # BASELINE config 1: ~0.9 s a step, 10 timed steps
    torch.set_num_threads(1)
    t_one = timed(lambda: m.forward(imgs[:1], get_cls_capt=False, traces=traces[:1]), 1)
    torch.set_num_threads(cores)
    med = sorted(t_all)[len(t_all) // 2]
    return {"value": BATCH / med, "unit": "captions/s", "cores": cores, "kind": "port",
            "sample": "10 timed steps of the same workload (batch 16, 224^2, 12-layer ViT-B/14, 591753x768 fp32 bank, 30-step "
                      "cache-less decode) on torch-CPU fp32 after 3 full-batch warm-up passes: %s s (the bounded ~30 s sample); "
                      "value = 16 / median" % ", ".join("%.1f" % t for t in t_all),
            "samples_s": t_all,
            "one_thread": {"value": 1.0 / t_one[0], "unit": "captions/s", "cores": 1,
                           "sample": "1 step of 1 image / 1 trace (bounded sample), %.1f s" % t_one[0]},
            "config1_cls_batch4": {"value": 4.0 / sorted(t_c1)[len(t_c1) // 2], "unit": "captions/s", "cores": cores,
                                   "sample": "BASELINE config 1: 4 x 224^2 images, caption_from=cls, 10 timed steps after 3 warm-up passes: %s s"
                                             % ", ".join("%.1f" % t for t in t_c1)}}


def other_configs(device_index: int):
    """Per-GPU shards of BASELINE configs 3, 4 and 5 on the HIP path (rank 0, N = 1; ~40 s): one record each with the synchronous
    forward (the reference's call pattern: median / p95 over timed calls), the pipelined figure where a pipeline exists, the stage
    breakdown from HIP-event brackets and the ViT GEMM / attention rates against the MFMA peak.  `value` stays config 2.
    c3: eval_trace_captioning.py:264-330 at 518^2 with 16 regions per image as boxes; c4: eval_densecap.py:437-450 (CapDec head:
    no bank); c5: configs/mlp.viecap.k.yaml:1-31 with a ViT-L/14 backbone (synthetic GPT-2-base language model, 64 greedy steps)."""
    import numpy as np
    import golden_cases as gc
    from patchioner_amd import Patchioner, viecap as V, weights as W
    from patchioner_amd.pipeline import RegionCaptionPipeline, BoxRegions
    dev = "cuda:%d" % device_index

    def measure(m, call, n_capt, n_img, flops_img, attn_flops_img, timed=12, warm=3, pipe_factory=None, pipe_batches=24):
        for _ in range(warm):
            call()
        torch.cuda.synchronize()
        ms = []
        for _ in range(timed):
            t0 = time.perf_counter()
            call()                                   # returns python strings: the forward has completed
            ms.append((time.perf_counter() - t0) * 1e3)
        st = _stats(ms)
        m.engine.profile_enable(True)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        prof = m.engine.profile_read()
        m.engine.profile_enable(False)
        stages = {k: {"ms_per_forward": v["ms"] / 3, "launches_per_forward": v["launches"] / 3,
                      "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] and v["ms"] > 0 else None}
                  for k, v in prof.items() if v["launches"]}
        g, a = prof["vit_gemm"], prof.get("vit_attention", {"ms": 0.0, "flops": 0.0, "launches": 0})
        rec = {"forward_sync": {"captions_per_s": n_capt * 1e3 / st["median"], "ms_per_forward": st, "ms_per_image": st["median"] / n_img},
               "stages": stages,
               "vit_gemm": {"tflops": g["flops"] / (g["ms"] * 1e-3) / 1e12, "frac": g["flops"] / (g["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                            "avg_launch_us": g["ms"] * 1e3 / max(g["launches"], 1), "gflop_per_image": flops_img / 1e9,
                            "bracket": "HIP events on the launch stream, one synchronous forward at a time"},
               "vit_attention": {"tflops": a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else None,
                                 "avg_launch_us": a["ms"] * 1e3 / max(a["launches"], 1), "gflop_per_image": attn_flops_img / 1e9}}
        if pipe_factory is not None:
            pipe, feed = pipe_factory()
            list(pipe.run(feed() for _ in range(4)))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in pipe.run(feed() for _ in range(pipe_batches)):
                pass
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / pipe_batches
            rec["pipelined"] = {"captions_per_s": n_capt / dt, "ms_per_batch": dt * 1e3, "api": "RegionCaptionPipeline.run (stage 1 of the next "
                                "batch under the decode of this one; outputs identical to forward())", "batches": pipe_batches}
            pipe.close()
        return rec

    out = {}
    g = torch.Generator(device="cuda").manual_seed(6)
    bank = torch.empty(BANK_ROWS, 768, device="cuda", dtype=torch.float32)
    for s in range(0, BANK_ROWS, 65536):
        e = min(BANK_ROWS, s + 65536)
        bank[s:e] = torch.randn(e - s, 768, device="cuda", generator=g)
    # ---- config 3: talk2dino_decap, 518^2 (T = 1374), batch 8, 16 gaussian boxes per image, full bank
    B, NB, crop = 8, 16, 518
    m = Patchioner.from_config({"decap_weights": W.synth_decap(3), "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": BANK_ROWS,
                                "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": crop, "crop_dim": crop,
                                "dino_weights": W.synth_dinov2(1), "memory_bank": bank, "max_batch": B, "max_prefixes": 256}, device=dev)
    del bank
    imgs = W.synth_images(5, B, crop).cuda()
    rng = np.random.RandomState(4)
    boxes = torch.tensor(np.concatenate([rng.randint(0, 30, size=(B, NB, 2)) * 14.0, rng.randint(1, 8, size=(B, NB, 2)) * 14.0], -1), dtype=torch.float32)
    kw = dict(get_cls_capt=False, gaussian_avg=True, gaussian_bbox_variance=1.0)
    T, D = 1374, 768
    out["c3"] = measure(m, lambda: m(imgs, bboxes=boxes.clone(), **kw), B * NB, B,
                        2.0 * 1369 * 588 * D + 12 * T * 14.156e6 * (D / 768.0) ** 2, 12 * 4.0 * T * T * D,
                        pipe_factory=lambda: (RegionCaptionPipeline(m, group_batches=1, decode_clones=2),
                                              lambda: (imgs, BoxRegions(boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=1.0))))
    out["c3"]["workload"] = ("config 3 at full size: talk2dino_decap, ViT-B/14-reg 518^2 (37 x 37 patches, T = 1374), batch 8, 16 gaussian-weighted "
                             "boxes per image = 128 captions per forward, bank 591753 x 768, 30-step greedy decode")
    m.engine.close()
    del m, imgs
    torch.cuda.empty_cache()
    # ---- config 4, one GPU's shard of the 64-image batch: CapDec head (no bank), 8 images x 8 dense boxes
    B, NB, crop = 8, 8, 224
    m = Patchioner.from_config({"decap_weights": W.synth_decap(3), "dino_weights": W.synth_dinov2(1), "memory_bank": None, "prefix_size": 768,
                                "linear_talk2dino": False, "support_memory_size": 0, "dino_model": "dinov2_vitb14_reg", "normalize": True,
                                "resize_dim": crop, "crop_dim": crop, "max_batch": B, "max_prefixes": 64}, device=dev)
    imgs = W.synth_images(4, B, crop).cuda()
    rng = np.random.RandomState(4)
    b = np.concatenate([rng.randint(0, 12, size=(B, NB, 2)) * 14.0, rng.randint(1, 9, size=(B, NB, 2)) * 14.0], -1).astype(np.float32)
    b[:, -1] = [0.0, 0.0, 1.0, 1.0]                      # the dense-captioning driver's padding box (eval_densecap.py:332-333)
    boxes4 = torch.tensor(b)
    kw4 = dict(get_cls_capt=False, gaussian_avg=True, gaussian_bbox_variance=0.5)
    T = 261
    out["c4_shard"] = measure(m, lambda: m(imgs, bboxes=boxes4.clone(), **kw4), B * NB, B,
                              2.0 * 256 * 588 * D + 12 * T * 14.156e6, 12 * 4.0 * T * T * D,
                              pipe_factory=lambda: (RegionCaptionPipeline(m, group_batches=2, decode_clones=2),
                                                    lambda: (imgs, BoxRegions(boxes4.clone(), gaussian_avg=True, gaussian_bbox_variance=0.5))))
    out["c4_shard"]["workload"] = ("config 4, the per-GPU shard (64 images over 8 GPUs): talk2dino_capdec (no bank, raw region features into the "
                                   "decoder), ViT-B/14-reg 224^2, 8 images x 8 dense boxes = 64 captions per forward")
    m.engine.close()
    del m, imgs
    torch.cuda.empty_cache()
    # ---- config 5, the per-GPU shard (128 images over 8 GPUs): ViT-L/14-reg at depth 24, fp16, attention-weighted traces, ViECap head
    B, crop, D = 16, 224, 1024
    vocab, merges = W.synth_bpe(0)
    tok = V.ByteLevelBPE(vocab, merges)
    ents = list(W.SYNTH_ENTITIES)
    vc = dict(clip_hidden_size=D, weights=W.synth_viecap(311, clip_hidden_size=D, n_layer=12, tok_vocab=len(vocab)), tokenizer=tok,
              entities_text=ents, texts_embeddings=W.synth_entity_embeddings(312, len(ents), D), temperature=gc.VIECAP["temperature"],
              top_k=gc.VIECAP["top_k"], threshold=gc.VIECAP["threshold"], using_hard_prompt=True, soft_prompt_first=True, using_greedy_search=True)
    m = Patchioner.from_config({"decap_weights": None, "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 0,
                                "dino_model": "dinov2_vitl14_reg", "normalize": False, "resize_dim": crop, "crop_dim": crop,
                                "dino_weights": W.synth_dinov2(93, "dinov2_vitl14_reg"), "max_batch": B, "max_prefixes": 16, "viecap": vc,
                                "vit_dtype": "fp16"}, device=dev)
    imgs = W.synth_images(8, B, crop).cuda()
    traces = [gc.block_trace(i % 13, (5 * i) % 13) for i in range(B)]
    out["c5_shard"] = measure(m, lambda: m(imgs, get_cls_capt=False, traces=traces, use_attention_tracing=True), B, B,
                              2.0 * 256 * 588 * D + 24 * T * 14.156e6 * (D / 768.0) ** 2, 24 * 4.0 * T * T * D)
    out["c5_shard"]["workload"] = ("config 5, the per-GPU shard (128 images over 8 GPUs): ViT-L/14-reg 224^2 at depth 24 (D = 1024, 16 heads; 164.7 GFLOP "
                                   "per image), fp16 operands, batch 16, attention-weighted trace regions, ViECap head (hard + soft prompt, synthetic "
                                   "GPT-2-base, 64 greedy steps)")
    m.engine.close()
    return out


def self_launch(n_gpus: int) -> int:
    """``python bench.py --gpus N`` without a torch.distributed.run environment: start the N ranks as FRESH child processes
    (one per GPU) and return their exit code.  Nothing in this parent has touched the GPU (importing torch does not), and
    the parent never replaces itself: it waits for the child and propagates its code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_run(args) -> None:
    """PIO_BENCH_DRY=1: the N > 1 control path with no GPU work (gloo on the CPU): rendezvous, the barriers around the timed
    region, the per-step id all-gather, the MAX-over-ranks time and rank 0's ONE JSON line.  Used by tests/test_bench_launch_cpu.py
    to cover the self-launch path in a container without a GPU."""
    from patchioner_amd import dist as pdist
    import torch.distributed as dist
    os.environ.setdefault("PIO_DIST_BACKEND", "gloo")
    rank, world, _ = pdist.init_from_env("gloo")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("PIO_BENCH_DRY_FAIL") == "1" and rank == world - 1:
        raise SystemExit(3)                                   # test hook: a failing rank must fail the whole command
    ids = None
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.warmup + args.steps):
        mine = torch.full((BATCH, 30), rank * 1000 + k, dtype=torch.int32)
        ids = pdist.all_gather_equal_ids(mine)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert ids.shape == (BATCH * world, 30) and all(int(ids[BATCH * r, 0]) // 1000 == r for r in range(world))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work): control path of bench.py --gpus N", "value": BATCH * world * args.steps / dt,
                          "unit": "captions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry": True,
                          "scaling": "weak", "higher_is_better": True}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128,
                    help="timed steps (16 groups of 8 batches by default: the pipeline's fill and drain, about 5 ms, are inside the timed region)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the records of BASELINE configs 3 / 4 / 5 (`configs` in the line; ~40 s)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("PIO_BENCH_IN_FLIGHT", "8")),
                    help="batches per decode group (mode=group: stage 1 runs per bs-16 batch, ONE greedy decode serves "
                         "this many batches, <= 8 = 128 prefixes) or forwards in flight (mode=streams); 1 = the "
                         "reference's synchronous forward, which is also always measured and reported as `sync`")
    ap.add_argument("--mode", choices=["group", "streams"], default=os.environ.get("PIO_BENCH_MODE", "group"))
    ap.add_argument("--stage-streams", type=int, default=int(os.environ.get("PIO_BENCH_STAGE_STREAMS", "1")),
                    help="mode=group: model replicas (own ViT workspace, own stream) that stage 1 alternates over")
    ap.add_argument("--decode-streams", type=int, default=int(os.environ.get("PIO_BENCH_DECODE_STREAMS", "3")),
                    help="mode=group: decoders (the model's own + clones on the same weights, each with its own workspace and stream) that consecutive groups' decodes alternate over")
    ap.add_argument("--vit-batches", type=int, default=int(os.environ.get("PIO_BENCH_VIT_BATCHES", "5")),
                    help="mode=group: consecutive bs-16 batches that share one ViT launch (1 = a launch per batch)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # the driver's plain `python bench.py --gpus N`
        raise SystemExit(self_launch(args.gpus))
    if os.environ.get("PIO_BENCH_DRY") == "1":
        return dry_run(args)

    from patchioner_amd import dist as pdist
    rank, world, local = pdist.init_from_env("nccl" if args.gpus > 1 else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (torch.distributed.run started with another --nproc-per-node)" % (args.gpus, world))
    if os.environ.get("PIO_DIST_SHARE_DEVICE") == "1":      # rehearsal of the N > 1 code path on a one-GPU box (with PIO_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    torch.set_grad_enabled(False)
    import torch.distributed as dist

    P = max(1, args.in_flight)
    S = max(1, args.stage_streams)
    VB = max(1, args.vit_batches) if args.mode == "group" and P > 1 else 1       # a ViT launch may feed the tail of one decode group and the head of the next
    DS = max(1, args.decode_streams)
    # the pipeline is told the length of the stream (run(..., total=K): the last decode groups are cut at the last ViT launch);
    # PIO_BENCH_END_AWARE=0 feeds it as an endless source would
    END_AWARE = os.environ.get("PIO_BENCH_END_AWARE", "1") != "0"
    VB_BIG = 10          # the second launch size measured beside the default (`vit_launch_160`): 160 images per ViT launch
    models = build_models(local, P if args.mode == "streams" else S,
                          max_prefixes=min(256, max(64, BATCH * P)) if args.mode == "group" else 64,
                          max_batch=BATCH * (max(VB, VB_BIG) if args.mode == "group" and P > 1 else VB))
    model = models[0]
    streams = [torch.cuda.Stream() for _ in range(P)]
    imgs, traces = make_inputs()

    def step():
        outs = model(imgs, get_cls_capt=False, traces=traces, gaussian_avg=True)
        ids = pdist.all_gather_equal_ids(model.last_ids)       # the path's only exchange: final captions' ids
        return outs, ids

    pipe = None
    if args.mode == "group" and P > 1:
        from patchioner_amd.pipeline import TraceCaptionPipeline
        pipe = TraceCaptionPipeline(model, group_batches=P, stage_replicas=models[1:S], vit_batches=VB, decode_clones=DS - 1,
                                    stage_cus=int(os.environ.get("PIO_STAGE_CUS", "0")) or None,
                                    decode_cus=int(os.environ.get("PIO_DECODE_CUS", "0")) or None,
                                    eager_first=os.environ.get("PIO_BENCH_EAGER", "0") == "1")

    def run_steps(n):
        """n forwards.  mode=group: stage 1 (ViT .. projection) per batch, ONE decode per P batches, the two stages
        overlapped on two streams (pipeline.py).  mode=streams: P whole forwards in flight, model instance i % P on
        stream i % P."""
        if P == 1:
            for _ in range(n):
                outs, ids = step()
            return outs, ids
        if pipe is not None:
            caps = seen = ids = None
            for caps in pipe.run(((imgs, traces) for _ in range(n)), total=n if END_AWARE else None):
                if pipe.last_ids is not seen:          # a new group was decoded: the path's only exchange, its ids
                    seen = pipe.last_ids
                    ids = pdist.all_gather_equal_ids(seen)
            return {"trace_capts": caps}, ids[-BATCH * world:] if world == 1 else ids.view(world, -1, ids.shape[1])[:, -BATCH:].reshape(-1, ids.shape[1])
        from collections import deque
        pend = deque()
        outs = ids = None
        for i in range(n):
            if len(pend) == P:
                h = pend.popleft()
                outs = h.result()
                ids = pdist.all_gather_equal_ids(h.ids("trace_capts"))
            pend.append(models[i % P].forward_async(imgs, stream=streams[i % P], get_cls_capt=False, traces=traces,
                                                    gaussian_avg=True))
        while pend:
            h = pend.popleft()
            outs = h.result()
            ids = pdist.all_gather_equal_ids(h.ids("trace_capts"))
        return outs, ids

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(max(args.warmup, P))
    if pipe is not None:
        # untimed: every decode engine captures its graphs (a full group and the last, partial one) before the clock starts
        prime = torch.zeros(BATCH * P, 768, device="cuda")
        for eng in pipe.decode_engines:
            for rows in sorted({BATCH * P} | {BATCH * k for k in (pipe.plan_groups(args.steps) if END_AWARE else [args.steps % P])} - {0}):
                eng.decode_greedy(prime[:rows], steps=pipe.steps)
        del prime
    fence()
    t0 = time.perf_counter()
    outs, ids = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    # Per-group statistics of the pipelined path over >= 100 groups (median + p95), outside the K timed steps:
    # the interval between consecutive groups' ids becoming available on the host.
    group_ms = []
    if pipe is not None:
        n_groups = int(os.environ.get("PIO_BENCH_STAT_GROUPS", "104"))
        last_t, seen = None, None
        fence()
        for _ in pipe.run((imgs, traces) for _ in range(P * n_groups)):
            if pipe.last_ids is not seen:
                seen = pipe.last_ids
                pdist.all_gather_equal_ids(seen)
                now = time.perf_counter()
                if last_t is not None:
                    group_ms.append((now - last_t) * 1e3)
                last_t = now
        fence()
        group_ms = group_ms[3:]                 # the first intervals include the pipeline's fill
    # The reference's own call pattern: one synchronous forward at a time (no pipelining), every call timed.
    sync_steps = int(os.environ.get("PIO_BENCH_SYNC_STEPS", "100"))
    step()                                  # untimed: the 16-prefix decode graph is captured on its first call
    fence()
    sync_ms = []
    ts = time.perf_counter()
    for _ in range(sync_steps):
        t1 = time.perf_counter()
        step()                              # returns python strings: the forward has completed
        sync_ms.append((time.perf_counter() - t1) * 1e3)
    fence()
    dt_sync = time.perf_counter() - ts
    # Third region with the live HIP-event brackets on (pio_profile_*): every bracketed launch gets a (start,
    # stop) event pair on its stream.  Kept out of the regions above because the event records cost ~10 us of
    # idle per bracketed launch (85 launches per step).
    prof_steps = min(args.steps, 10)
    model.engine.profile_enable(True)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = model.engine.profile_read()
    model.engine.profile_enable(False)
    # ... and once more over the TIMED region's own launches: two groups through the pipeline with the brackets on the
    # staging engine (its ViT launches hold VB batches each and share the chip with the decodes in flight, exactly as in the
    # timed region).  The `roofline` object is taken here; the synchronous region's figure is kept as `roofline_sync`.
    prof_pipe = None
    if pipe is not None:
        model.engine.profile_enable(True)
        import math
        run_steps(math.lcm(P, VB) * max(1, (2 * P) // math.lcm(P, VB)))     # whole groups AND whole ViT launches (16 or 40 batches)
        torch.cuda.synchronize()
        prof_pipe = model.engine.profile_read()
        model.engine.profile_enable(False)
    # ... and the SAME launches (VB batches per ViT launch) with nothing else in flight: an event pair on the launch stream also brackets
    # the time a launch waits for compute units that other streams' kernels hold, so inside the pipeline it reads ~13-17 % above the
    # kernel's own duration (rocprofv3's begin -> end, profiles/r0N_vit_gemm_pipelined_by_grid.csv).  `roofline` is taken HERE -- the
    # kernel's launch duration, which is what the committed rocprofv3 summary has to agree with -- and the in-pipeline figure stays in the
    # line as `roofline_in_pipeline`.
    prof_alone = None
    if pipe is not None and VB > 1:
        big = torch.cat([imgs] * VB)
        for _ in range(2):
            model.engine.vit_forward(big, want_qkv=True)
        torch.cuda.synchronize()
        model.engine.profile_enable(True)
        for _ in range(6):
            model.engine.vit_forward(big, want_qkv=True)
        torch.cuda.synchronize()
        prof_alone = model.engine.profile_read()
        model.engine.profile_enable(False)
        del big
    # ... and the other launch size: 160 images per ViT launch (10 batches; a launch feeds the tail of one decode group and the head
    # of the next).  proj / fc2 then have two 256-tiles per workgroup, so that the second tile's multiplies run beside the first
    # one's stores (vit_gemm_roll.hip).  Its own pipeline on the same model, the same K steps timed the same way, the same two
    # brackets: `vit_launch_160`.  `value` stays the default launch size.
    big_launch = None
    if pipe is not None and VB != VB_BIG and os.environ.get("PIO_BENCH_NO_160", "0") != "1":
        from patchioner_amd.pipeline import TraceCaptionPipeline
        pipe.close()
        pipe10 = TraceCaptionPipeline(model, group_batches=P, stage_replicas=models[1:S], vit_batches=VB_BIG, decode_clones=DS - 1)

        def run10(n):
            seen = None
            for _ in pipe10.run(((imgs, traces) for _ in range(n)), total=n if END_AWARE else None):
                if pipe10.last_ids is not seen:
                    seen = pipe10.last_ids
                    pdist.all_gather_equal_ids(seen)
        import math
        run10(math.lcm(P, VB_BIG))
        prime = torch.zeros(BATCH * P, 768, device="cuda")       # untimed: the decode graphs of every group size of the timed run
        for eng in pipe10.decode_engines:
            for rows in sorted({BATCH * k for k in (pipe10.plan_groups(args.steps) if END_AWARE else [P, args.steps % P])} - {0}):
                eng.decode_greedy(prime[:rows], steps=pipe10.steps)
        del prime
        fence()
        t10 = time.perf_counter()
        run10(args.steps)
        fence()
        dt10 = time.perf_counter() - t10
        model.engine.profile_enable(True)
        run10(math.lcm(P, VB_BIG))
        torch.cuda.synchronize()
        prof10 = model.engine.profile_read()
        model.engine.profile_enable(False)
        pipe10.close()
        big = torch.cat([imgs] * VB_BIG)
        for _ in range(2):
            model.engine.vit_forward(big, want_qkv=True)
        torch.cuda.synchronize()
        model.engine.profile_enable(True)
        for _ in range(4):
            model.engine.vit_forward(big, want_qkv=True)
        torch.cuda.synchronize()
        prof10a = model.engine.profile_read()
        model.engine.profile_enable(False)
        del big
        big_launch = (dt10, prof10, prof10a)
    assert len(outs["trace_capts"]) == BATCH and ids.shape[0] == BATCH * world
    # Image transforms (the reference times them apart from inference, eval_trace_captioning.py:233-262): 16 camera-sized
    # RGB images -> [16,3,224,224] on the device (pio_preprocess: raw pixels over PCIe, resize / crop / normalise on the
    # GPU, bit-exact to PIL) beside the host PIL transform of the mirror.  Not part of `value`.
    prep = None
    if rank == 0:
        try:
            from PIL import Image
            import golden_cases as gc
            raw = [gc.prep_image(300 + i, 640, 480) for i in range(BATCH)]
            for _ in range(9):                      # every pinned staging slot of the ring allocated (one-off, ~ms each)
                model.preprocess_images(raw)
            torch.cuda.synchronize()
            per_call = []
            for _ in range(20):                     # median of synchronised calls: robust to an allocator / GC hiccup
                tp = time.perf_counter()
                dev_imgs = model.preprocess_images(raw)
                torch.cuda.synchronize()
                per_call.append(time.perf_counter() - tp)
            dt_dev = sorted(per_call)[len(per_call) // 2]
            pil = [Image.fromarray(a) for a in raw]
            tp = time.perf_counter()
            host_imgs = torch.stack([model.image_transforms(im) for im in pil])
            dt_host = time.perf_counter() - tp
            assert torch.equal(dev_imgs.cpu(), host_imgs)
            prep = {"device_ms_per_batch": dt_dev * 1e3, "device_images_per_s": BATCH / dt_dev,
                    "host_pil_ms_per_batch": dt_host * 1e3, "input": "16 x 640x480 RGB uint8 (14.7 MB over PCIe)",
                    "note": "bit-identical outputs; outside the timed region"}
        except ImportError:
            pass
    if world > 1:
        t = torch.tensor([dt, dt_sync], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_sync = float(t[0].item()), float(t[1].item())
    sync_stats = _stats(sync_ms)
    group_stats = _stats(group_ms) if group_ms else None

    if rank == 0:
        # HBM bytes per launch and MFMA utilisation come from rocprofv3 --pmc passes (separate runs of this command,
        # tools/collect_profiles.sh) committed under profiles/: REPLAYED here, not measured by this process, and dropped
        # when the committed file does not name the kernel this library ships.
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        if tj.get("gemm_kernel") != GEMM_KERNEL:
            tj = {}

        def roofline_of(g, images_per_launch, where, traffic):
            achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
            return {"kernel": "%s (fp16 MFMA 32x32x16; qkv, proj, fc1, fc2 of 12 blocks + patch embed = 49 launches per "
                              "ViT forward of %d images, %s)" % (GEMM_KERNEL, images_per_launch, where),
                    "bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_source": ("replayed from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                       "this command, %s)" % tj.get("collected", "?")) if traffic else None,
                    "mfma_util": tj.get("mfma_util"), "mfma_util_source": tj.get("mfma_util_source"),
                    "avg_launch_us": g["ms"] * 1e3 / max(g["launches"], 1),
                    "flops_per_launch": g["flops"] / max(g["launches"], 1)}
        roof_sync = roofline_of(prof["vit_gemm"], BATCH, "one synchronous forward at a time", tj.get("vit_gemm_hbm_bytes_per_launch"))
        roof = roof_sync
        roof_pipe = None
        if prof_pipe is not None and prof_pipe["vit_gemm"]["launches"] > 0:
            roof_pipe = roofline_of(prof_pipe["vit_gemm"], BATCH * VB, "inside the pipelined timed region, decodes of other streams in flight: "
                                    "the brackets include the wait for compute units", tj.get("vit_gemm_hbm_bytes_per_launch_pipelined"))
            roof = roof_pipe
        if prof_alone is not None and prof_alone["vit_gemm"]["launches"] > 0:
            roof = roofline_of(prof_alone["vit_gemm"], BATCH * VB, "the pipelined region's launch size, nothing else in flight: the kernel's own launch duration",
                               tj.get("vit_gemm_hbm_bytes_per_launch_pipelined"))
        stages = {}
        for k, v in prof.items():
            if v["launches"] == 0:
                continue
            sec = v["ms"] * 1e-3
            stages[k] = {"ms_per_step": v["ms"] / prof_steps, "launches_per_step": v["launches"] / prof_steps,
                         "tflops": v["flops"] / sec / 1e12 if v["flops"] else None,
                         "gbs": v["bytes"] / sec / 1e9 if v["bytes"] else None}
        line = {
            "metric": "captions/sec (whole node) + ms/image, ViT-B/14 224^2 bs16, 1/2/4/8 MI355X",
            "value": BATCH * world * args.steps / dt, "unit": "captions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "ms_per_image": dt / args.steps * 1e3 / BATCH,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
            "value_api": ("TraceCaptionPipeline.run (pipeline.py): the same forward, batches staged %d per ViT launch and decoded %d "
                          "per greedy decode; the reference's call pattern, one synchronous model.forward() per batch, is `forward_sync`"
                          % (VB, P)) if pipe is not None else "Patchioner.forward, one synchronous call per batch",
            "forward_sync": {"value": BATCH * world * 1e3 / sync_stats["median"], "unit": "captions/s",
                             "ms_per_forward": sync_stats, "note": "SURVEY 8d's timed region: model.forward(imgs, traces=...) on "
                             "device-resident images, id->string included; %d calls, each timed" % sync_steps},
            "pipelined_groups": {"ms_per_group": group_stats, "batches_per_group": P,
                                 "captions_per_s_at_median": BATCH * P * world * 1e3 / group_stats["median"]} if group_stats else None,
            "config": {"workload": ("config 2, bs16 traces; value: %d images per ViT launch, %d prefixes per decode (pipeline API); forward_sync: "
                                    "1 forward(16) per call. " % (BATCH * VB, BATCH * P) if pipe is not None else "config 2, bs16 traces, one forward(16) per step. ") +
                                   "talk2dino_decap_COCO, ViT-B/14-reg 224^2, batch 16/GPU, caption_from=patches "
                                   "(one 16-patch trace region per image), bank 591753x768 fp32, 30-step greedy decode; `value` is measured "
                                   "through the throughput API (value_api), the drop-in forward() figure is forward_sync",
                       "env": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
                       "arithmetic": {"vit": "fp16 operands, fp32 accumulation (v_mfma_f32_32x32x16_f16), fp32 residual stream / LayerNorm",
                                      "projection": "operands as fp16 hi + lo pairs (22 significant bits) on v_mfma_f32_16x16x32_f16, fp32 "
                                                    "accumulation and soft-max; as close to fp64 as an fp32 evaluation (DESIGN.md section 3)",
                                      "decoder": "fp32 (v_mfma_f32_16x16x4_f32) at <= 64 prefixes per decode; above, the three wide layer GEMMs on fp16 hi + lo pairs (three v_mfma_f32_16x16x32_f16 per product, fp32 accumulation; same ids on every test); greedy ids through an fp16 filter + exact fp32 re-evaluation"},
                       "global_batch": BATCH * world, "parallelism": "dp%d (image shards, ids all-gather)" % world, "batches_in_flight_per_gpu": P, "batches_per_vit_launch": VB, "concurrent_decodes": DS if args.mode == "group" else 1,
                       "pipelining": "none" if P == 1 else ("one decode per %d batches (up to %d decodes in flight, one decoder clone and stream each), overlapped with the next batches' ViT (one launch per %d batches) on %d stream(s)" % (P, DS, VB, S)
                                                           if args.mode == "group" else "%d forwards on %d streams" % (P, P))},
            "roofline": roof,
            "roofline_in_pipeline": roof_pipe,
            "roofline_sync": roof_sync,
            "vit_launch_160": None if big_launch is None else {
                "value": BATCH * world * args.steps / big_launch[0], "unit": "captions/s", "steps": args.steps,
                "note": "the same K steps through TraceCaptionPipeline with 160 images per ViT launch (10 batches); rank 0's clock",
                "roofline": roofline_of(big_launch[2]["vit_gemm"], BATCH * VB_BIG, "160 images per launch, nothing else in flight", None),
                "roofline_in_pipeline": roofline_of(big_launch[1]["vit_gemm"], BATCH * VB_BIG, "160 images per launch inside the pipelined region", None)},
            "sync": {"value": BATCH * world * sync_steps / dt_sync, "unit": "captions/s", "steps": sync_steps,
                     "ms_per_step": dt_sync / sync_steps * 1e3,
                     "note": "one forward at a time (batches_in_flight = 1), the reference eval scripts' call pattern"},
            "stages": stages,
            "preprocess": prep,
        }
        if not args.no_configs and world == 1:           # BASELINE configs 3 / 4 / 5, per-GPU shards (N = 1 only: keeps an N-rank run inside the driver's budget)
            if pipe is not None and big_launch is None:
                pipe.close()
            line["configs"] = other_configs(local)
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
