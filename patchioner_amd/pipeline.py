"""Throughput mode for caption_from=patches / cls over a stream of batches: two-stage software pipeline.

The reference's call pattern is one synchronous ``model(batch)`` at a time.  On an MI355X that leaves the
GPU mostly idle during the 30-step greedy decode: a decode step is a chain of ~22 small dependent kernels
whose cost barely depends on the number of prefixes (16 prefixes: 4.5 ms, 128 prefixes: 8 ms per 30 steps),
and whole forwards on independent streams overlap poorly (4 forwards on 4 streams: 3.0 k captions/s against 6.5 k
here).  So this driver

  stage 1 (stream A): per batch  ViT -> CLS read-out -> trace grids -> weighted mean -> memory projection,
                      prefixes appended to a group buffer;
  stage 2 (stream B): ONE greedy decode for the whole group (up to ``max_prefixes``: 128 = 8 batches of 16),

and stage 2 of group g runs while stage 1 of group g+1 is being computed (small decode kernels fill the CUs
the big ViT GEMMs leave).  ``vit_batches`` consecutive batches may also share one ViT launch (the backbone costs
118 us per image at 16 images per launch, 86 us at 48 and more: fewer GEMM tails and launch gaps; the model's
``max_batch`` must cover them).  The token ids are bit-identical to the synchronous path either way (every ViT
output element is computed in the same order whatever the launch holds; the decoder is exact fp32 and
row-independent); captions come back per batch, in order.  ``decode_clones`` / ``decode_replicas`` let consecutive groups decode
concurrently (each on its own decoder -- a clone on the same weights, or a replica -- and stream): next to stage 1 a decode stretches about 2x, and with three
in flight stage 1 never waits for a group buffer.
"""
from __future__ import annotations

from collections import deque
import itertools
import os
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import torch

import patchioner_amd as _pkg


# CU-masked streams (pio_stream_create) live for the life of the PROCESS, like torch's own pooled streams: torch's caching
# allocators -- device blocks with record_stream'ed uses, pinned host blocks used by a non-blocking copy -- keep the raw
# stream handle in the block and record an event on it when the block is freed, however much later; a stream destroyed in
# between is a dangling handle there (round 1's host segfault in a create / destroy loop: amd::Marker::submit under
# CachingHostAllocatorImpl::free, tests/test_gpu_zz_lifecycle.py).  close() returns the streams to this pool instead.
_CU_STREAM_POOL: dict = {}             # (device index, first CU, CUs) -> [raw stream handles not in use]


class BoxRegions:
    """Region spec of a batch for the pipeline: ``model(imgs, bboxes=..., gaussian_avg=..., gaussian_bbox_variance=...)`` --
    dense / region captioning (P/src/model.py:994-1041; BASELINE configs 3 and 4).  ``bboxes`` [B, NB, 4] xywh crop pixels; it is
    floor-divided by the patch size IN PLACE when the batch is staged, as the reference's forward does (bbox_utils.py:19).
    A batch contributes B * NB prefixes to its decode group and comes back as [B][NB] captions."""

    def __init__(self, bboxes: torch.Tensor, gaussian_avg: bool = False, gaussian_bbox_variance: float = 0.5,
                 use_attn_map_for_bboxes: bool = False):
        if bboxes.dim() != 3 or bboxes.shape[-1] != 4:
            raise ValueError("bboxes must be [B, NB, 4]")
        self.bboxes, self.gaussian_avg, self.variance = bboxes, gaussian_avg, gaussian_bbox_variance
        self.use_attn_map = use_attn_map_for_bboxes

    @property
    def n_boxes(self) -> int:
        return int(self.bboxes.shape[1])


def _rows_of(imgs, regions) -> int:
    return imgs.shape[0] * regions.n_boxes if isinstance(regions, BoxRegions) else imgs.shape[0]


class _Group:
    def __init__(self, cap: int, dim: int, steps: int, device):
        self.prefix = torch.empty(cap, dim, device=device, dtype=torch.float32)
        self.ids_host = torch.empty(cap, steps, dtype=torch.int32).pin_memory()
        self.ids_dev: Optional[torch.Tensor] = None
        self.counts: List[int] = []
        self.rows = 0
        self.flushed = 0                  # rows of `prefix` already written
        self.pending: List = []           # (embeddings, stream, model) of staged batches not yet projected
        self.staged: List[torch.cuda.Event] = []
        self.decoded = torch.cuda.Event()
        self.busy = False


class TraceCaptionPipeline:
    """``for captions in TraceCaptionPipeline(model).run(batches)`` with batches = iterable of (imgs, regions): ``regions`` = the
    traces of the batch (caption_from=patches), ``None`` (the CLS token, caption_from=cls) or a ``BoxRegions`` (dense / region
    captioning: [B][NB] captions per batch).  ``RegionCaptionPipeline`` is the same class under the name that fits boxes."""

    def __init__(self, model, group_batches: int = 4, use_attention_tracing: bool = False, steps: int = 30,
                 stage_replicas: Sequence = (), stage_cus: Optional[int] = None, decode_cus: Optional[int] = None,
                 vit_batches: int = 1, decode_replicas: Sequence = (), decode_clones: int = 0, eager_first: bool = False):
        """``stage_replicas``: further Patchioner instances holding the SAME weights (each has its own ViT
        workspace); stage 1 of consecutive batches then alternates over the replicas, each on its own stream,
        so that one batch's GEMM tails, epilogues and launch gaps are filled by the other's kernels."""
        self.m, self.eng = model, model.engine
        self.stage_models = [model] + list(stage_replicas)
        # ``decode_replicas``: further instances (same weights, own decoder workspace) so that consecutive groups'
        # decodes run concurrently, each on its own stream: a decode is a chain of ~660 small dependent kernels that
        # stretches 2x when it shares the chip with stage 1, and two chains in flight hide each other's waits.
        # ``decode_clones`` makes them here: decoders on the model's OWN weights (Engine.clone_decoder: borrowed weights, own
        # KV caches / scratch / graphs) -- no second copy of the weights or the bank, unlike whole-model replicas.
        self._own_clones = [model.engine.clone_decoder() for _ in range(max(0, int(decode_clones)))]
        self.decode_engines = [model.engine] + [getattr(r, "engine", r) for r in decode_replicas] + self._own_clones
        self.group_batches = group_batches
        # eager_first: a group is decoded early, with the batches staged so far, whenever NO decode is in flight (run()).  Off by
        # default: measured on the bench workload it LOSES (20 steps: 6.52 k against 7.44 k captions/s; 128 steps: 7.53 against 7.63 k)
        # -- the early, smaller decode shares the chip with the next ViT launches and stretches both, and the stream's last group
        # grows.  It is for sources slower than the decoders (time to first caption), not for throughput.
        self.eager_first = bool(eager_first)
        self.vit_batches = max(1, int(vit_batches))
        self._held: List = []             # batches waiting for their shared ViT launch
        # batches whose embeddings share one projection call: one 32-query bank pass serves two batches (0.44 instead
        # of 0.56 ms per batch for the projection alone).  Holding a batch back just for that gained nothing (5.24 k vs
        # 5.27 k captions/s); batches that already share a ViT launch are projected in pairs (+2 %: 6.02 k vs 5.90 k).
        # Round 4: the projection has a 48-query pass (768-thread workgroups: 0.53 ms against 0.44 for 32 and 0.37 for 16), so ALL the
        # batches of a ViT launch are projected in one call: 80 queries = 48 + 32 (0.92 ms) instead of 32 + 32 + 16 (1.25 ms).
        self.project_batches = max(1, int(os.environ.get("PIO_PROJECT_BATCHES", "0")) or self.vit_batches)
        self.use_attention_tracing = use_attention_tracing
        self.steps = steps
        # Stage 1 may be confined to the first `stage_cus` compute units so that the decode's small dependent
        # kernels find idle CUs instead of queueing behind resident GEMM workgroups (None / 0: no restriction).
        self._raw_streams = []
        # Stage 1 runs on a HIGH-PRIORITY stream -- not for the priority: HIP maps the streams of one priority class round-robin onto
        # GPU_MAX_HW_QUEUES = 4 hardware queues, and with one stage stream, three decode streams and the caller's stream two of them
        # SHARED a queue, i.e. one decode chain ran serialised with the ViT launches instead of beside them.  A stream of another
        # priority class gets a queue of its own: 8.62 against 7.95 k captions/s on one box (the same with GPU_MAX_HW_QUEUES=8
        # and default priorities; decode streams high and stage normal: 8.52 k; both high: 8.57 k).
        self.stage_streams = [self._make_stream(stage_cus, priority=int(os.environ.get("PIO_STAGE_PRIO", "-1"))) for _ in self.stage_models]
        # ... and the decode to the LAST `decode_cus` compute units (a true partition when stage_cus + decode_cus <= total)
        if decode_cus:
            self.decode_streams = [self._make_stream(decode_cus, from_top=True) for _ in self.decode_engines]
        else:
            self.decode_streams = self._concurrent_streams(len(self.decode_engines), int(os.environ.get("PIO_DECODE_PRIO", "0")),
                                                           beside=self.stage_streams)
        self.sb = self.decode_streams[0]
        self._ndecoded = 0
        self._nstaged = 0
        cap = self.eng.max_prefixes
        self.groups = [_Group(cap, self.eng.prefix_size, steps, self.eng.device) for _ in range(1 + len(self.decode_engines))]
        self.last_ids: Optional[torch.Tensor] = None

    def _concurrent_streams(self, n: int, priority: int = 0, candidates: int = 16, beside: Sequence = ()) -> List[torch.cuda.Stream]:
        """``n`` streams that the device really runs side by side.  HIP deals the streams of a priority class round-robin onto
        its few hardware queues (GPU_MAX_HW_QUEUES, default 4) in creation order -- the caller's, torch's and this library's own
        included -- and two streams that share a queue run one after the other: whether three decode streams overlap depended
        on what had been created before them (BASELINE config 3 through RegionCaptionPipeline: 6.8 or 10.3 k box captions/s for
        the same settings).  So candidates are taken from torch's pool and PROBED: a spin kernel on each stream of the set at
        once takes the time of one when they run concurrently, the sum when two share a queue (about 1 ms in all, once).
        ``beside``: streams the chosen ones must also be concurrent with (stage 1's)."""
        beside = list(beside)
        if n < 1 or (n == 1 and not beside) or os.environ.get("PIO_PROBE_STREAMS", "1") == "0" or not hasattr(torch.cuda, "_sleep"):
            return [torch.cuda.Stream(priority=priority) for _ in range(n)]
        import time
        dev = self.eng.device
        spin = 400_000                                        # cycles: ~0.2 ms

        def wall(streams):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for st in streams:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(spin)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t

        probe = torch.cuda.Stream(priority=priority)
        wall([probe])                                         # warm
        base = min(wall([probe]) for _ in range(3))
        chosen, spare, ratios = [], [], []
        for cand in [probe] + [None] * candidates:
            if len(chosen) == n:
                break
            cand = cand if cand is not None else torch.cuda.Stream(priority=priority)
            if any(cand.cuda_stream == c.cuda_stream for c in chosen + spare):
                continue                                      # the pool has wrapped around
            ratio = min(wall(beside + chosen + [cand]) for _ in range(2)) / base
            ratios.append(round(ratio, 2))
            if ratio < 1.45:
                chosen.append(cand)
            else:
                spare.append(cand)
        # what was decided, for whoever wonders about a run-dependent throughput (a busy device can make the probe misjudge): kept on the
        # object and, with PIO_PIPELINE_LOG=1, printed
        self.stream_probe = {"wanted": n, "concurrent": len(chosen), "fallback_shared_queue": max(0, n - len(chosen)),
                             "ratios_to_one_stream": ratios, "base_ms": round(base * 1e3, 3)}
        if len(chosen) < n or os.environ.get("PIO_PIPELINE_LOG") == "1":
            import sys
            print("patchioner_amd.pipeline: %d of %d decode streams run beside the stage stream(s); %d share a hardware queue "
                  "(probe ratios %s, one spin %.3f ms)%s" % (len(chosen), n, max(0, n - len(chosen)), ratios, base * 1e3,
                  "" if len(chosen) == n else " -- use fewer decode clones, or start the process with GPU_MAX_HW_QUEUES=8 (HIP reads it "
                  "once, at its first GPU call: %s)" % ("this process had initialised the GPU before patchioner_amd was imported, so the "
                                                        "package's default of 8 came too late" if _pkg._HIP_UP_AT_IMPORT else
                                                        "effective value here: %s" % os.environ.get("GPU_MAX_HW_QUEUES"))), file=sys.stderr)
        return chosen + spare[:n - len(chosen)]               # not enough independent queues: take what there is

    def _make_stream(self, n_cus, from_top: bool = False, priority: int = 0):
        if not n_cus:
            return torch.cuda.Stream(priority=priority)
        from ._lib import load, check
        import ctypes
        total = torch.cuda.get_device_properties(self.eng.device).multi_processor_count
        skip = max(0, total - int(n_cus)) if from_top else 0
        key = (self.eng.device.index or 0, skip, int(n_cus))
        free = _CU_STREAM_POOL.setdefault(key, [])
        if free:
            raw = free.pop()
        else:
            raw = ctypes.c_void_p()
            check(load().pio_stream_create(key[0], skip, int(n_cus), ctypes.byref(raw)))
        self._raw_streams.append((key, raw))
        return torch.cuda.ExternalStream(raw.value, device=self.eng.device)

    def close(self):
        """Drain, drop everything recorded on this pipeline's streams, hand the CU-masked ones back to the process-wide pool
        (they are never destroyed, see _CU_STREAM_POOL) and close the decoder clones made here."""
        torch.cuda.synchronize()
        for g in self.groups:
            g.staged = []
            g.pending = []
            g.decoded = torch.cuda.Event()
            g.ids_dev = None                  # allocator blocks keyed to the streams that are about to go
        self.last_ids = None
        # events the engines recorded on those streams (staging rings of trace_grids / preprocess) must not be waited on later
        for eng in {id(m.engine): m.engine for m in self.stage_models}.values():
            for slot in eng._stage_ring:
                slot["event"] = None
        self.stage_streams = [torch.cuda.Stream() for _ in self.stage_models]
        self.decode_streams = [torch.cuda.Stream() for _ in self.decode_engines]
        self.sb = self.decode_streams[0]
        for key, raw in self._raw_streams:
            _CU_STREAM_POOL[key].append(raw)
        self._raw_streams = []
        for c in self._own_clones:
            c.close()
        self.decode_engines = self.decode_engines[:len(self.decode_engines) - len(self._own_clones)]
        self._own_clones = []
        self.decode_streams = self.decode_streams[:len(self.decode_engines)]
        self.groups = self.groups[:1 + len(self.decode_engines)]

    # ---- stage 1: everything up to the decoder prefix, on stream A ------------------------------------
    def _stage(self, held: List) -> None:
        """One ViT launch for the held batches -- entries (imgs, traces, group): a launch may span the end of one decode group
        and the start of the next -- then per batch: read-out, region mean, projection into its group's prefix buffer."""
        k = self._nstaged % len(self.stage_models)
        self._nstaged += 1
        m, stream = self.stage_models[k], self.stage_streams[k]
        eng = m.engine
        # The caller's batches were produced on ITS stream (an asynchronous H2D copy, model.preprocess_images, ...): the
        # stage stream must not read them before that work has run, and the caching allocator must not hand their blocks
        # to the loader's next batch while the stage stream still reads them (the caller drops its reference right after).
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.eng.device))
        stream.wait_event(ready)
        for im, _, _ in held:
            if im.is_cuda:
                im.record_stream(stream)
        with torch.cuda.stream(stream):
            imgs = held[0][0] if len(held) == 1 else torch.cat([h[0] for h in held], dim=0)
            want_qkv = any((self.use_attention_tracing and h[1] is not None and not isinstance(h[1], BoxRegions)) or
                           (isinstance(h[1], BoxRegions) and h[1].use_attn_map) for h in held)
            tokens_all, qkv_all = eng.vit_forward(imgs, want_qkv=want_qkv)
            s = 0
            for im, traces, g in held:
                n = im.shape[0]
                tokens = tokens_all[s:s + n]
                if isinstance(traces, BoxRegions):
                    # extract_bboxes_feats on this batch's tokens (model._bbox_feats: boxes //= patch_size in place, host draw of
                    # the var == 0 centre cells, weights and the weighted means on the device)
                    attn = eng.cls_attention(qkv_all[s:s + n], tokens)[0] if traces.use_attn_map else None
                    emb = m._bbox_feats(tokens, traces.bboxes, traces.gaussian_avg, traces.variance, False, attn)
                    emb = emb.reshape(-1, emb.shape[-1])
                    s += n
                    g.rows += emb.shape[0]
                    g.counts.append((n, traces.n_boxes))
                    g.pending.append((emb, stream, m))
                    self._flush(g)                 # a batch of boxes is many queries already: projected on its own
                    continue
                if traces is None:
                    emb = tokens[:, 0].contiguous()
                else:
                    grids = eng.trace_grids(traces).view(n, -1)
                    if self.use_attention_tracing:
                        self_attn, _, _, _ = eng.cls_attention(qkv_all[s:s + n], tokens)
                        grids = self_attn * grids
                    emb = eng.region_reduce(tokens, grids, None, 1.0 / m.num_patch_tokens)
                s += n
                g.rows += n
                g.counts.append((n, None))
                g.pending.append((emb, stream, m))
                if len(g.pending) == self.project_batches or len(g.counts) == self.group_batches:
                    self._flush(g)

    def _flush(self, g: _Group) -> None:
        """Project (and invert) the pending region embeddings of the group together, into its prefix buffer."""
        if not g.pending:
            return
        m, stream = g.pending[-1][2], g.pending[-1][1]
        eng = m.engine
        with torch.cuda.stream(stream):
            for emb, st, _ in g.pending[:-1]:
                if st is not stream:                      # embeddings staged on another replica's stream
                    ev = torch.cuda.Event()
                    ev.record(st)
                    stream.wait_event(ev)
                    emb.record_stream(stream)
            emb = g.pending[0][0] if len(g.pending) == 1 else torch.cat([p[0] for p in g.pending], dim=0)
            pre = eng.project(emb.contiguous(), normalize=m.normalize) if m.im_proj is not None else emb
            if m.embed_inversion:
                pre = eng.revert_transformation(pre)
            n = pre.shape[0]
            g.prefix[g.flushed:g.flushed + n].copy_(pre)
            g.flushed += n
            ev = torch.cuda.Event()
            ev.record(stream)
            g.staged.append(ev)
        g.pending = []

    # ---- stage 2: one decode for the group, on stream B ---------------------------------------------------
    def _decode(self, g: _Group) -> None:
        self._flush(g)
        k = self._ndecoded % len(self.decode_engines)
        self._ndecoded += 1
        eng, sb = self.decode_engines[k], self.decode_streams[k]
        with torch.cuda.stream(sb):
            for ev in g.staged:
                sb.wait_event(ev)
            g.staged = []
            ids, _ = eng.decode_greedy(g.prefix[:g.rows], steps=self.steps)
            g.ids_dev = ids
            g.ids_host[:g.rows].copy_(ids, non_blocking=True)
            g.decoded.record(sb)
        g.busy = True

    def _collect(self, g: _Group) -> List[List[str]]:
        g.decoded.synchronize()
        rows = g.ids_host[:g.rows].tolist()
        self.last_ids = g.ids_dev
        out, s = [], 0
        for n, nb in g.counts:
            k = n if nb is None else n * nb
            caps = self.m.tokenizer.batch_captions(rows[s:s + k], decoding_method=self.m.decoding_method)
            if nb is not None and caps is not None:        # [B][NB], as forward()'s bbox_capts (model.py:1037-1041)
                caps = [caps[i * nb:(i + 1) * nb] for i in range(n)]
            out.append(caps)
            s += k
        g.rows, g.counts, g.busy, g.flushed = 0, [], False, 0
        return out

    def _groups_cut_at(self, total: int, cut: int) -> List[int]:
        sizes, n = [], 0
        for i in range(int(total)):
            n += 1
            if n == self.group_batches or i + 1 == cut:
                sizes.append(n)
                n = 0
        return sizes + ([n] if n else [])

    def _last_launch_start(self, total: Optional[int]) -> int:
        """Where run(..., total=) closes the open group early: the index of the first batch of the stream's LAST ViT launch -- or 0
        (no cut) when the length is unknown, when there is one launch in all, or when the cut would make one decode more (a decode
        costs nearly the same for 32 prefixes as for 128: measured with 10 batches per launch and 20 in all, groups of 8 / 2 / 8 / 2
        gave 7.7 k captions/s where 8 / 8 / 4 give 8.3 k)."""
        if not total:
            return 0
        cut = self.vit_batches * ((int(total) - 1) // self.vit_batches)
        return cut if cut and len(self._groups_cut_at(total, cut)) <= len(self._groups_cut_at(total, 0)) else 0

    def plan_groups(self, total: int) -> List[int]:
        """Batches per decode group, in order, for a stream of ``total`` equal batches (what ``run(..., total=total)`` does;
        callers use it to capture the decode graphs of those sizes ahead of time)."""
        return self._groups_cut_at(total, self._last_launch_start(total))

    def run(self, batches: Iterable[Tuple[torch.Tensor, Optional[Sequence]]], total: Optional[int] = None,
            plan: Optional[Sequence[int]] = None) -> Iterator[List[str]]:
        """Batches are dealt to decode groups of ``group_batches`` (or as many as fit ``max_prefixes``) in order; independently
        of that, every ``vit_batches`` consecutive batches share one ViT launch (a launch may feed the tail of one group and
        the head of the next: 5 batches of 16 are 83 row tiles x 3 = 249 workgroups on 256 CUs for the N = 768 GEMMs, where 4
        batches leave 58 CUs idle).  A group is decoded as soon as the launch holding its last batch has been staged.

        ``total``: the number of batches the source will yield, when the caller knows it (a dataset's length).  The end of a
        stream is then cut differently: the open group closes where the LAST ViT launch begins, so that no decode waits for
        that launch except the one of its own batches -- with 20 batches, 5 per launch and 8 per group the groups are 8 / 7 / 5
        instead of 8 / 8 / 4, and the second one decodes beside the last launch instead of after it (only where that does not
        take one decode more: _last_launch_start).  The captions do not depend on the grouping (decoder rows are independent);
        a wrong ``total`` costs time only.  ``plan`` (instead of ``total``): the batches per decode group, in order, given outright
        (a group still closes when it is full); past its end the default rule applies."""
        cuts = {self._last_launch_start(total)} if plan is None else set(itertools.accumulate(int(k) for k in plan))
        cuts.discard(0)
        i_batch = 0
        cur = 0
        pending = deque()          # groups whose decode is in flight, oldest first
        closing: List[_Group] = []  # complete groups whose last batches are still held for their ViT launch
        n_assigned, rows_assigned = 0, 0

        def launch_held():
            nonlocal cur, n_assigned, rows_assigned
            if self._held:
                self._stage(self._held)
                self._held = []
            # Nothing is decoding (the start of a stream, or a source slower than the decoders) and the open group has staged
            # batches: decode them now rather than keep every decoder idle until the group is full.  In the steady state a decode
            # is always in flight and groups fill up as before; the captions do not depend on the grouping either way.
            if (getattr(self, "eager_first", False) and not closing and n_assigned
                    and all(gg.decoded.query() for gg in pending)):
                closing.append(self.groups[cur])
                cur = (cur + 1) % len(self.groups)
                n_assigned, rows_assigned = 0, 0
            for gg in closing:
                self._decode(gg)
                pending.append(gg)
            del closing[:]

        aligned = self.vit_batches <= self.group_batches and self.group_batches % self.vit_batches == 0
        for imgs, traces in batches:
            g = self.groups[cur]
            n = _rows_of(imgs, traces)
            if n > g.prefix.shape[0]:
                raise ValueError("batch of %d does not fit the %d-prefix decode group" % (n, g.prefix.shape[0]))
            if n_assigned and rows_assigned + n > g.prefix.shape[0]:    # a larger batch than the last one: close the group first
                closing.append(g)
                cur = (cur + 1) % len(self.groups)
                n_assigned, rows_assigned = 0, 0
                g = self.groups[cur]
                if aligned:
                    launch_held()
            if n_assigned == 0:
                if any(g is c for c in closing):   # a launch longer than all the group buffers together: send what is held
                    launch_held()
                while g.busy:      # its previous decode must be collected before the buffer is reused
                    for caps in self._collect(pending.popleft()):
                        yield caps
            self._held.append((imgs, traces, g))
            n_assigned += 1
            rows_assigned += n
            i_batch += 1
            if n_assigned == self.group_batches or rows_assigned + n > g.prefix.shape[0] or i_batch in cuts:
                closing.append(g)
                cur = (cur + 1) % len(self.groups)
                n_assigned, rows_assigned = 0, 0
            # (aligned settings keep the old behaviour: a group's last launch goes out with its last batch)
            if len(self._held) == self.vit_batches or (closing and aligned):
                launch_held()
        launch_held()
        g = self.groups[cur]
        if n_assigned and g.rows:
            self._decode(g)
            pending.append(g)
        while pending:
            for caps in self._collect(pending.popleft()):
                yield caps


RegionCaptionPipeline = TraceCaptionPipeline
