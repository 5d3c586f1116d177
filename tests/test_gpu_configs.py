"""GPU parity at the shapes of BASELINE.json's other configurations (reduced depth / batch so that the CPU
oracle finishes in seconds): config 3 (518^2, 37x37 grid, 16 boxes per image), config 4 (CapDec head = no
memory bank, many boxes per image, chunked decode), config 5's backbone (ViT-L/14, D = 1024, 16 heads)."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from parity_helpers import assert_ids_explained
from patchioner_amd import weights as W

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def O():
    from oracle import patchioner_oracle
    return patchioner_oracle


def _model(crop, with_bank, depth=2, dino="dinov2_vitb14_reg", max_batch=4, **over):
    from patchioner_amd import Patchioner
    cfg = {"decap_weights": W.synth_decap(3), "prefix_size": 768, "linear_talk2dino": False,
           "support_memory_size": 2048 if with_bank else 0, "dino_model": dino, "normalize": True,
           "resize_dim": crop, "crop_dim": crop, "dino_weights": W.synth_dinov2(91, dino, depth=depth),
           "memory_bank": W.synth_bank(61, 2048) if with_bank else None, "max_batch": max_batch}
    cfg.update(over)
    return Patchioner.from_config(cfg, device="cuda")


def _oracle_for(O, crop, with_bank, depth=2):
    from patchioner_amd.tokenizer import ClipDetokenizer
    vit = O.DinoV2Oracle(W.synth_dinov2(91, depth=depth), num_heads=12)
    dec = O.DeCapOracle(W.synth_decap(3))
    bank = W.synth_bank(61, 2048) if with_bank else None
    return O.PatchionerOracle(vit, dec, bank, ClipDetokenizer().decode, crop_dim=crop)


def test_config3_518_boxes_16_regions(O):
    """talk2dino_decap at 518^2 (T = 1374), 16 gaussian-weighted boxes per image: box features against the
    oracle on the GPU's own tokens (fp32 tolerance) and the token ids of the whole path against the oracle's
    (parity_helpers: identical, or departing at a near-tie of the top-2 logits only)."""
    B, NB, crop = 2, 16, 518
    m = _model(crop, True, max_batch=2)
    orc = _oracle_for(O, crop, True)
    imgs = W.synth_images(5, B, crop)
    rng = np.random.RandomState(4)
    xy = rng.randint(0, 30, size=(B, NB, 2)) * 14.0
    wh = rng.randint(1, 8, size=(B, NB, 2)) * 14.0 + rng.randint(0, 14, size=(B, NB, 2))
    boxes = torch.tensor(np.concatenate([xy, wh], -1), dtype=torch.float32)

    tokens, _ = m.engine.vit_forward(imgs)
    feats = m._bbox_feats(tokens, boxes.clone(), True, 1.0, False, None)
    ref = O.extract_bboxes_feats(tokens[:, 5:].cpu(), boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=1.0)
    np.testing.assert_allclose(feats.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-6)

    m.call_log, orc.call_log, orc.prefix_log = [], [], []
    got = m(imgs.cuda(), get_cls_capt=False, bboxes=boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=1.0)
    want = orc.forward(imgs.clone(), get_cls_capt=False, bboxes=boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=1.0)
    assert len(got["bbox_capts"]) == B and all(len(r) == NB for r in got["bbox_capts"])
    _, total = assert_ids_explained(orc.decoder, m.call_log, orc.call_log, "config3 (518^2, 16 boxes)", ref_prefixes=orc.prefix_log)
    assert total == B * NB


def test_config4_capdec_dense_boxes_chunked(O):
    """talk2dino_capdec (support_memory_size 0: raw, un-normalised features go to the decoder), 6 images x 24
    boxes = 144 prefixes: caption chunks of bs*bs_factor = 24 and the engine's 64-prefix decode chunks."""
    B, NB, crop = 6, 24, 224
    m = _model(crop, False, max_batch=8)
    orc = _oracle_for(O, crop, False)
    imgs = W.synth_images(6, B, crop)
    rng = np.random.RandomState(8)
    xy = rng.randint(0, 12, size=(B, NB, 2)) * 14.0
    wh = rng.randint(1, 8, size=(B, NB, 2)) * 14.0
    b = np.concatenate([xy, wh], -1).astype(np.float32)
    b[:, -3:] = [0.0, 0.0, 1.0, 1.0]                       # padding boxes of the dense-captioning driver
    boxes = torch.tensor(b)
    m.call_log, orc.call_log, orc.prefix_log = [], [], []
    got = m(imgs.cuda(), get_cls_capt=False, bboxes=boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=0.5)
    want = orc.forward(imgs.clone(), get_cls_capt=False, bboxes=boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=0.5)
    assert len(sum(got["bbox_capts"], [])) == len(sum(want["bbox_capts"], [])) == B * NB
    assert_ids_explained(orc.decoder, m.call_log, orc.call_log, "config4 (CapDec, dense boxes)", ref_prefixes=orc.prefix_log)
    # (the decoder stage alone on identical prefixes, bit-exact for all 144, is clause (1) of assert_ids_explained above)
    # config 4's multi-GPU driver on the real model (no process group = one shard; the 2-rank sharding itself is tested with
    # gloo in tests/test_dist_cpu.py): nested [B][NB] captions of the forward, the caller's boxes floor-divided in place
    from patchioner_amd import dist as pdist
    mine = boxes.clone()
    nested = pdist.sharded_box_captions(m, imgs.cuda(), mine, lambda rows: m.tokenizer.batch_captions(rows), gaussian_avg=True,
                                        gaussian_bbox_variance=0.5)
    assert nested == got["bbox_capts"] and torch.equal(mine, boxes // 14)


def test_config5_backbone_vitl14(O):
    """ViT-L/14-reg geometry (D = 1024, 16 heads; depth 2 of 24) in fp16: tokens and the qkv capture vs the oracle."""
    from patchioner_amd.engine import Engine
    sd = W.synth_dinov2(23, "dinov2_vitl14_reg", depth=2)
    e = Engine(embed_dim=1024, depth=2, num_heads=16, num_registers=4, crop_dim=224, max_batch=4, vit_dtype="fp16")
    try:
        e.load_state_dict(sd)
        e.finalize()
        imgs = W.synth_images(29, 3, 224)
        tokens, qkv = e.vit_forward(imgs)
        vit = O.DinoV2Oracle(sd, num_heads=16)
        d = vit(imgs)
        ref = torch.cat([d["x_norm_clstoken"][:, None], d["x_norm_regtokens"], d["x_norm_patchtokens"]], 1)
        err = (tokens.cpu() - ref).abs().max() / ref.abs().max()
        qerr = (qkv.cpu() - vit.last_qkv).abs().max() / vit.last_qkv.abs().max()
        print("vit-L rel-max-err %.2e, qkv %.2e" % (err, qerr))
        assert err <= 4e-3 and qerr <= 4e-3
        # read-out on the 1024-wide capture: 16 heads x 64 channels
        sa, maps, avg, dis = e.cls_attention(qkv, tokens, want_maps=True, want_avg=True, want_disentangled=True)
        rsa, rmaps = O.process_self_attention(qkv.cpu(), 3, 261, 16, 1024, 0.125, 5)
        ravg, rdis = O.attention_weighted_means(rsa, rmaps, tokens[:, 5:].cpu())
        np.testing.assert_allclose(sa.cpu().numpy(), rsa.numpy(), rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(avg.cpu().numpy(), ravg.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dis.cpu().numpy(), rdis.numpy(), rtol=1e-4, atol=1e-6)
    finally:
        e.close()


def test_nested_caption_outputs_heads_patches_registers(O):
    """get_attn_heads_capt / get_patch_capts / get_register_capts (P/src/model.py:938-975): 16 + 256 + 4
    captions per image (the largest decode batches of the path), nesting and captions vs the oracle."""
    B, crop = 1, 224
    m = _model(crop, True, max_batch=2)
    orc = _oracle_for(O, crop, True)
    imgs = W.synth_images(12, B, crop)
    kw = dict(get_cls_capt=False, get_attn_heads_capt=True, get_patch_capts=True, get_register_capts=True)
    m.call_log, orc.call_log, orc.prefix_log = [], [], []
    got = m(imgs.cuda(), **kw)
    want = orc.forward(imgs.clone(), **kw)
    assert set(got) == set(want) == {"attn_heads_capts", "patch_tokens_capts", "register_capts"}
    for key, per in (("attn_heads_capts", 16), ("patch_tokens_capts", 256), ("register_capts", 4)):
        assert len(got[key]) == B and all(len(r) == per for r in got[key])
    _, total = assert_ids_explained(orc.decoder, m.call_log, orc.call_log, "nested outputs (heads, patches, registers)", ref_prefixes=orc.prefix_log)
    assert total == B * (16 + 256 + 4)


def test_grouped_decode_pipeline_matches_synchronous_forward():
    """pipeline.TraceCaptionPipeline (stage 1 per batch, one decode per group of batches, two streams) returns
    exactly the captions of the synchronous forward, batch by batch and in order; ragged batches, a partial
    last group, attention-weighted traces and the cls path included."""
    from patchioner_amd.pipeline import TraceCaptionPipeline
    m = _model(224, True, max_batch=8)
    rng = np.random.RandomState(3)
    batches = []
    for i, b in enumerate([4, 8, 3, 8, 8, 5, 2]):
        imgs = W.synth_images(100 + i, b, 224).cuda()
        traces = [gc.block_trace(int(rng.randint(0, 13)), int(rng.randint(0, 13))) for _ in range(b)]
        batches.append((imgs, traces))
    for attn in (False, True):
        want = [m(imgs, get_cls_capt=False, traces=tr, use_attention_tracing=attn)["trace_capts"] for imgs, tr in batches]
        got = list(TraceCaptionPipeline(m, group_batches=3, use_attention_tracing=attn).run(batches))
        assert got == want
        # consecutive batches sharing one ViT launch (here cut again by max_batch = 8 inside the engine): same captions
        got = list(TraceCaptionPipeline(m, group_batches=4, use_attention_tracing=attn, vit_batches=3).run(batches))
        assert got == want
    want = [m(imgs, get_cls_capt=True)["cls_capt"] for imgs, _ in batches]
    got = list(TraceCaptionPipeline(m, group_batches=4).run((imgs, None) for imgs, _ in batches))
    assert got == want
    # consecutive groups decoded concurrently on replicas (own decoder workspace and stream each): same captions, in order
    reps = [_model(224, True, max_batch=8) for _ in range(2)]
    got = list(TraceCaptionPipeline(m, group_batches=2, vit_batches=2, decode_replicas=reps).run((imgs, None) for imgs, _ in batches))
    assert got == want
    del reps
    # ... or on decoder clones (pio_clone_decoder: the model's own weights, separate KV caches / scratch / graphs)
    pipe = TraceCaptionPipeline(m, group_batches=2, vit_batches=2, decode_clones=2)
    assert list(pipe.run((imgs, None) for imgs, _ in batches)) == want
    from patchioner_amd._lib import load
    assert load().pio_destroy(m.engine.h) < 0 and b"clones" in load().pio_last_error()    # the owner outlives its clones
    pipe.close()
    # a ViT launch longer than a decode group (5 batches per launch, 3 per decode: one launch feeds up to three groups, the
    # bench's 5-per-launch / 8-per-decode in small) and the other misaligned way round
    for gb, vb in ((3, 5), (5, 2), (2, 7)):
        pipe = TraceCaptionPipeline(m, group_batches=gb, vit_batches=vb, decode_clones=1)
        assert list(pipe.run((imgs, None) for imgs, _ in batches)) == want, (gb, vb)
        # the source's length known (run(..., total=): the last groups are cut at the last ViT launch): the same captions
        assert list(pipe.run(((imgs, None) for imgs, _ in batches), total=len(batches))) == want, (gb, vb)
        pipe.close()
    # CU-masked streams (pio_stream_create): same captions; close() releases them and the pipeline stays usable
    pipe = TraceCaptionPipeline(m, group_batches=4, stage_cus=192, decode_cus=64)
    assert list(pipe.run((imgs, None) for imgs, _ in batches)) == want
    pipe.close()
    assert list(pipe.run((imgs, None) for imgs, _ in batches[:2])) == want[:2]


def test_double_dino_boxes_vs_reference_golden(golden):
    """extract_bboxes_feats_double_dino through pio_bbox_double_dino (last block re-run per box, fp16
    operands) against the REFERENCE's outputs on the same seeded tokens / weights / boxes, every return type."""
    from patchioner_amd import Patchioner
    c = gc.DDINO
    cfg = {"decap_weights": W.synth_decap(3), "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 0,
           "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": 224, "crop_dim": 224,
           "dino_weights": W.synth_dinov2(c["seed_w"], "dinov2_vitb14_reg", depth=c["depth"]), "memory_bank": None,
           "max_batch": 4}                                   # 15 sequences -> four chunks of <= 4
    m = Patchioner.from_config(cfg, device="cuda")
    g = golden("double_dino")
    tokens = gc.ddino_tokens().cuda()
    for name, boxes in (("regular", gc.boxes_regular()), ("dummies", gc.boxes_with_dummies())):
        for use_cls in (True, False):
            for rt in ("cls", "avg", "gaussian_avg"):
                if rt == "cls" and not use_cls:
                    continue
                b = boxes.clone()
                got = m._bbox_feats_double_dino(tokens, b, use_cls, rt, c["variance"]).cpu().numpy()
                assert torch.equal(b, boxes)                # this path works on a clone: the caller's boxes survive
                want = g["%s_%s_%s" % (name, "cls" if use_cls else "nocls", rt)]
                assert np.array_equal(np.isnan(got), np.isnan(want)), (name, use_cls, rt)
                ok = ~np.isnan(want)
                tol = 2e-5 if rt == "gaussian_avg" else 4e-3 * np.abs(want[ok]).max()      # fp32 reduce / fp16 block
                err = np.abs(got[ok] - want[ok]).max()
                print("%s use_cls=%s %-12s max err %.2e (tol %.2e)" % (name, use_cls, rt, err, tol))
                assert err <= tol, (name, use_cls, rt, err)
    # routed through forward(): captions come back as [bs][n_boxes]
    imgs = W.synth_images(7, 3, 224).cuda()
    outs = m(imgs, get_cls_capt=False, bboxes=gc.boxes_regular(), double_DINO_for_bboxes=True,
             double_DINO_for_bboxes_return_type="avg", double_DINO_use_cls=True)
    assert len(outs["bbox_capts"]) == 3 and all(len(r) == 5 for r in outs["bbox_capts"])


def test_vits14_backbone_readout_and_captions(O, golden):
    """dinov2_vits14_reg (D = 384, 6 heads; the reference then reads the CLS attention out with 6 heads x 64 channels,
    model.py:336): read-out kernels vs the reference's golden outputs, backbone vs the oracle, and the whole DeCap path
    (384-wide memory bank and decoder prefix) vs the oracle's captions."""
    from patchioner_amd import Patchioner
    from patchioner_amd.engine import Engine
    from patchioner_amd.tokenizer import ClipDetokenizer
    c = gc.ATTN_VITS
    g = golden("attn_readout_vits")
    e = Engine(embed_dim=384, depth=1, num_heads=6, num_registers=4, crop_dim=224, max_batch=2, vit_dtype="fp16",
               readout_heads=6, prefix_size=384)
    try:
        e.load_state_dict(W.synth_dinov2(30, "dinov2_vits14_reg", depth=1))
        e.finalize()
        qkv, patches = gc.attn_vits_inputs()
        tokens = torch.cat([torch.zeros(c["B"], c["G"], c["D"]), patches], 1).cuda()
        sa, maps, avg, dis = e.cls_attention(qkv.cuda(), tokens, want_maps=True, want_avg=True, want_disentangled=True)
        np.testing.assert_allclose(sa.cpu().numpy(), g["self_attn"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(maps.cpu().numpy(), g["maps"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(avg.cpu().numpy(), g["avg_self_attn_token"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dis.cpu().numpy(), g["disentangled"], rtol=1e-4, atol=1e-6)
    finally:
        e.close()
    sd = W.synth_dinov2(31, "dinov2_vits14_reg", depth=2)
    dec = W.synth_decap(5, prefix_size=384)
    bank = W.synth_bank(62, 1024, 384)
    cfg = {"decap_weights": dec, "prefix_size": 384, "linear_talk2dino": False, "support_memory_size": 1024,
           "dino_model": "dinov2_vits14_reg", "normalize": True, "resize_dim": 224, "crop_dim": 224, "dino_weights": sd,
           "memory_bank": bank, "max_batch": 4}
    m = Patchioner.from_config(cfg, device="cuda")
    assert m.embed_dim == 384 and m.num_attn_heads == 6
    imgs = W.synth_images(41, 3, 224)
    tokens, qkv = m.engine.vit_forward(imgs)
    vit = O.DinoV2Oracle(sd, num_heads=6)
    d = vit(imgs)
    ref = torch.cat([d["x_norm_clstoken"][:, None], d["x_norm_regtokens"], d["x_norm_patchtokens"]], 1)
    err = (tokens.cpu() - ref).abs().max() / ref.abs().max()
    print("vit-S rel-max-err %.2e" % err)
    assert err <= 4e-3
    orc = O.PatchionerOracle(vit, O.DeCapOracle(dec), bank, ClipDetokenizer().decode, crop_dim=224, num_attn_heads=6)
    kw = dict(get_cls_capt=True, get_avg_self_attn_capt=True, traces=[gc.block_trace(3, 4), gc.block_trace(9, 2), gc.block_trace(0, 12)])
    m.call_log, orc.call_log, orc.prefix_log = [], [], []
    got, want = m(imgs.cuda(), **kw), orc.forward(imgs.clone(), **kw)
    assert set(got) == set(want)
    _, total = assert_ids_explained(orc.decoder, m.call_log, orc.call_log, "ViT-S/14 DeCap path", ref_prefixes=orc.prefix_log)
    assert total == 9


def test_ctx_cleaner_kernel_and_forward(O, golden):
    """Patchioner.ctx_cleaner on the GPU vs the reference's golden outputs, and forward(cleaning_type=...)
    (all patch tokens projected through the bank, cleaned before / after the projection, box captions without a second
    projection) vs the oracle."""
    from patchioner_amd.engine import Engine
    g = golden("ctx_cleaner")
    dirty, ctx = gc.ctx_inputs()
    e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, vit_dtype="fp16")
    try:
        for ct in ("orthogonal_projection", "contrastive_mask"):
            for alpha in (1.0, 0.35):
                got = e.ctx_clean(dirty, ctx, ct, alpha).cpu().numpy()
                np.testing.assert_allclose(got, g["%s_%g" % (ct, alpha)], rtol=1e-5, atol=1e-6)
            dn, cn = dirty / dirty.norm(dim=-1, keepdim=True), ctx / ctx.norm(dim=-1, keepdim=True)
            np.testing.assert_allclose(e.ctx_clean(dirty, ctx, ct, 0.7, normalize_inputs=True).cpu().numpy(),
                                       O.ctx_cleaner(dn, cn, cleaning_type=ct, alpha=0.7).numpy(), rtol=1e-5, atol=1e-6)
    finally:
        e.close()
    m = _model(224, True, max_batch=2)
    orc = _oracle_for(O, 224, True)
    imgs = W.synth_images(17, 2, 224)
    boxes = gc.boxes_regular()[:2, :4]
    for ct, after, cf in (("orthogonal_projection", True, "cls"), ("contrastive_mask", False, "avg_self_attn")):
        kw = dict(get_cls_capt=True, bboxes=None, cleaning_type=ct, clean_after_projection=after, alpha=0.8, clean_from=cf,
                  gaussian_avg=True, gaussian_bbox_variance=0.5)
        m.call_log, orc.call_log, orc.prefix_log = [], [], []
        got = m(imgs.cuda(), **{**kw, "bboxes": boxes.clone()})
        want = orc.forward(imgs.clone(), **{**kw, "bboxes": boxes.clone()})
        assert len(got["cls_capt"] + sum(got["bbox_capts"], [])) == len(want["cls_capt"] + sum(want["bbox_capts"], [])) == 10
        _, total = assert_ids_explained(orc.decoder, m.call_log, orc.call_log, "cleaning %s after=%s from=%s" % (ct, after, cf), ref_prefixes=orc.prefix_log)
        assert total == 10


def test_pipeline_soak_is_deterministic():
    """300 batches of the same input through the grouped-decode pipeline (staging rings, shared ViT launches, decodes of
    consecutive groups in flight on three engines, graph replays, hand-counted asynchronous loads in the decoder GEMMs
    and the LM head): every group's token ids equal the first group's."""
    from patchioner_amd.pipeline import TraceCaptionPipeline
    m = _model(224, True, max_batch=32)
    imgs = W.synth_images(55, 16, 224).cuda()
    traces = [gc.block_trace(i % 13, (3 * i) % 13) for i in range(16)]
    # the bench's shape: shared ViT launches, paired projections, three decodes in flight (the engine and two decoder clones)
    pipe = TraceCaptionPipeline(m, group_batches=4, vit_batches=2, decode_clones=2)
    first, seen, groups = None, None, 0
    for _ in pipe.run((imgs, traces) for _ in range(300)):
        if pipe.last_ids is not seen:
            seen = pipe.last_ids
            ids = seen.cpu()
            groups += 1
            if first is None:
                first = ids
                assert torch.equal(ids[:16], ids[16:32]) and torch.equal(ids[:16], ids[48:64])
            else:
                assert torch.equal(ids, first), "group %d differs" % groups
    assert groups == 75


def test_pipeline_reads_batches_produced_on_the_callers_stream_and_freed_at_once():
    """Batches that are still being produced on the caller's stream when they are handed over (asynchronous H2D copy
    plus a kernel), dropped by the caller right after, and whose memory block the caching allocator hands to a tensor
    that is immediately overwritten with NaN: the stage stream must wait for the producer and keep the block alive
    (TraceCaptionPipeline._stage: event wait + record_stream)."""
    from patchioner_amd.pipeline import TraceCaptionPipeline
    m = _model(224, True, max_batch=16)
    host = [W.synth_images(200 + i, 8, 224).pin_memory() for i in range(12)]
    traces = [gc.block_trace(i % 13, (5 * i) % 13) for i in range(8)]
    want = [m(h.cuda(), get_cls_capt=False, traces=traces)["trace_capts"] for h in host]

    def produce():
        for h in host:
            x = h.to("cuda", non_blocking=True) * 1.0
            yield x, traces
            del x
            junk = torch.empty(8, 3, 224, 224, device="cuda")       # same size: the allocator's first choice is x's block
            junk.fill_(float("nan"))
            del junk

    for vb in (1, 2):
        got = list(TraceCaptionPipeline(m, group_batches=4, vit_batches=vb).run(produce()))
        assert got == want, "vit_batches=%d" % vb


def test_caption_bboxes_crop_and_recaption_vs_oracle(O):
    """SURVEY 8f.2, Patchioner.caption_bboxes (P/src/model.py:1356-1390): every box cropped from its PIL image, transformed
    (both crop_boxes modes) on the GPU and captioned by a whole forward pass (both capt_types), against the oracle fed with
    the HOST transform of the same crops (process_bboxes, P/src/bbox_utils.py:406-421), chunk by chunk as the reference loops."""
    from PIL import Image
    from patchioner_amd.preprocess import process_bboxes
    m = _model(224, True, max_batch=4)
    orc = _oracle_for(O, 224, True)
    rng = np.random.RandomState(21)
    sizes = [(320, 240), (200, 333), (500, 375)]
    imgs = [Image.fromarray(rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)) for w, h in sizes]
    boxes = torch.tensor([[[10.0, 20.0, 150.0, 100.0], [0.0, 0.0, 320.0, 240.0], [100.5, 60.25, 80.0, 120.75], [300.0, 200.0, 60.0, 80.0]],
                          [[5.0, 5.0, 100.0, 300.0], [50.0, 100.0, 140.0, 60.0], [0.0, 0.0, 30.0, 30.0], [150.0, 250.0, 50.0, 83.0]],
                          [[0.0, 0.0, 499.0, 374.0], [250.0, 100.0, 200.0, 200.0], [20.0, 300.0, 460.0, 70.0], [400.0, 10.0, 150.0, 90.0]]])
    for crop_boxes in (False, True):
        tf = m.image_transforms if crop_boxes else m.image_transforms_no_crop
        crops = process_bboxes(imgs, boxes, tf)
        dev = m.preprocess_images([im.crop((x, y, x + w, y + h)) for im, bb in zip(imgs, boxes.tolist()) for x, y, w, h in bb],
                                  no_crop=not crop_boxes)
        assert torch.equal(dev.cpu(), crops), "device transform of the crops differs from the host transform"
        for capt_type in ("cls_capt", "avg_self_attn_capt"):
            m.call_log, orc.call_log, orc.prefix_log = [], [], []
            got = m.caption_bboxes(imgs, boxes.clone(), capt_type=capt_type, crop_boxes=crop_boxes)
            bs, nb = len(imgs), boxes.shape[1]
            want = []
            for i in range(nb):                               # the reference's chunking: bs crops per forward
                s, e = i * bs, (i * bs + bs if i < nb - 1 else crops.shape[0])
                want += orc.forward(crops[s:e].clone(), get_cls_capt=capt_type == "cls_capt",
                                    get_avg_self_attn_capt=capt_type == "avg_self_attn_capt")[capt_type]
            assert set(got) == {"bbox_capts"} and [len(r) for r in got["bbox_capts"]] == [nb] * bs
            _, total = assert_ids_explained(orc.decoder, m.call_log, orc.call_log, "caption_bboxes crop=%s %s" % (crop_boxes, capt_type), ref_prefixes=orc.prefix_log)
            assert total == bs * nb
    m.call_log = None
    with pytest.raises(KeyError):                             # the reference never asks forward() for the scores it then reads
        m.caption_bboxes(imgs, boxes.clone(), compute_scores=True)


def test_argmax_text_and_n_best_sims_from_the_h5_bank(O):
    """calculate_argmax_text (P/src/model.py:1408-1411 -> im2txtprojection.py:367-375): the bank and its caption texts come
    from a real HDF5 file (tests/golden/bank_tiny.h5 through patchioner_amd/h5lite.py), the caption is the text of the most
    similar row, return_n_best_sims adds the k largest cosines; box captions nest them as bbox_sims (model.py:1023-1041)."""
    import os
    from patchioner_amd import Patchioner
    emb, texts = gc.h5_bank_case()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bank_tiny.h5")
    cfg = {"decap_weights": None, "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 40,
           "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": 224, "crop_dim": 224,
           "dino_weights": W.synth_dinov2(91, depth=2), "memory_bank": path, "max_batch": 4, "calculate_argmax_text": True}
    m = Patchioner.from_config(cfg, device="cuda")
    assert m.engine.bank_rows == 38                                   # the two all-zero rows are dropped at load
    bank = O.load_bank_rows(torch.from_numpy(emb))
    tb = [t.encode() for t in texts]
    g = torch.Generator().manual_seed(5)
    q = torch.randn(21, 768, generator=g) * 0.1 + torch.from_numpy(emb[[1, 8, 9, 24, 39, 0, 17] * 3])
    want_caps, want_sims = O.project(q.clone(), bank, return_argmax_text=True, return_n_best_sims=3, text_dataset=tb)
    qd = q.cuda()
    caps, sims = m.caption_tokens(qd, return_n_best_sims=3)
    assert caps == want_caps
    np.testing.assert_allclose(np.array(sims), np.array(want_sims), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(qd.cpu().numpy(), (q / q.norm(dim=-1, keepdim=True)).numpy(), rtol=1e-6, atol=1e-7)   # in place
    assert m.caption_tokens(q.cuda()) == want_caps
    assert m.caption_tokens(q.cuda(), compute_scores=True) == (want_caps, [1.0] * 21)
    # through forward(): cls captions are bank texts; boxes carry bbox_sims
    imgs = W.synth_images(3, 2, 224).cuda()
    boxes = gc.e2e_boxes()[:2, :3]
    out = m(imgs, get_cls_capt=True, bboxes=boxes.clone(), gaussian_avg=True, return_n_best_sims=2)
    assert len(out["cls_capt"]) == 2 and all(c in texts for c in out["cls_capt"])           # model.py:926: no sims for cls
    assert [len(r) for r in out["bbox_capts"]] == [3, 3] and [len(r) for r in out["bbox_sims"]] == [3, 3]
    assert all(len(s) == 2 and s[0] >= s[1] for r in out["bbox_sims"] for s in r)
    with pytest.raises(ValueError):
        _model(224, True, max_batch=2)(imgs, return_n_best_sims=2)    # not a calculate_argmax_text model


@pytest.mark.parametrize("n_big", [80, 160])
def test_vit_80_images_per_launch_rolling_gemm_is_bitwise_the_16_image_launches(n_big):
    """One ViT launch of 80 (160) images sends qkv / fc1 (747 / 996 tiles of 256^2; twice that at 160) and -- round 5 -- proj / fc2
    (249 tiles = one per workgroup; 495 = two at 160, so the second tile's multiplies run beside the first one's stores) to the
    persistent kernel with the rolling epilogue (vit_gemm_roll.hip); 16-image launches use the 128- and 256-tile kernels.  Every kernel
    computes an output element with the same arithmetic -- for proj / fc2 that includes WHEN the old residual value joins the sum
    (kernels.h: resid_join_ktile) -- so the tokens and the captured qkv are the same BITS whatever a launch holds: the property the
    pipeline's shared ViT launches rely on.  Depth 3: middle blocks (rolling qkv) and the last block (qkv with the fp32 capture, the 256
    kernel)."""
    from patchioner_amd.engine import Engine
    sd = W.synth_dinov2(17, depth=3)
    big = Engine(embed_dim=768, depth=3, num_heads=12, num_registers=4, crop_dim=224, max_batch=n_big, vit_dtype="fp16")
    small = Engine(embed_dim=768, depth=3, num_heads=12, num_registers=4, crop_dim=224, max_batch=16, vit_dtype="fp16")
    for e in (big, small):
        e.load_state_dict(sd)
        e.finalize()
    imgs = W.synth_images(5, n_big, 224).cuda()
    tok_b, qkv_b = big.vit_forward(imgs)
    tok_s, qkv_s = small.vit_forward(imgs)
    torch.cuda.synchronize()
    assert torch.isfinite(tok_b).all()
    assert torch.equal(tok_b, tok_s) and torch.equal(qkv_b, qkv_s)
    tok_b2, _ = big.vit_forward(imgs)                 # and the same bits again (a race would not reproduce)
    assert torch.equal(tok_b, tok_b2)
    big.close(); small.close()


def test_residual_gemm_takes_layerscale_folded_and_any_magnitude():
    """proj / fc2 multiply by W' = op(ls W) (LayerScale folded at load, api.cpp) and take the old x into the sum as one of its terms.
    Against the oracle's fp32 blocks at depth 2 with LayerScale vectors spread over six orders of magnitude (1e-5, the DINOv2
    initial value, to 3; some negative), ragged 518^2 rows and a batch that takes the 128-wide kernels: the tokens hold the ViT
    tolerance, i.e. folding gamma into fp16 weights (subnormal for the smallest) loses nothing that matters at the residual's scale."""
    from oracle import patchioner_oracle as O
    from patchioner_amd.engine import Engine
    for crop, B in ((224, 3), (518, 2)):
        sd = W.synth_dinov2(29, depth=2)
        g = torch.Generator().manual_seed(31)
        for l in range(2):
            for k in ("ls1", "ls2"):
                e = torch.rand(768, generator=g) * 5.5 - 5.0                     # 1e-5 .. 3
                sign = torch.where(torch.rand(768, generator=g) < 0.2, -1.0, 1.0)
                sd["blocks.%d.%s.gamma" % (l, k)] = (10.0 ** e) * sign
        eng = Engine(embed_dim=768, depth=2, num_heads=12, num_registers=4, crop_dim=crop, max_batch=B, vit_dtype="fp16")
        try:
            eng.load_state_dict(sd)
            eng.finalize()
            imgs = W.synth_images(37, B, crop)
            tokens, _ = eng.vit_forward(imgs)
            d = O.DinoV2Oracle(sd, num_heads=12)(imgs)
            ref = torch.cat([d["x_norm_clstoken"][:, None], d["x_norm_regtokens"], d["x_norm_patchtokens"]], 1)
            err = float((tokens.cpu() - ref).abs().max() / ref.abs().max())
            print("LayerScale over six orders of magnitude at %d^2: rel-max-err %.2e" % (crop, err))
            assert err <= 4e-3
        finally:
            eng.close()


def test_clip_vit_backbone_decap_original_config(O):
    """The reference's "DeCap original" configuration (P/configs/decap_B16.k.yaml: dino_model vit_base_patch16_clip_224.openai,
    prefix_size 512, normalize False; P/src/model.py:358-392, 786-796, 864-865): timm OpenAI-CLIP ViT with QuickGELU, norm_pre,
    no LayerScale, the 768 -> 512 head on every token, CLIP image statistics, memory bank L2-normalised at load
    (model.py:174), no attention hook.  cls / gaussian-box / trace captions against the oracle (ClipViTOracle, itself held to
    HF's CLIPVisionModelWithProjection in the CPU suite); options that need the hook fail as in the reference."""
    from patchioner_amd import Patchioner
    from patchioner_amd.tokenizer import ClipDetokenizer
    name = "vit_base_patch16_clip_224.openai"
    vit_sd, dec_sd = W.synth_clip_vit(31, name, depth=2), W.synth_decap(33, prefix_size=512)
    bank = W.synth_bank(35, 2048, 512) * 3.0                   # not unit-norm: the load-time normalisation must happen
    cfg = {"decap_weights": dec_sd, "prefix_size": 512, "support_memory_size": 2048, "dino_model": name, "normalize": False,
           "resize_dim": 224, "crop_dim": 224, "clip_model_name": "ViT-B/16", "use_talk2dino_project": False,
           "dino_weights": vit_sd, "memory_bank": bank.clone(), "max_batch": 4}
    m = Patchioner.from_config(cfg, device="cuda")
    assert (m.backbone_type, m.patch_size, m.num_tokens, m.embed_dim, m.num_global_tokens) == ("CLIP", 16, 197, 768, 1)
    dec = O.DeCapOracle(dec_sd)
    vit = O.ClipViTOracle(vit_sd, num_heads=12, patch_size=16)
    orc = O.PatchionerOracle(vit, dec, bank / bank.norm(dim=-1, keepdim=True), ClipDetokenizer().decode, normalize=False, crop_dim=224)
    imgs = W.synth_images(37, 3, 224)
    tokens, qkv = m.engine.vit_forward(imgs, want_qkv=False)
    d = vit(imgs)
    ref = torch.cat([d["x_norm_clstoken"][:, None], d["x_norm_patchtokens"]], 1)
    err = float((tokens.cpu() - ref).abs().max() / ref.abs().max())
    print("CLIP ViT-B/16 depth 2: tokens rel-max-err %.2e" % err)
    assert qkv is None and tokens.shape == (3, 197, 512) and err <= 4e-3
    rng = np.random.RandomState(9)
    xy = rng.randint(0, 10, size=(3, 4, 2)) * 16.0
    wh = rng.randint(1, 5, size=(3, 4, 2)) * 16.0 + rng.randint(0, 16, size=(3, 4, 2))
    boxes = torch.tensor(np.concatenate([xy, wh], -1), dtype=torch.float32)
    traces = [[{"x": 0.1 + 0.05 * i, "y": 0.2 + 0.04 * i} for i in range(8)], gc.block_trace(3, 7), gc.block_trace(9, 2)]
    kw = dict(get_cls_capt=True, traces=traces, gaussian_avg=True, gaussian_bbox_variance=1.0)
    m.call_log, orc.call_log, orc.prefix_log = [], [], []
    mine = boxes.clone()
    got = m(imgs.cuda(), bboxes=mine, **kw)
    want = orc.forward(imgs.clone(), bboxes=boxes.clone(), **kw)
    assert torch.equal(mine, boxes // 16)                          # bboxes //= patch_size on the caller's tensor (bbox_utils.py:19)
    assert set(got) == set(want) == {"cls_capt", "bbox_capts", "trace_capts"}
    _, total = assert_ids_explained(dec, m.call_log, orc.call_log, "CLIP ViT-B/16 (decap_B16)", ref_prefixes=orc.prefix_log)
    assert total == 3 + 12 + 3
    for bad in (dict(get_avg_self_attn_capt=True), dict(get_attn_heads_capt=True), dict(traces=traces, use_attention_tracing=True)):
        with pytest.raises(UnboundLocalError):                     # model.py:864-872: never assigned without the hook
            m(imgs.cuda(), get_cls_capt=False, **bad)
    # the image transforms carry CLIP's statistics (model.py:377-391), on the host mirror and on the device
    from PIL import Image
    raw = gc.prep_image(301, 320, 256)
    host = m.image_transforms(Image.fromarray(raw))
    assert torch.equal(m.preprocess_images([raw]).cpu()[0], host)
    assert abs(float(host.mean()) - float((torch.from_numpy(np.asarray(Image.fromarray(raw).resize((280, 224), Image.BICUBIC).crop((28, 0, 252, 224)), dtype=np.float32).transpose(2, 0, 1) / 255.0)
                                           - torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(3, 1, 1)).div(torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(3, 1, 1)).mean())) < 1e-4


def test_clip_vit_backbone_decap_resize_config_592(O):
    """P/configs/decap_B16_resize.k.yaml: the CLIP ViT-B/16 at resize_dim = crop_dim = 592 (37 x 37 patches; P/src/model.py:371:
    timm.create_model(..., img_size=592) resamples the position table at load) -- every key of that YAML, with the tensors the
    paths would name; cls / gaussian boxes / traces against the oracle, whose ClipViTOracle restates timm's resampling and is held
    to HF CLIP at 592 in the CPU suite."""
    from patchioner_amd import Patchioner
    from patchioner_amd.tokenizer import ClipDetokenizer
    name = "vit_base_patch16_clip_224.openai"
    vit_sd, dec_sd = W.synth_clip_vit(41, name, depth=2), W.synth_decap(43, prefix_size=512)
    assert vit_sd["pos_embed"].shape == (1, 197, 768)              # the checkpoint's own 14 x 14 table
    bank = W.synth_bank(45, 2048, 512) * 2.0
    cfg = {"decap_weights": dec_sd, "prefix_size": 512, "support_memory_size": 500000, "dino_model": name, "normalize": False,
           "resize_dim": 592, "crop_dim": 592, "clip_model_name": "ViT-B/16", "use_talk2dino_project": False,
           "projection_type": "coco", "dino_weights": vit_sd, "memory_bank": bank.clone(), "max_batch": 2}
    m = Patchioner.from_config(cfg, device="cuda")
    assert (m.backbone_type, m.patch_size, m.num_tokens, m.num_patch_tokens, m.resize_dim, m.crop_dim) == ("CLIP", 16, 1370, 1369, 592, 592)
    dec = O.DeCapOracle(dec_sd)
    vit = O.ClipViTOracle(vit_sd, num_heads=12, patch_size=16)
    orc = O.PatchionerOracle(vit, dec, bank / bank.norm(dim=-1, keepdim=True), ClipDetokenizer().decode, normalize=False, crop_dim=592)
    imgs = W.synth_images(47, 2, 592)
    tokens, _ = m.engine.vit_forward(imgs, want_qkv=False)
    d = vit(imgs)
    ref = torch.cat([d["x_norm_clstoken"][:, None], d["x_norm_patchtokens"]], 1)
    err = float((tokens.cpu() - ref).abs().max() / ref.abs().max())
    print("CLIP ViT-B/16 at 592, depth 2: tokens rel-max-err %.2e" % err)
    assert tokens.shape == (2, 1370, 512) and err <= 4e-3
    rng = np.random.RandomState(11)
    xy = rng.randint(0, 30, size=(2, 5, 2)) * 16.0
    wh = rng.randint(1, 7, size=(2, 5, 2)) * 16.0 + rng.randint(0, 16, size=(2, 5, 2))
    boxes = torch.tensor(np.concatenate([xy, wh], -1), dtype=torch.float32)
    traces = [[{"x": 0.05 + 0.03 * i, "y": 0.9 - 0.02 * i} for i in range(24)], [{"x": (3 + i % 4 + 0.5) / 37, "y": (20 + i // 4 + 0.5) / 37} for i in range(16)]]
    kw = dict(get_cls_capt=True, traces=traces, gaussian_avg=True, gaussian_bbox_variance=1.0)
    m.call_log, orc.call_log, orc.prefix_log = [], [], []
    mine = boxes.clone()
    got = m(imgs.cuda(), bboxes=mine, **kw)
    want = orc.forward(imgs.clone(), bboxes=boxes.clone(), **kw)
    assert torch.equal(mine, boxes // 16)
    assert set(got) == set(want) == {"cls_capt", "bbox_capts", "trace_capts"}
    _, total = assert_ids_explained(dec, m.call_log, orc.call_log, "CLIP ViT-B/16 at 592 (decap_B16_resize)", ref_prefixes=orc.prefix_log)
    assert total == 2 + 10 + 2
    # device-side transforms at this size: shorter side to 592 (bicubic), centre crop 592, CLIP statistics
    from PIL import Image
    raw = gc.prep_image(303, 700, 640)
    assert torch.equal(m.preprocess_images([raw]).cpu()[0], m.image_transforms(Image.fromarray(raw)))


def test_box_pipeline_matches_synchronous_forward():
    """pipeline.RegionCaptionPipeline with BoxRegions (dense / region captioning, BASELINE configs 3 and 4: ViT + box weights +
    weighted means + projection of batch i+1 under the greedy decode of batch i) returns exactly forward(bboxes=...)'s nested
    [B][NB] captions, batch by batch; gaussian, uniform and attention-map weights; ragged batches; boxes mixed with trace
    batches in one stream; the caller's boxes floor-divided in place as the reference's forward does."""
    from patchioner_amd.pipeline import RegionCaptionPipeline, BoxRegions
    m = _model(224, True, max_batch=8)
    rng = np.random.RandomState(11)
    batches, raw = [], []
    for i, (b, nb) in enumerate([(4, 6), (8, 4), (3, 6), (2, 16), (5, 3)]):
        imgs = W.synth_images(200 + i, b, 224).cuda()
        xy = rng.randint(0, 12, size=(b, nb, 2)) * 14.0
        wh = rng.randint(1, 6, size=(b, nb, 2)) * 14.0 + rng.randint(0, 14, size=(b, nb, 2))
        raw.append((imgs, torch.tensor(np.concatenate([xy, wh], -1), dtype=torch.float32)))
    for kw in (dict(gaussian_avg=True, gaussian_bbox_variance=1.0), dict(), dict(use_attn_map_for_bboxes=True)):
        want = [m(imgs, get_cls_capt=False, bboxes=bx.clone(), **kw)["bbox_capts"] for imgs, bx in raw]
        mine = [bx.clone() for _, bx in raw]
        got = list(RegionCaptionPipeline(m, group_batches=2).run((imgs, BoxRegions(bx, **kw)) for (imgs, _), bx in zip(raw, mine)))
        assert got == want
        assert all(torch.equal(a, bx // 14) for a, (_, bx) in zip(mine, raw))
        pipe = RegionCaptionPipeline(m, group_batches=1, decode_clones=1)          # config 3's setting: one batch per decode
        assert list(pipe.run((imgs, BoxRegions(bx.clone(), **kw)) for imgs, bx in raw)) == want
        pipe.close()
    # boxes and traces in one stream
    traces = [gc.block_trace(2, 3)] * 4
    mixed = [(raw[0][0], BoxRegions(raw[0][1].clone(), gaussian_avg=True)), (raw[0][0], traces), (raw[2][0], BoxRegions(raw[2][1].clone()))]
    got = list(RegionCaptionPipeline(m, group_batches=3).run(mixed))
    assert got[0] == m(raw[0][0], get_cls_capt=False, bboxes=raw[0][1].clone(), gaussian_avg=True)["bbox_capts"]
    assert got[1] == m(raw[0][0], get_cls_capt=False, traces=traces)["trace_capts"]
    assert got[2] == m(raw[2][0], get_cls_capt=False, bboxes=raw[2][1].clone())["bbox_capts"]
