"""Size-independent properties at BASELINE config 2's FULL size (12-layer ViT-B/14-reg, batch 16, 591 753-row bank, 30-step
decode), where the CPU oracle is too slow to be the checker: batch-permutation equivariance, batch-composition
independence, bank-row-permutation invariance of the projection, duplicate-prefix consistency of the decoder, and the
pipelined path against the synchronous one."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from patchioner_amd import weights as W

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

M_FULL = 591753


@pytest.fixture(scope="module")
def full():
    from patchioner_amd import Patchioner
    g = torch.Generator(device="cuda").manual_seed(6)
    bank = torch.empty(M_FULL, 768, device="cuda")
    for s in range(0, M_FULL, 65536):
        bank[s:s + 65536] = torch.randn(min(65536, M_FULL - s), 768, device="cuda", generator=g)
    cfg = {"decap_weights": W.synth_decap(3), "dino_weights": W.synth_dinov2(1), "memory_bank": bank, "prefix_size": 768,
           "linear_talk2dino": False, "support_memory_size": M_FULL, "dino_model": "dinov2_vitb14_reg", "normalize": True,
           "resize_dim": 224, "crop_dim": 224, "max_batch": 16, "max_prefixes": 256}
    m = Patchioner.from_config(cfg, device="cuda")
    imgs = W.synth_images(9, 16, 224).cuda()
    traces = [gc.block_trace(int(i * 5 % 13), int(i * 7 % 13)) for i in range(16)]
    return m, bank, imgs, traces


def test_batch_permutation_and_composition(full):
    m, _, imgs, traces = full
    kw = dict(get_cls_capt=True, get_avg_self_attn_capt=True)
    base = m(imgs, traces=traces, **kw)
    assert all(len(base[k]) == 16 for k in ("cls_capt", "avg_self_attn_capt", "trace_capts"))
    perm = torch.randperm(16, generator=torch.Generator().manual_seed(1)).tolist()
    out = m(imgs[perm].contiguous(), traces=[traces[p] for p in perm], **kw)
    for k in base:
        assert out[k] == [base[k][p] for p in perm], k                  # captions follow their images
    sub = m(imgs[3:8].contiguous(), traces=traces[3:8], **kw)
    for k in base:
        assert sub[k] == base[k][3:8], k                                # ... whatever else is in the batch


def test_projection_is_invariant_to_the_order_of_the_bank(full):
    m, bank, _, _ = full
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(40, 768, device="cuda", generator=g)
    a = m.engine.project(q.clone(), normalize=True)
    from patchioner_amd.engine import Engine
    perm = torch.randperm(M_FULL, device="cuda", generator=g)
    e2 = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, max_prefixes=64)
    try:
        e2.set_memory_bank(bank[perm].contiguous())
        b = e2.project(q.clone(), normalize=True)
    finally:
        e2.close()
    # the same softmax-weighted sum in another association order: fp32 reassociation only
    assert (a - b).abs().max().item() <= 2e-5
    c = m.engine.project(q[:7].clone(), normalize=True)                  # 16-query pass vs the 32-query passes above
    assert (a[:7] - c).abs().max().item() <= 2e-5


def test_decoder_rows_are_independent_at_every_batch_size(full):
    m, _, _, _ = full
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(16, 768, device="cuda", generator=g)
    x = x / x.norm(dim=-1, keepdim=True)
    ids16, _ = m.engine.decode_greedy(x)
    for reps in (2, 4, 8, 11, 16):                                       # 32 .. 256 prefixes: other kernels / tile shapes, same rows
        ids, _ = m.engine.decode_greedy(x.repeat(reps, 1))
        assert torch.equal(ids.view(reps, 16, -1), ids16.expand(reps, -1, -1)), reps
    y = torch.randn(200, 768, device="cuda", generator=g)                # a ragged count above 128: one decode = two decodes
    y = y / y.norm(dim=-1, keepdim=True)
    whole, _ = m.engine.decode_greedy(y)
    assert torch.equal(whole, torch.cat([m.engine.decode_greedy(y[:128])[0], m.engine.decode_greedy(y[128:])[0]]))
    ids_lp, lp = m.engine.decode_greedy(x, want_logprob=True)            # exact head vs filtered head
    assert torch.equal(ids_lp, ids16) and torch.isfinite(lp).all() and (lp <= 0).all()


def test_pipeline_equals_synchronous_forward_at_full_size(full):
    from patchioner_amd.pipeline import TraceCaptionPipeline
    m, _, imgs, traces = full
    batches = [(imgs.roll(i, 0).contiguous(), traces[-i:] + traces[:-i] if i else traces) for i in range(10)]
    want = [m(b, get_cls_capt=False, traces=t)["trace_capts"] for b, t in batches]
    got = list(TraceCaptionPipeline(m, group_batches=8).run(batches))
    assert got == want


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs 2 (dense variant), 3 and 4 at FULL size: full-depth backbone, full bank, the configs' own batch shapes
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def O():
    from oracle import patchioner_oracle
    return patchioner_oracle


def _stagewise_checks(O, m, bank_cpu, imgs, boxes, variance, label):
    """One forward with gaussian-weighted boxes, every stage AFTER the backbone held to the oracle on the HIP path's own
    inputs of that stage: box features (fp32 tolerance), projection (fp32 tolerance, the full bank), greedy ids (bit-exact)."""
    B, NB = boxes.shape[:2]
    tokens, _ = m.engine.vit_forward(imgs)
    assert torch.isfinite(tokens).all()
    feats = m._bbox_feats(tokens, boxes.clone(), True, variance, False, None)
    ref = O.extract_bboxes_feats(tokens[:, m.num_global_tokens:].cpu(), boxes.clone(), gaussian_avg=True,
                                 gaussian_bbox_variance=variance)
    np.testing.assert_allclose(feats.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-6)
    m.call_log = []
    out = m(imgs, get_cls_capt=False, bboxes=boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=variance)
    assert len(out["bbox_capts"]) == B and all(len(r) == NB and all(isinstance(s, str) for s in r) for r in out["bbox_capts"])
    prefix = torch.cat([p for p, _ in m.call_log]).cpu()
    ids = torch.cat([i for _, i in m.call_log]).cpu().long()
    m.call_log = None
    assert prefix.shape[0] == B * NB
    if bank_cpu is not None:
        sub = slice(0, 16)                                   # one bank pass of the oracle: 3 x 1.8 GB on the host
        want = O.project(feats.view(-1, feats.shape[-1])[sub].cpu().clone(), bank_cpu, normalize=True)
        np.testing.assert_allclose(prefix[sub].numpy(), want.numpy(), rtol=2e-4, atol=5e-6)
    ref_ids, _, margin = O.DeCapOracle(W.synth_decap(3)).decode_ids(prefix)
    assert torch.equal(ids, ref_ids), "%s: ids differ from the oracle on identical prefixes (min margin %.2e)" % (label, float(margin.min()))
    return out


def test_config2_dense_variant_full_size(full, O):
    """SURVEY 8d C2, dense variant: bboxes [16, 1, 4] = [14 cx, 14 cy, 42, 42], gaussian_avg, variance 1.0."""
    m, bank, imgs, _ = full
    rng = np.random.RandomState(2)
    c = rng.randint(0, 13, size=(16, 2))
    boxes = torch.tensor(np.concatenate([14.0 * c, np.full((16, 2), 42.0)], 1)[:, None, :], dtype=torch.float32)
    out = _stagewise_checks(O, m, bank.cpu(), imgs, boxes, 1.0, "config 2 dense")
    perm = torch.randperm(16, generator=torch.Generator().manual_seed(5)).tolist()
    again = m(imgs[perm].contiguous(), get_cls_capt=False, bboxes=boxes[perm].clone(), gaussian_avg=True, gaussian_bbox_variance=1.0)
    assert again["bbox_capts"] == [out["bbox_capts"][p] for p in perm]


def test_config3_full_size_518_batch8_16_regions(full, O):
    """talk2dino_decap at 518^2 (37 x 37 grid, T = 1374), batch 8, 16 gaussian boxes per image = 128 captions, 12 blocks,
    the full bank."""
    from patchioner_amd import Patchioner
    _, bank, _, _ = full
    cfg = {"decap_weights": W.synth_decap(3), "dino_weights": W.synth_dinov2(1), "memory_bank": bank, "prefix_size": 768,
           "linear_talk2dino": False, "support_memory_size": M_FULL, "dino_model": "dinov2_vitb14_reg", "normalize": True,
           "resize_dim": 518, "crop_dim": 518, "max_batch": 8, "max_prefixes": 128}
    m = Patchioner.from_config(cfg, device="cuda")
    imgs = W.synth_images(3, 8, 518).cuda()
    rng = np.random.RandomState(13)
    xy = rng.randint(0, 30, size=(8, 16, 2)) * 14.0
    wh = rng.randint(1, 8, size=(8, 16, 2)) * 14.0 + rng.randint(0, 14, size=(8, 16, 2))
    boxes = torch.tensor(np.concatenate([xy, wh], -1), dtype=torch.float32)
    out = _stagewise_checks(O, m, bank.cpu(), imgs, boxes, 1.0, "config 3")
    half = m(imgs[4:].contiguous(), get_cls_capt=False, bboxes=boxes[4:].clone(), gaussian_avg=True, gaussian_bbox_variance=1.0)
    assert half["bbox_capts"] == out["bbox_capts"][4:]          # batch composition does not change a caption
    m.engine.close()


def test_config4_per_gpu_shard_full_size(O):
    """talk2dino_capdec (no bank: raw region features into the decoder), one GPU's shard of the 64-image batch = 8 images
    x 8 dense boxes, 12 blocks: the whole path against the oracle's ids through its fp32 backbone (parity_helpers bar)."""
    from parity_helpers import assert_ids_explained
    from patchioner_amd import Patchioner
    from patchioner_amd.tokenizer import ClipDetokenizer
    vit_sd, dec_sd = W.synth_dinov2(1), W.synth_decap(3)
    cfg = {"decap_weights": dec_sd, "dino_weights": vit_sd, "memory_bank": None, "prefix_size": 768, "linear_talk2dino": False,
           "support_memory_size": 0, "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": 224, "crop_dim": 224,
           "max_batch": 8, "max_prefixes": 64}
    m = Patchioner.from_config(cfg, device="cuda")
    dec = O.DeCapOracle(dec_sd)
    orc = O.PatchionerOracle(O.DinoV2Oracle(vit_sd, num_heads=12), dec, None, ClipDetokenizer().decode, crop_dim=224)
    imgs = W.synth_images(4, 8, 224)
    rng = np.random.RandomState(4)
    xy = rng.randint(0, 12, size=(8, 8, 2)) * 14.0
    wh = rng.randint(1, 9, size=(8, 8, 2)) * 14.0
    b = np.concatenate([xy, wh], -1).astype(np.float32)
    b[:, -1] = [0.0, 0.0, 1.0, 1.0]                       # the dense-captioning driver's padding box
    boxes = torch.tensor(b)
    _stagewise_checks(O, m, None, imgs.cuda(), boxes, 0.5, "config 4")
    m.call_log, orc.call_log, orc.prefix_log = [], [], []
    got = m(imgs.cuda(), get_cls_capt=False, bboxes=boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=0.5)
    want = orc.forward(imgs.clone(), get_cls_capt=False, bboxes=boxes.clone(), gaussian_avg=True, gaussian_bbox_variance=0.5)
    assert [len(r) for r in got["bbox_capts"]] == [len(r) for r in want["bbox_capts"]] == [8] * 8
    assert_ids_explained(dec, m.call_log, orc.call_log, "config 4 shard (8 x 8 boxes, CapDec, depth 12)", ref_prefixes=orc.prefix_log)
    m.engine.close()
