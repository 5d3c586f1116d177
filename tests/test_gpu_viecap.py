"""ViECap head on the GPU (BASELINE config 5): the C-ABI pieces against the REFERENCE's own outputs
(tests/golden/viecap.npz, tools/oracle/gen_golden_viecap.py) and the oracle, then the whole head behind
Patchioner.caption_tokens / forward, and a config-5-shaped run (ViT-L/14, fp16, 16 images, attention-weighted traces).

Bars: mapping network and entity probabilities fp32 round-off (exact-fp32 MFMA, another summation order than ATen);
hard-prompt token ids exact; greedy ids bit-exact against the oracle on the HIP path's own prompt, and against the
reference's ids through the parity_helpers bar (identical, or departing at a near-tie only)."""
import json

import numpy as np
import pytest
import torch

import golden_cases as gc
from parity_helpers import MARGIN_BOUND
from patchioner_amd import viecap as V
from patchioner_amd import weights as W

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def O():
    from oracle import patchioner_oracle
    return patchioner_oracle


@pytest.fixture(scope="module")
def case():
    w, (vocab, merges), ents, emb, x = gc.viecap_case()
    return w, V.ByteLevelBPE(vocab, merges), ents, emb, x


def _viecap_cfg(w, tok, ents, emb, C, **over):
    c = gc.VIECAP
    cfg = dict(clip_hidden_size=C, weights=w, tokenizer=tok, entities_text=ents, texts_embeddings=emb, temperature=c["temperature"],
               top_k=c["top_k"], threshold=c["threshold"], using_hard_prompt=True, soft_prompt_first=True, using_greedy_search=True)
    cfg.update(over)
    return cfg


def _model(viecap_cfg, dino="dinov2_vitb14_reg", depth=2, **over):
    from patchioner_amd import Patchioner
    cfg = {"decap_weights": None, "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 0, "dino_model": dino,
           "normalize": False, "resize_dim": 224, "crop_dim": 224, "dino_weights": W.synth_dinov2(91, dino, depth=depth),
           "max_batch": 16, "max_prefixes": 16, "viecap": viecap_cfg}
    cfg.update(over)
    return Patchioner.from_config(cfg, device="cuda")


def _explained(orc_margins, got, ref, label):
    worst = 0.0
    for r in (got != ref).any(dim=1).nonzero().flatten().tolist():
        t = int((got[r] != ref[r]).nonzero()[0])
        m = float(orc_margins[r, t])
        worst = max(worst, m)
        assert m <= MARGIN_BOUND, "%s: row %d departs at step %d where the top-2 margin is %.3e" % (label, r, t, m)
    print("%s: %d / %d id rows identical, largest margin at a departure %.2e" % (label, int((got == ref).all(dim=1).sum()), got.shape[0], worst))


def test_viecap_pieces_vs_reference_and_oracle(O, golden, case):
    g = golden("viecap")
    meta = json.loads(bytes(g["meta_json"]).decode())
    w, tok, ents, emb, x = case
    c = gc.VIECAP
    m = _model(_viecap_cfg(w, tok, ents, emb, c["C"]))
    eng = m.engine
    xd = x.cuda().contiguous()
    cont = eng.viecap_mapping(xd)
    np.testing.assert_allclose(xd.cpu().numpy(), (x / x.norm(dim=-1, keepdim=True)).numpy(), rtol=1e-6, atol=1e-7)     # in place
    np.testing.assert_allclose(cont.cpu().numpy(), g["cont"], rtol=2e-4, atol=5e-5)
    probs = eng.viecap_entity_logits(xd, c["temperature"])
    np.testing.assert_allclose(probs.cpu().numpy(), g["entity_probs"], rtol=2e-4, atol=1e-6)
    # whole head: captions, hard-prompt ids, greedy ids
    caps = m.caption_tokens(x.clone().cuda())
    assert np.array_equal(m.viecap.last_prompt_tokens.numpy(), g["prompt_tokens"])
    ids = m.viecap.last_ids.cpu().long()
    orc = O.ViECapOracle(w, tok, ents, emb, temperature=c["temperature"], top_k=c["top_k"], threshold=c["threshold"],
                         using_hard_prompt=True, soft_prompt_first=True)
    # (1) bit-exact against the oracle's search on the HIP path's OWN prompt (soft prompt from the GPU, same hard prompt)
    wte = orc.gpt.w["decoder.transformer.wte.weight"]
    emb_in = torch.cat((cont.cpu(), wte[torch.from_numpy(g["prompt_tokens"]).long()]), dim=1)
    o_ids, o_margin = [], []
    logits, past = orc.gpt.gpt2_logits_cached(emb_in)
    for step in range(64):
        nxt = logits.argmax(-1, keepdim=True)
        t2 = logits.topk(2, dim=-1).values
        o_ids.append(nxt); o_margin.append((t2[:, 0] - t2[:, 1])[:, None])
        if step < 63:
            logits, past = orc.gpt.gpt2_logits_cached(wte[nxt], past)
    o_ids, o_margin = torch.cat(o_ids, 1), torch.cat(o_margin, 1)
    assert torch.equal(ids, o_ids), "greedy ids differ from the oracle on an identical prompt (min margin %.2e)" % float(o_margin.min())
    # (2) against the reference's ids (they end at the first full stop): identical or a near-tie
    ref = torch.from_numpy(g["decoded_ids"]).long()
    n = (ref >= 0).sum(1)
    got_cut = ids.clone()
    for r in range(ids.shape[0]):
        got_cut[r, n[r]:] = -1
    _explained(o_margin, got_cut, ref, "ViECap ids vs the reference")
    if all(torch.equal(got_cut[r], ref[r]) for r in range(ids.shape[0])):
        assert caps == meta["sentences"]
    one = m.caption_tokens(x[1:2].clone().cuda())
    assert isinstance(one, str)                                   # search.py:172-181: a single feature gives a str
    ids1 = m.viecap.last_ids.cpu().long()
    want1 = orc.forward(x[1:2].clone())                           # batch 1 may legitimately part from the batch-6 row at a near-tie
    _explained(orc.last["margins"], ids1, orc.last["ids"], "ViECap single feature vs the oracle")
    ref1 = torch.from_numpy(g["single_decoded_ids"]).long()[None]
    cut1 = ids1[:, :ref1.shape[1]]
    _explained(orc.last["margins"][:, :ref1.shape[1]], cut1, ref1, "ViECap single feature vs the reference")
    if torch.equal(cut1, ref1):
        assert one == meta["single_sentence"] == want1
    with pytest.raises(Exception):
        m.caption_tokens(x.clone().cuda(), return_n_best_sims=2)
    # compute_scores: VieCap.compute_perplexity (entrypoint.py:155-172) -- the teacher-forced pass (pio_lm_score, exact fp32
    # head) on the reference's own sentences against the reference's perplexities, and through caption_tokens
    ppl = m.viecap.compute_perplexity(meta["perplexity_sentences"])
    np.testing.assert_allclose(ppl, g["perplexity"], rtol=5e-4)
    np.testing.assert_allclose(ppl, orc.compute_perplexity(meta["perplexity_sentences"]), rtol=5e-4)
    caps2, scores = m.caption_tokens(x.clone().cuda(), compute_scores=True)
    assert caps2 == caps and len(scores) == len(caps)
    np.testing.assert_allclose(scores, orc.compute_perplexity(caps), rtol=5e-4)
    ragged = ["a dog.", "c", " traffic light and a bird on the road", ""]
    got_r, want_r = m.viecap.compute_perplexity(ragged), orc.compute_perplexity(ragged)
    for a_, b_ in zip(got_r, want_r):
        assert (np.isnan(a_) and np.isnan(b_)) or abs(a_ - b_) <= 5e-4 * abs(b_), (got_r, want_r)


def test_viecap_batched_prompt_prefill_gives_the_ids_of_the_position_by_position_path(case, monkeypatch):
    """Round 5: the prompt of a ViECap decode goes through the layers as ONE batch of N x P rows (decoder.hip: dec_prefill_layers: the layer
    GEMMs on 128 rows per launch, the keys / values of every position appended first) instead of P passes over N rows.  Same products,
    another order of the fp32 sums inside a GEMM row, so the cache differs at rounding level; the ids of both paths are held to each
    other here (32 prefixes with prompts of different lengths, padded as the reference pads them: pad id 0, no mask) and, in
    test_viecap_pieces_vs_reference_and_oracle, to the reference's own fixture.  PIO_DEC_PREFILL is read when an engine is created."""
    w, tok, ents, emb, x = case
    c = gc.VIECAP
    feats = torch.cat([x, gc.randn(909, 26, c["C"])]).cuda()
    ids = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PIO_DEC_PREFILL", mode)
        m = _model(_viecap_cfg(w, tok, ents, emb, c["C"]), max_batch=4, max_prefixes=32)
        caps = m.caption_tokens(feats.clone())
        ids[mode] = (m.viecap.last_ids.cpu().clone(), m.viecap.last_prompt_tokens.clone(), caps)
        m.engine.close()
    assert torch.equal(ids["1"][1], ids["0"][1]) and ids["1"][1].shape[0] == 32
    assert torch.equal(ids["1"][0], ids["0"][0]) and ids["1"][2] == ids["0"][2]


def test_viecap_soft_prompt_only_and_routing_through_forward(O, case):
    """using_hard_prompt False (the soft prompt alone) against the oracle, and forward(): cls + trace captions come from the head."""
    w, tok, ents, emb, x = case
    c = gc.VIECAP
    m = _model(_viecap_cfg(w, tok, ents, emb, c["C"], using_hard_prompt=False))
    orc = O.ViECapOracle(w, tok, ents, emb, using_hard_prompt=False)
    want = orc.forward(x[:3].clone())
    got = m.caption_tokens(x[:3].clone().cuda())
    ids = m.viecap.last_ids.cpu().long()
    _explained(orc.last["margins"], ids, orc.last["ids"], "ViECap soft prompt only vs the oracle")
    if torch.equal(ids, orc.last["ids"]):
        assert got == want
    imgs = W.synth_images(5, 3, 224).cuda()
    out = m(imgs, get_cls_capt=True, traces=[gc.block_trace(1, 2), gc.block_trace(8, 8), gc.block_trace(12, 0)])
    assert len(out["cls_capt"]) == 3 and len(out["trace_capts"]) == 3 and all(isinstance(s, str) for s in out["cls_capt"] + out["trace_capts"])


def test_viecap_only_hard_prompt_vs_the_oracle(O, case):
    """only_hard_prompt (entrypoint.py:130-131): the prompt is the word embeddings of the entity sentence alone -- no soft
    prompt in the sequence (pio_viecap_decode with cont == NULL); prompt tokens and ids against the oracle."""
    w, tok, ents, emb, x = case
    c = gc.VIECAP
    kw = dict(temperature=c["temperature"], top_k=c["top_k"], threshold=c["threshold"], using_hard_prompt=True)
    m = _model(_viecap_cfg(w, tok, ents, emb, c["C"], only_hard_prompt=True, **kw))
    orc = O.ViECapOracle(w, tok, ents, emb, only_hard_prompt=True, **kw)
    want = orc.forward(x[:4].clone())
    got = m.viecap.forward(x[:4].clone())
    assert np.array_equal(m.viecap.last_prompt_tokens.numpy(), orc.last["prompt_tokens"].numpy())
    ids = m.viecap.last_ids.cpu().long()
    _explained(orc.last["margins"], ids, orc.last["ids"], "ViECap only_hard_prompt vs the oracle")
    if torch.equal(ids, orc.last["ids"]):
        assert got == want
    # the C-ABI refuses a call with neither kind of prompt
    with pytest.raises(Exception):
        m.engine.viecap_decode(None, None)


def test_viecap_beam_search_vs_reference_and_oracle(O, golden, case):
    """using_greedy_search False: VieCap.forward calls beam_search per image (entrypoint.py:143-148, search.py:193-285).  The beams
    of the HIP path (pio_lm_prefill / pio_lm_advance / pio_beam_select + the reference's bookkeeping in viecap.py) against the
    REFERENCE's (tests/golden/viecap_beam.npz): every beam's ids and the returned order, for the reference's own end-of-sentence
    strings and for two the seeded model emits (beams stop at different steps).  A departure must sit at a selection whose
    margin (last chosen vs first rejected candidate, from the oracle) is inside the parity bound."""
    g = golden("viecap_beam")
    meta = json.loads(bytes(g["meta_json"]).decode())
    w, tok, ents, emb, x = case
    c = gc.VIECAP
    m = _model(_viecap_cfg(w, tok, ents, emb, c["C"], using_greedy_search=False, beam_width=5))
    orc = O.ViECapOracle(w, tok, ents, emb, temperature=c["temperature"], top_k=c["top_k"], threshold=c["threshold"],
                         using_hard_prompt=True, soft_prompt_first=True)
    prompts_ref = orc.prompt_embeddings(x[:3].clone())
    head = m.viecap
    xd = x[:3].clone().cuda()
    cont = m.engine.viecap_mapping(xd)
    # the prompt tokens as forward() builds them
    head.forward(x[:1].clone())                 # also exercises the routing: one beam search, a list with one sentence
    probs = m.engine.viecap_entity_logits(xd, c["temperature"]).cpu()
    rows = [tok.encode(V.compose_discrete_prompt_text(V.top_k_entities(ents, probs[i], c["top_k"], c["threshold"]))) for i in range(3)]
    L = max(len(r) for r in rows)
    pad = tok.pad_token_id if tok.pad_token_id is not None else 0
    tokens = torch.full((3, L), pad, dtype=torch.int32)
    for i, r in enumerate(rows):
        tokens[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
    prompts = m.engine.viecap_build_prompt(cont, tokens, soft_first=True)
    np.testing.assert_allclose(prompts.cpu().numpy(), prompts_ref.numpy(), rtol=1e-4, atol=2e-5)
    exact = 0
    for call in meta["calls"]:
        eos = (".", " .") if call["label"] == "default" else tuple(meta["eos_strings"])
        i = call["image"]
        got = head.beam_search(prompts[i:i + 1], 5, end_of_sentences=eos)
        ref = g["%s_%d_ids" % (call["label"], i)]
        same = all(head.last_beams[b][0] == ref[b, :int((ref[b] >= 0).sum())].tolist() for b in range(5))
        if same:
            assert got == call["sentences"], (call["label"], i)
            exact += 1
        else:
            orc.beam_search(prompts_ref[i:i + 1], beam_width=5, end_of_sentences=eos)
            worst = min(orc.last_beam["margins"])
            assert worst <= MARGIN_BOUND, "%s image %d: beams differ although every selection margin is >= %.2e" % (call["label"], i, worst)
    print("beam search: %d / %d calls identical to the reference (ids of all 5 beams and the order returned)" % (exact, len(meta["calls"])))
    assert exact >= len(meta["calls"]) - 1
    out = m.caption_tokens(x[:2].clone().cuda())
    assert isinstance(out, list) and len(out) == 2 and all(isinstance(s, str) for s in out)


def test_config5_shape_vitl14_fp16_attention_weighted_traces(O):
    """BASELINE config 5 on one GPU's shard: ViT-L/14 (D = 1024, 2 of its 24 blocks here), fp16 operands, 16 images,
    attention-weighted trace regions into the ViECap head (clip_hidden_size 1024, entity matrix [K, 1024]): the decoder
    stage bit-exact against the oracle on the HIP path's own region features."""
    vocab, merges = W.synth_bpe(0)
    tok = V.ByteLevelBPE(vocab, merges)
    w = W.synth_viecap(311, clip_hidden_size=1024, n_layer=12, tok_vocab=len(vocab))
    ents = list(W.SYNTH_ENTITIES)
    emb = W.synth_entity_embeddings(312, len(ents), 1024)
    m = _model(_viecap_cfg(w, tok, ents, emb, 1024), dino="dinov2_vitl14_reg", depth=2, vit_dtype="fp16")
    assert m.embed_dim == 1024
    imgs = W.synth_images(6, 16, 224).cuda()
    traces = [gc.block_trace(i % 13, (7 * i) % 13) for i in range(16)]
    eng = m.engine
    tokens, qkv = eng.vit_forward(imgs)
    self_attn, _, _, _ = eng.cls_attention(qkv, tokens)
    feats = eng.region_reduce(tokens, self_attn * eng.trace_grids(traces).view(16, -1), None, 1.0 / m.num_patch_tokens)
    out = m(imgs, get_cls_capt=False, traces=traces, use_attention_tracing=True)
    assert len(out["trace_capts"]) == 16 and all(isinstance(s, str) for s in out["trace_capts"])
    ids = m.viecap.last_ids.cpu().long()
    orc = O.ViECapOracle(w, tok, ents, emb, temperature=0.01, top_k=3, threshold=0.4, using_hard_prompt=True, soft_prompt_first=True)
    want = orc.forward(feats.cpu().clone())
    assert np.array_equal(m.viecap.last_prompt_tokens.numpy(), orc.last["prompt_tokens"].numpy())
    _explained(orc.last["margins"], ids, orc.last["ids"], "config 5 shape (ViT-L/14, 16 traces)")
    if torch.equal(ids, orc.last["ids"]):
        assert out["trace_capts"] == want


def test_config5_full_size_vitl14_depth24_batch128(O):
    """BASELINE config 5 at its full per-GPU size: ViT-L/14-reg at its full depth 24 (1024 wide, 16 heads), fp16 operands,
    batch 128, attention-weighted trace regions, ViECap head (hard + soft prompt, 64 greedy steps).  The CPU oracle cannot
    follow 128 images in a test's time, so: (i) the backbone's tokens of 2 of the 128 images against the fp32 oracle at the
    full depth; (ii) composition: the first 16 images captioned on their own give the captions they get inside the batch of
    128 (every stage is per-image, whatever the launch holds); (iii) the head stage-wise on 16 of the 128 rows: hard-prompt
    token ids exact and greedy ids exact against the oracle on the HIP path's own region features (near-ties explained)."""
    vocab, merges = W.synth_bpe(0)
    tok = V.ByteLevelBPE(vocab, merges)
    w = W.synth_viecap(311, clip_hidden_size=1024, n_layer=12, tok_vocab=len(vocab))
    ents = list(W.SYNTH_ENTITIES)
    emb = W.synth_entity_embeddings(312, len(ents), 1024)
    sd = W.synth_dinov2(93, "dinov2_vitl14_reg")
    m = _model(_viecap_cfg(w, tok, ents, emb, 1024), dino="dinov2_vitl14_reg", depth=None, vit_dtype="fp16", max_batch=64,
               max_prefixes=128, dino_weights=sd)
    B = 128
    imgs = W.synth_images(8, B, 224).cuda()
    traces = [gc.block_trace(i % 13, (5 * i) % 13) for i in range(B)]
    eng = m.engine
    # (i) full-depth backbone vs the oracle
    pick = [0, 77]
    tokens, qkv = eng.vit_forward(imgs)
    vit = O.DinoV2Oracle(sd, num_heads=16)
    d = vit(imgs[pick].cpu())
    ref = torch.cat([d["x_norm_clstoken"][:, None], d["x_norm_regtokens"], d["x_norm_patchtokens"]], 1)
    err = float((tokens[pick].cpu() - ref).abs().max() / ref.abs().max())
    print("ViT-L/14 depth 24, images %s of 128: rel-max-err %.2e" % (pick, err))
    assert err <= 4e-3 and bool(torch.isfinite(tokens).all())
    # (ii) the whole forward at batch 128; composition of everything per-image (backbone, read-out, region means): the first
    # 16 images on their own give bitwise the region features they have inside the batch.  (Captions are NOT composable in
    # the reference either: hard prompts are padded to the batch's longest with pad id 0 and no mask, entrypoint.py:126.)
    out = m(imgs, get_cls_capt=False, traces=traces, use_attention_tracing=True)
    caps = out["trace_capts"]
    ids_all = m.viecap.last_ids.cpu().long().clone()
    prompts_all = m.viecap.last_prompt_tokens.clone()
    assert len(caps) == B and all(isinstance(s, str) for s in caps) and ids_all.shape == (B, 64)
    self_attn, _, _, _ = eng.cls_attention(qkv, tokens)
    feats = eng.region_reduce(tokens, self_attn * eng.trace_grids(traces).view(B, -1), None, 1.0 / m.num_patch_tokens)
    t16, q16 = eng.vit_forward(imgs[:16].clone())
    sa16, _, _, _ = eng.cls_attention(q16, t16)
    f16 = eng.region_reduce(t16, sa16 * eng.trace_grids(traces[:16]).view(16, -1), None, 1.0 / m.num_patch_tokens)
    assert torch.equal(t16, tokens[:16]) and torch.equal(f16, feats[:16])
    # (iii) the head on 16 rows spread over the batch, on the HIP path's own region features
    rows = list(range(0, B, 8))
    sub = feats[rows].clone()
    got = m.viecap.forward(sub.clone())
    ids = m.viecap.last_ids.cpu().long()
    orc = O.ViECapOracle(w, tok, ents, emb, temperature=0.01, top_k=3, threshold=0.4, using_hard_prompt=True, soft_prompt_first=True)
    want = orc.forward(sub.cpu().clone())
    assert np.array_equal(m.viecap.last_prompt_tokens.numpy(), orc.last["prompt_tokens"].numpy())
    _explained(orc.last["margins"], ids, orc.last["ids"], "config 5 full size (ViT-L/14 x 24, 16 of 128 rows)")
    if torch.equal(ids, orc.last["ids"]):
        assert got == want
    # ... and each of those rows carried the same hard prompt (before padding) inside the batch of 128
    for k, r in enumerate(rows):
        a_, b_ = prompts_all[r].tolist(), m.viecap.last_prompt_tokens[k].tolist()
        assert [t for t in a_ if t != 0] == [t for t in b_ if t != 0]      # the two runs pad to different lengths (pad id 0)
