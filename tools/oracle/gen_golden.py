"""Generate ``tests/golden/*.npz`` by running the REFERENCE python modules (through ``refshim``) on the
seeded inputs of ``tests/golden_cases.py``.  Run in the build container only:

    python tools/oracle/gen_golden.py

The committed fixtures hold inputs' seeds/parameters and the reference's OUTPUTS (data), never any
reference source.  The ViT (third-party DINOv2, absent) is not pinned by these; ``vit_hf_crosscheck``
is produced from the independent ``transformers`` port instead (second opinion, see oracle header).
"""
import copy
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import golden_cases as gc  # noqa: E402
import refshim  # noqa: E402
from dino_standin import DinoStandIn  # noqa: E402
from patchioner_amd import weights as W  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)
torch.manual_seed(0)


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def gen_attn(ref):
    c = gc.ATTN
    qkv, patches = gc.attn_inputs()
    T = c["G"] + c["n"] ** 2
    self_attn, maps = ref.dino_extraction.process_self_attention(
        qkv, c["B"], T, c["heads"], c["D"], c["scale"], c["G"], ret_self_attn_maps=True)
    # the three inline lines of the reference forward (P/src/model.py:869-872)
    avg = (self_attn.unsqueeze(-1) * patches).mean(dim=1)
    maps_sm = maps.softmax(dim=-1)
    dis = (patches.unsqueeze(1) * maps_sm.unsqueeze(-1)).mean(dim=2)
    save("attn_readout", self_attn=self_attn, maps=maps, avg_self_attn_token=avg, disentangled=dis)


def gen_traces(ref):
    arrs = {}
    for i, (n, pts) in enumerate(gc.trace_cases()):
        arrs["grid%d" % i] = ref.bbox_utils.map_traces_to_grid(pts, n)
    save("trace_grids", **arrs)


def gen_boxes(ref):
    f = ref.bbox_utils.extract_bboxes_feats
    patches, attn = gc.box_patches(), gc.box_attn()
    arrs = {}

    def run(tag, boxes, **kw):
        b = boxes.clone()
        a = kw.pop("attention_map", None)
        a = a.clone() if a is not None else None
        out = f(patches.clone(), b, patch_size=gc.BOX["patch_size"], attention_map=a, **kw)
        arrs[tag] = out
        arrs[tag + "__boxes_after"] = b         # quirk: caller's tensor is floor-divided in place
        if a is not None:
            arrs[tag + "__attn_after"] = a      # quirk: attention slices renormalised in place

    run("uniform", gc.boxes_regular())
    run("gauss05", gc.boxes_regular(), gaussian_avg=True, gaussian_bbox_variance=0.5)
    run("gauss10", gc.boxes_regular(), gaussian_avg=True, gaussian_bbox_variance=1.0)
    run("attnmap", gc.boxes_regular(), attention_map=attn)
    run("dummy_nan", gc.boxes_with_dummies(), gaussian_avg=True)
    run("single_uniform", gc.boxes_with_dummies(), get_single_embedding_per_image=True)
    run("single_gauss", gc.boxes_with_dummies(), gaussian_avg=True, gaussian_bbox_variance=0.5,
        get_single_embedding_per_image=True)
    run("single_attn", gc.boxes_with_dummies(), attention_map=attn, get_single_embedding_per_image=True)
    run("center_odd", gc.boxes_odd_spans(), gaussian_avg=True, gaussian_bbox_variance=0)
    random.seed(123)
    run("center_even_seed123", gc.boxes_regular(), gaussian_avg=True, gaussian_bbox_variance=0)
    save("bbox_feats", **arrs)


def gen_region_means(ref):
    patches = gc.box_patches()
    arrs = {}
    for v in gc.REGION_VARIANCES:
        arrs["var_%s" % v] = ref.model.compute_region_means(patches.clone(), v)
    random.seed(321)
    arrs["var_0_seed321"] = ref.model.compute_region_means(patches.clone(), 0)
    save("region_means", **arrs)


def make_projector(ref, bank):
    P = ref.im2txt.Im2TxtProjector
    p = object.__new__(P)
    p.device = torch.device("cpu")
    p.embs_dataset = bank
    p.text_dataset = None
    return p


def gen_project(ref):
    arrs = {}
    for tag, clustered in (("gauss", False), ("clustered", True)):
        bank, q = gc.proj_inputs(clustered)
        p = make_projector(ref, bank)
        q1 = q.clone()
        arrs[tag + "_norm"] = p.project(q1, normalize=True)
        arrs[tag + "_q_after"] = q1                      # quirk: query normalised in place
        arrs[tag + "_raw"] = p.project(q.clone(), normalize=False)
        out, sims = p.project(q.clone(), normalize=True, return_n_best_sims=5)
        arrs[tag + "_best5"] = np.asarray(sims, dtype=np.float32)
    save("projection", **arrs)


def build_ref_decap(ref, w, prefix_size=768):
    m = ref.decap.DeCap(prefix_size)
    missing, unexpected = m.load_state_dict(w, strict=False)
    missing = [k for k in missing if not k.endswith(".attn.bias") and not k.endswith("masked_bias")]
    assert not missing and not unexpected, (missing, unexpected)
    return m.eval()


def ref_decode(ref, model, feats, compute_scores=True):
    got = []

    def capture(ids):
        got.append([int(t) for t in ids])
        return ref.decap._Tokenizer.decode(ids)

    ret = ref.decap.decoding_batched(model, feats, compute_scores=compute_scores, decoding_method=capture)
    caps, scores = ret if compute_scores else (ret, None)
    ids = torch.tensor(got, dtype=torch.int64)
    # per-token log-probs / top-2 margins from ONE causal pass of the reference's own module
    emb = torch.cat([model.clip_project(feats).view(feats.shape[0], 1, -1),
                     model.decoder.transformer.wte(ids[:, :-1])], dim=1)
    logits = model.decoder(inputs_embeds=emb).logits
    lp = torch.log_softmax(logits, -1).gather(2, ids.unsqueeze(-1)).squeeze(-1)
    assert torch.equal(logits.argmax(-1), ids), "causal re-run disagrees with the step-wise decode"
    top2 = logits.topk(2, dim=-1).values
    return caps, scores, ids, lp, top2[..., 0] - top2[..., 1]


def gen_decoder(ref):
    w = W.synth_decap(gc.DEC["seed_w"])
    model = build_ref_decap(ref, w)
    arrs, meta = {}, {}
    for kind in ("unit", "raw"):
        x = gc.decoder_prefixes(kind)
        caps, scores, ids, lp, margin = ref_decode(ref, model, x)
        arrs[kind + "_ids"] = ids
        arrs[kind + "_logprob"] = lp
        arrs[kind + "_margin"] = margin
        arrs[kind + "_scores"] = np.asarray(scores, dtype=np.float64)
        meta[kind + "_captions"] = caps
        print("  decoder[%s]: min top-2 margin %.3e  median %.3f" % (kind, margin.min(), margin.median()))
    arrs["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("decoder", **arrs)
    return model


def gen_tokenizer(ref):
    tok = ref.decap._Tokenizer
    rng = np.random.RandomState(9)
    rows = [rng.randint(0, 49408, size=30).tolist() for _ in range(24)]
    rows.append([49406, 320, 1125, 539, 49407, 320, 320])
    rows.append([49407] + [0] * 5)
    strs = [tok.decode(r) for r in rows]
    # the byte table the product's detokeniser is built from (data derived from the BPE vocabulary asset)
    table = [bytes([tok.byte_decoder[c] for c in tok.decoder[i]]) for i in range(len(tok.decoder))]
    offs = np.zeros(len(table) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(t) for t in table])
    blob = np.frombuffer(b"".join(table), dtype=np.uint8)
    asset = os.path.join(ROOT, "patchioner_amd", "assets")
    os.makedirs(asset, exist_ok=True)
    np.savez_compressed(os.path.join(asset, "clip_bpe_decode_table.npz"), offsets=offs, blob=blob)
    save("tokenizer", ids=np.asarray([r + [-1] * (30 - len(r)) for r in rows], dtype=np.int64),
         strings_json=np.frombuffer(json.dumps(strs).encode(), dtype=np.uint8))


def gen_pinv(ref):
    A = gc.randn(71, 768, 512) * 0.05          # synthetic stand-in for the Talk2DINO first Linear
    b = gc.randn(72, 768) * 0.02
    x = gc.randn(73, 6, 768)
    A_pinv = ref.embedding_utils.get_pseudo_inverse(A)
    y = ref.embedding_utils.revert_transformation(x, A_pinv=A_pinv, b=b)
    save("pinv", A_pinv_rows=A_pinv[::32], y=y)


def gen_e2e(ref):
    c = gc.E2E
    D, _, heads = W.dino_arch("dinov2_vitb14_reg")
    vit_w = W.synth_dinov2(c["seed_vit"], depth=c["depth"])
    standin = DinoStandIn(D, c["depth"], heads)
    standin.load_state_dict(vit_w, strict=True)
    dec_w = W.synth_decap(c["seed_dec"])
    tmp = tempfile.NamedTemporaryFile(suffix=".pt", delete=False)
    torch.save(dec_w, tmp.name)
    orig_hub = torch.hub.load
    torch.hub.load = lambda *a, **k: standin
    try:
        model = ref.model.Patchioner.from_config(
            {"decap_weights": tmp.name, "prefix_size": 768, "support_memory_size": 0, "linear_talk2dino": False,
             "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": c["crop"], "crop_dim": c["crop"]},
            device="cpu")
    finally:
        torch.hub.load = orig_hub
        os.unlink(tmp.name)
    model.eval()
    imgs = W.synth_images(c["seed_img"], c["B"], c["crop"])
    bank = W.synth_bank(c["seed_bank"], c["M"])
    calls = []

    def capture(ids):
        calls[-1].append([int(t) for t in ids])
        return ref.decap._Tokenizer.decode(ids)

    model.decoding_method = capture
    arrs, meta = {}, {}

    def run(tag, with_bank, **kw):
        model.im_proj = make_projector(ref, bank) if with_bank else None
        calls.clear()
        orig = ref.model.decoding_batched

        def wrapped(*a, **k):
            calls.append([])
            return orig(*a, **k)

        ref.model.decoding_batched = wrapped
        try:
            outs = model(imgs.clone(), **copy.deepcopy(kw))
        finally:
            ref.model.decoding_batched = orig
        meta[tag] = outs
        for i, ids in enumerate(calls):
            arrs["%s__ids%d" % (tag, i)] = np.asarray(ids, dtype=np.int64)
        # backbone tensors the reference saw (pins the stand-in == oracle ViT equivalence too)
        return outs

    traces, boxes = gc.e2e_traces(), gc.e2e_boxes()
    for with_bank, cfg in ((True, "decap"), (False, "capdec")):
        run(cfg + "_cls_trace", with_bank, get_cls_capt=True, traces=traces)
        run(cfg + "_attn_family", with_bank, get_cls_capt=False, get_avg_self_attn_capt=True,
            get_avg_patch_capt=True, gaussian_img_variance=1, traces=traces, use_attention_tracing=True)
        run(cfg + "_bbox_gauss_scores", with_bank, get_cls_capt=False, bboxes=boxes.clone(), gaussian_avg=True,
            gaussian_bbox_variance=1.0, compute_scores=True, bs_factor=1)
        run(cfg + "_bbox_attnmap", with_bank, get_cls_capt=False, bboxes=boxes.clone(), use_attn_map_for_bboxes=True)
        run(cfg + "_controllable", with_bank, get_cls_capt=False, bboxes=boxes.clone(), get_controllable_capts=True,
            gaussian_avg=True)
    d = standin(imgs, is_training=True)
    arrs["vit_cls"] = d["x_norm_clstoken"]
    arrs["vit_patch_sample"] = d["x_norm_patchtokens"][:, ::37]
    arrs["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("e2e", **arrs)


def _hf_dinov2(w, depth, hidden=768, heads=12):
    """transformers.Dinov2WithRegistersModel (the independent port) carrying OUR state dict ``w``."""
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    cfg = Dinov2WithRegistersConfig(hidden_size=hidden, num_hidden_layers=depth, num_attention_heads=heads, mlp_ratio=4,
                                    image_size=518, patch_size=14, num_register_tokens=4, layerscale_value=1.0,
                                    hidden_act="gelu", qkv_bias=True, use_swiglu_ffn=False, layer_norm_eps=1e-6,
                                    attn_implementation="eager")
    m = Dinov2WithRegistersModel(cfg).eval()
    sd = {}
    sd["embeddings.cls_token"] = w["cls_token"]
    sd["embeddings.mask_token"] = w["mask_token"]
    sd["embeddings.register_tokens"] = w["register_tokens"]
    sd["embeddings.position_embeddings"] = w["pos_embed"]
    sd["embeddings.patch_embeddings.projection.weight"] = w["patch_embed.proj.weight"]
    sd["embeddings.patch_embeddings.projection.bias"] = w["patch_embed.proj.bias"]
    for i in range(depth):
        s, t = "blocks.%d." % i, "encoder.layer.%d." % i
        q, k, v = w[s + "attn.qkv.weight"].chunk(3, 0)
        qb, kb, vb = w[s + "attn.qkv.bias"].chunk(3, 0)
        for nm, ww, bb in (("query", q, qb), ("key", k, kb), ("value", v, vb)):
            sd[t + "attention.attention.%s.weight" % nm] = ww
            sd[t + "attention.attention.%s.bias" % nm] = bb
        sd[t + "attention.output.dense.weight"] = w[s + "attn.proj.weight"]
        sd[t + "attention.output.dense.bias"] = w[s + "attn.proj.bias"]
        sd[t + "norm1.weight"], sd[t + "norm1.bias"] = w[s + "norm1.weight"], w[s + "norm1.bias"]
        sd[t + "norm2.weight"], sd[t + "norm2.bias"] = w[s + "norm2.weight"], w[s + "norm2.bias"]
        sd[t + "layer_scale1.lambda1"] = w[s + "ls1.gamma"]
        sd[t + "layer_scale2.lambda1"] = w[s + "ls2.gamma"]
        sd[t + "mlp.fc1.weight"], sd[t + "mlp.fc1.bias"] = w[s + "mlp.fc1.weight"], w[s + "mlp.fc1.bias"]
        sd[t + "mlp.fc2.weight"], sd[t + "mlp.fc2.bias"] = w[s + "mlp.fc2.weight"], w[s + "mlp.fc2.bias"]
    sd["layernorm.weight"], sd["layernorm.bias"] = w["norm.weight"], w["norm.bias"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m


def _hf_forward(m, imgs):
    try:
        return m(pixel_values=imgs, interpolate_pos_encoding=True).last_hidden_state
    except TypeError:
        return m(pixel_values=imgs).last_hidden_state


def gen_vit_hf():
    """Second opinion on the ViT restatement: the independent HF port, random weights mapped from ours."""
    depth = 2
    m = _hf_dinov2(W.synth_dinov2(81, depth=depth), depth)
    arrs = {}
    for size in (224, 518):
        imgs = W.synth_images(82, 2, size)
        out = _hf_forward(m, imgs)
        arrs["out%d_cls" % size] = out[:, 0]
        arrs["out%d_reg" % size] = out[:, 1:5]
        arrs["out%d_patch_sample" % size] = out[:, 5::53]
    save("vit_hf_crosscheck", **arrs)


def gen_vit_hf12():
    """Full-depth (12 blocks) outputs of the independent HF port for the GPU tests (tests/test_gpu_vit_hf.py): plain
    synthetic weights and the outlier-channel variant (weights.add_outlier_channels), 224^2 and 518^2."""
    depth = 12
    arrs = {}
    for tag, w in (("plain", W.synth_dinov2(83, depth=depth)),
                   ("outlier", W.add_outlier_channels(W.synth_dinov2(83, depth=depth)))):
        m = _hf_dinov2(w, depth)
        for size, stride in ((224, 7), (518, 29)):
            out = _hf_forward(m, W.synth_images(84, 2, size))
            arrs["%s%d_global" % (tag, size)] = out[:, :5]
            arrs["%s%d_patch_sample" % (tag, size)] = out[:, 5::stride]
            arrs["%s%d_absmax" % (tag, size)] = out.abs().max()
    save("vit_hf_depth12", **arrs)


def gen_vit_hf_variants():
    """The HF port at the OTHER backbone sizes the build serves: ViT-L/14-reg at its full depth 24 (BASELINE config 5's
    backbone) and ViT-S/14-reg at depth 12, 224^2 (tests/test_gpu_vit_hf.py holds the HIP ViT to them)."""
    arrs = {}
    for tag, name, hidden, heads, depth in (("vitl", "dinov2_vitl14_reg", 1024, 16, 24), ("vits", "dinov2_vits14_reg", 384, 6, 12)):
        m = _hf_dinov2(W.synth_dinov2(85, name, depth=depth), depth, hidden, heads)
        out = _hf_forward(m, W.synth_images(86, 2, 224))
        arrs["%s_global" % tag] = out[:, :5]
        arrs["%s_patch_sample" % tag] = out[:, 5::7]
        arrs["%s_absmax" % tag] = out.abs().max()
    save("vit_hf_variants", **arrs)


def _hf_clip(w, depth, patch, heads=12, image_size=224):
    """transformers.CLIPVisionModelWithProjection (an independent implementation of the OpenAI vision tower) carrying a
    timm-named state dict ``w`` (weights.synth_clip_vit)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    D, out = w["cls_token"].shape[-1], w["head.weight"].shape[0]
    cfg = CLIPVisionConfig(hidden_size=D, intermediate_size=4 * D, num_hidden_layers=depth, num_attention_heads=heads,
                           image_size=image_size, patch_size=patch, projection_dim=out, hidden_act="quick_gelu",
                           layer_norm_eps=1e-5, attn_implementation="eager")
    m = CLIPVisionModelWithProjection(cfg).eval()
    sd = {"vision_model.embeddings.class_embedding": w["cls_token"].reshape(-1),
          "vision_model.embeddings.position_embedding.weight": w["pos_embed"].reshape(-1, D),
          "vision_model.embeddings.patch_embedding.weight": w["patch_embed.proj.weight"],
          "vision_model.pre_layrnorm.weight": w["norm_pre.weight"], "vision_model.pre_layrnorm.bias": w["norm_pre.bias"],
          "vision_model.post_layernorm.weight": w["norm.weight"], "vision_model.post_layernorm.bias": w["norm.bias"],
          "visual_projection.weight": w["head.weight"]}
    for i in range(depth):
        s, t = "blocks.%d." % i, "vision_model.encoder.layers.%d." % i
        q, k, v = w[s + "attn.qkv.weight"].chunk(3, 0)
        qb, kb, vb = w[s + "attn.qkv.bias"].chunk(3, 0)
        for nm, ww, bb in (("q_proj", q, qb), ("k_proj", k, kb), ("v_proj", v, vb)):
            sd[t + "self_attn.%s.weight" % nm], sd[t + "self_attn.%s.bias" % nm] = ww, bb
        sd[t + "self_attn.out_proj.weight"], sd[t + "self_attn.out_proj.bias"] = w[s + "attn.proj.weight"], w[s + "attn.proj.bias"]
        sd[t + "layer_norm1.weight"], sd[t + "layer_norm1.bias"] = w[s + "norm1.weight"], w[s + "norm1.bias"]
        sd[t + "layer_norm2.weight"], sd[t + "layer_norm2.bias"] = w[s + "norm2.weight"], w[s + "norm2.bias"]
        sd[t + "mlp.fc1.weight"], sd[t + "mlp.fc1.bias"] = w[s + "mlp.fc1.weight"], w[s + "mlp.fc1.bias"]
        sd[t + "mlp.fc2.weight"], sd[t + "mlp.fc2.bias"] = w[s + "mlp.fc2.weight"], w[s + "mlp.fc2.bias"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m


def gen_clip_hf():
    """The CLIP ViT variant (P/src/model.py:358-392, 786-796): every token through HF's own post_layernorm and
    visual_projection (HF applies them to the pooled cls only; the reference applies timm's norm + head to all tokens)."""
    arrs = {}
    for tag, name, patch, stride in (("b16", "vit_base_patch16_clip_224.openai", 16, 5), ("b32", "vit_base_patch32_clip_224.openai", 32, 1)):
        m = _hf_clip(W.synth_clip_vit(87, name, depth=12), 12, patch)
        hs = m.vision_model(pixel_values=W.synth_images(88, 2, 224)).last_hidden_state
        out = m.visual_projection(m.vision_model.post_layernorm(hs))
        arrs["%s_cls" % tag] = out[:, 0]
        arrs["%s_patch_sample" % tag] = out[:, 1::stride]
        arrs["%s_absmax" % tag] = out.abs().max()
    save("clip_vit_hf", **arrs)


def gen_clip_hf_592():
    """configs/decap_B16_resize.k.yaml (resize_dim = crop_dim = 592): P/src/model.py:371 builds the tower with
    ``timm.create_model(dino_model, pretrained=True, img_size=592)``, i.e. timm resamples the checkpoint's 14 x 14 position
    table to 37 x 37 AT LOAD (resample_abs_pos_embed: class position apart, F.interpolate(size=, 'bicubic', antialias=True)).
    The fixture is an HF CLIP vision model built NATIVELY for 592 x 592 whose position table is that resample, computed here
    by the torch call timm makes; the product gets the 14 x 14 table and must do the resampling itself (host C++)."""
    import torch.nn.functional as F
    name = "vit_base_patch16_clip_224.openai"
    w = dict(W.synth_clip_vit(87, name, depth=12))
    pe = w["pos_embed"]
    D = pe.shape[-1]
    grid = F.interpolate(pe[:, 1:].reshape(1, 14, 14, D).permute(0, 3, 1, 2), size=(37, 37), mode="bicubic", antialias=True)
    w592 = dict(w)
    w592["pos_embed"] = torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, 37 * 37, D)], 1)
    m = _hf_clip(w592, 12, 16, image_size=592)
    hs = m.vision_model(pixel_values=W.synth_images(89, 1, 592)).last_hidden_state
    out = m.visual_projection(m.vision_model.post_layernorm(hs))
    save("clip_vit_hf_592", b16_cls=out[:, 0], b16_patch_sample=out[:, 1::9], b16_absmax=out.abs().max(),
         pos_sample=w592["pos_embed"][0, ::17, ::3])


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = refshim.load()
    only = set(sys.argv[1:])

    def want(n):
        return not only or n in only

    if want("attn"): gen_attn(ref)
    if want("traces"): gen_traces(ref)
    if want("boxes"): gen_boxes(ref)
    if want("region"): gen_region_means(ref)
    if want("project"): gen_project(ref)
    if want("decoder"): gen_decoder(ref)
    if want("tokenizer"): gen_tokenizer(ref)
    if want("pinv"): gen_pinv(ref)
    if want("e2e"): gen_e2e(ref)
    if want("vit_hf"): gen_vit_hf()
    if want("vit_hf12"): gen_vit_hf12()
    if want("vit_hf_variants"): gen_vit_hf_variants()
    if want("clip_hf"): gen_clip_hf()
    if want("clip_hf_592"): gen_clip_hf_592()


if __name__ == "__main__":
    main()
