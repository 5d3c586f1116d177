"""Host-side mirror of the reference ``Patchioner`` (P/src/model.py:96-1581) over the HIP engine.

Same constructor keywords, ``from_config`` / ``forward`` / ``caption_tokens`` / ``caption_bboxes`` signatures,
attributes (``image_transforms``, ``image_transforms_no_crop``, ``resize_dim``, ``crop_dim``, ``patch_size``,
``num_tokens``, ``embed_dim`` ...) and output dictionary as the reference, so the ``eval-*-captioning``
drivers run unchanged.  Everything numerical is a C-ABI call into libpatchioner_hip.so; there is no CPU
path (constructing the model without a GPU or without the built library raises).

Scope: the DINOv2(+registers) backbone with the DeCap / CapDec decoder head.  The other
backbones and heads of the reference (ProxyCLIP, RegionCLIP, INViTE, DenseClip, AlphaClip, OpenCLIP, timm
CLIP, DINO.txt, MeaCap, ClipCap) raise ``NotImplementedError`` at construction.

Build-specific config keys (there is no network on the target, so nothing is fetched from torch.hub / HF):
  dino_weights     path (.pt/.pth state dict) or dict of the DINOv2 backbone weights
  memory_bank      path (.npy / .pt / .h5) or tensor [M, D] of the text memory bank
  synthetic_seed   int: synthesise any weights / bank not given (seeded; see weights.py)
  max_batch, max_prefixes, vit_dtype ("fp16" | "bf16")
"""
from __future__ import annotations

import math
import os
import random
from typing import List, Optional

import torch
import torch.nn as nn
import yaml

from . import weights as W
from .engine import Engine
from .preprocess import make_transforms
from .tokenizer import ClipDetokenizer

_OUT_OF_SCOPE = ("proxyclip_clipmodel", "regionclip_config", "invite_config", "denseclip_config",
                 "alphaclip_config", "clipcap_config")


def _load_state_dict(spec):
    if spec is None:
        return None
    if isinstance(spec, dict):
        return spec
    if not os.path.exists(spec):
        raise FileNotFoundError("checkpoint %r not found (no HuggingFace / torch.hub download on this target)" % (spec,))
    sd = torch.load(spec, map_location="cpu")
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


def load_memory_bank(spec, want_texts: bool = False):
    """[M, D] fp32 text-embedding bank (Im2TxtProjector._load_support_memory,
    P/src/decap/im2txtprojection/im2txtprojection.py:387-407: HDF5 datasets '<name>-embeddings' and '<name>-text',
    read by the dependency-free h5lite).  ``want_texts``: returns (bank, list of bytes | None)."""
    bank, texts = _load_memory_bank(spec)
    return (bank, texts) if want_texts else bank


def _load_memory_bank(spec):
    if isinstance(spec, torch.Tensor):
        return spec.float(), None
    if not os.path.exists(spec):
        raise FileNotFoundError("memory bank %r not found" % (spec,))
    if spec.endswith(".npy"):
        import numpy as np
        return torch.from_numpy(np.load(spec)).float(), None
    if spec.endswith(".pt") or spec.endswith(".pth"):
        return torch.load(spec, map_location="cpu").float(), None
    if spec.endswith(".h5") or spec.endswith(".hdf5"):
        from . import h5lite
        names = [k for k in h5lite.dataset_names(spec) if k.endswith("-embeddings")]
        if not names:
            raise KeyError("no '<name>-embeddings' dataset in %r" % spec)
        tname = names[0][:-len("-embeddings")] + "-text"
        d = h5lite.read_datasets(spec, names=(names[0], tname))
        return torch.from_numpy(d[names[0]]).float(), d.get(tname)
    raise ValueError("unsupported memory bank format: %r" % (spec,))


class Patchioner(nn.Module):

    def __init__(self, decoder_weights, device, prefix_size, linear_talk2dino, support_memory_size, projection_type=None,
                 dino_model=None, proxyclip_clipmodel=None, proxyclip_vfm=None, use_talk2dino_project=True, normalize=True,
                 attention_type='qkv', talk2dino_config=None, talk2dino_weights=None, resize_dim=518, crop_dim=518,
                 talk2dino_attn_type='qkv', calculate_argmax_text=False, online_texts=None, clip_model_name=None,
                 use_open_clip=False, viecap_config=None, regionclip_config=None, invite_config=None,
                 denseclip_config=None, alphaclip_config=None, clipcap_config=None, hf_repo_id=None,
                 dino_weights=None, memory_bank=None, synthetic_seed=None, max_batch=16, max_prefixes=128,
                 vit_dtype="fp16", memory_bank_texts=None, **kwargs):
        super().__init__(**kwargs)
        given = dict(proxyclip_clipmodel=proxyclip_clipmodel,
                     regionclip_config=regionclip_config, invite_config=invite_config,
                     denseclip_config=denseclip_config, alphaclip_config=alphaclip_config,
                     clipcap_config=clipcap_config)
        for k in _OUT_OF_SCOPE:
            if given[k] is not None:
                raise NotImplementedError("%s: backbone/head outside the MI355X hot-path scope (DINOv2 + DeCap/CapDec)" % k)
        if use_open_clip or online_texts is not None:
            raise NotImplementedError("use_open_clip / online_texts need the CLIP text tower: outside the hot-path scope")
        # P/src/model.py:339-392: 'dinov2' names load from torch.hub, 'openai' names are the timm OpenAI-CLIP towers of the
        # "DeCap original" configurations (configs/decap_B16*.k.yaml, decap_B32.k.yaml)
        is_clip = dino_model is not None and 'openai' in dino_model and 'dinov2' not in dino_model
        if dino_model is None or 'dinotxt' in dino_model or not ('dinov2' in dino_model or is_clip):
            raise ValueError("Unsupported backbone %r: this build implements the DINOv2 ViT-S/B/L-14 family and the timm "
                             "OpenAI-CLIP ViTs (vit_base_patch16/32_clip_224.openai)" % (dino_model,))
        if is_clip and resize_dim != crop_dim:
            # P/src/model.py:371 builds the tower with img_size=resize_dim: timm's PatchEmbed then asserts that every input is
            # resize_dim x resize_dim, while the transforms crop to crop_dim -- the reference's first forward fails; fail at once
            raise AssertionError("Input height (%d) doesn't match model (%d)." % (crop_dim, resize_dim))
        if viecap_config is not None and viecap_config.get('meacap', False):
            raise NotImplementedError("MeaCap head (retrieved-caption scene graphs, flan-T5): outside the hot-path scope")
        if decoder_weights is None and synthetic_seed is None and not calculate_argmax_text and viecap_config is None:
            raise ValueError("decap_weights is required (or synthetic_seed for seeded synthetic weights)")
        self.decoding_method = None
        self.viecap = None
        self.clipcap = None
        self.calculate_argmax_text = bool(calculate_argmax_text)
        self.text_dataset = memory_bank_texts

        # same validation order as the reference (P/src/model.py:144-162)
        if projection_type in ('coco', 'msmarco', 'blip', 'vg', 'vg-test') or support_memory_size == 0:
            pass
        elif projection_type is not None and os.path.exists(projection_type):
            pass
        elif memory_bank is None and synthetic_seed is None:
            raise Exception("The projection_type field must be 'coco', 'msmarco', 'blip' or 'vg'")

        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("patchioner_amd runs on an MI355X only (device=%r): there is no CPU path; the CPU "
                               "restatement used for parity lives in oracle/ and is test infrastructure" % (device,))
        self._device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())

        self.normalize = normalize
        self.resize_dim = resize_dim
        self.crop_dim = crop_dim
        self.model_name = dino_model
        self.num_global_tokens = 1 if "reg" not in dino_model else 5
        if is_clip:
            # timm VisionTransformer with pre_norm, nn.LayerNorm (eps 1e-5), QuickGELU (model.py:363-371); the reference hands
            # img_size=resize_dim to timm, which resamples the checkpoint's position table for any size but 224 at load
            # (configs/decap_B16_resize.k.yaml: 592): pio_finalize_weights does the same from the table's own grid
            self.embed_dim, depth, heads, patch_size, self.token_dim = W.clip_arch(dino_model)
        else:
            patch_size = 14
            self.embed_dim, depth, heads = W.dino_arch(dino_model)
            self.token_dim = self.embed_dim
        if crop_dim % patch_size != 0:
            raise ValueError("crop_dim must be a multiple of %d (the reference's reshape fails otherwise)" % patch_size)
        self.num_patch_tokens = crop_dim // patch_size * crop_dim // patch_size
        self.num_tokens = self.num_global_tokens + self.num_patch_tokens
        self.num_attn_heads = 16 if 'vits' not in dino_model else 6
        self.scale = 0.125
        self.patch_size = patch_size
        self.backbone_type = 'CLIP' if is_clip else 'DINO'            # model.py:650-651, :786
        if is_clip:
            from .preprocess import CLIP_MEAN, CLIP_STD
            self.image_transforms, self.image_transforms_no_crop = make_transforms(resize_dim, crop_dim, CLIP_MEAN, CLIP_STD)
        else:
            self.image_transforms, self.image_transforms_no_crop = make_transforms(resize_dim, crop_dim)

        vit_sd = _load_state_dict(dino_weights)
        if vit_sd is None:
            if synthetic_seed is None:
                raise FileNotFoundError("dino_weights is required: torch.hub.load('facebookresearch/dinov2', ...) / "
                                        "timm.create_model(..., pretrained=True) need network, which this target does not have")
            vit_sd = W.synth_clip_vit(synthetic_seed + 1, dino_model) if is_clip else W.synth_dinov2(synthetic_seed + 1, dino_model)
        depth = 1 + max(int(k.split(".")[1]) for k in vit_sd if k.startswith("blocks."))
        if attention_type != 'qkv' and not is_clip:
            # the reference re-orders the last block's fused q|k|v rows (P/src/model.py:569-582)
            vit_sd = dict(vit_sd)
            D = self.embed_dim
            wk, bk = "blocks.%d.attn.qkv.weight" % (depth - 1), "blocks.%d.attn.qkv.bias" % (depth - 1)
            ws = dict(zip("qkv", vit_sd[wk].reshape(3, D, D)))
            bs = dict(zip("qkv", vit_sd[bk].reshape(3, D)))
            vit_sd[wk] = torch.cat([ws[x] for x in attention_type], dim=0)
            vit_sd[bk] = torch.cat([bs[x] for x in attention_type], dim=0)

        dec_sd = _load_state_dict(decoder_weights)
        if dec_sd is None and synthetic_seed is not None and viecap_config is None:
            dec_sd = W.synth_decap(synthetic_seed + 2, prefix_size)
        viecap_sd = None
        if viecap_config is not None:
            # P/src/model.py:107-113: the ViECap head replaces the DeCap decode (caption_tokens, :1394-1398).  The engine
            # holds ONE language model: GPT-2-base (12 layers, 12 heads) from the ViECap checkpoint; the DeCap weights the
            # reference also loads are never used on this path and are not loaded here.
            from .viecap import load_viecap_weights
            viecap_sd = load_viecap_weights(viecap_config)
            dec_sd = None

        self.embed_inversion = talk2dino_weights is not None
        inv_sd = {}
        if self.embed_inversion:
            t2d = _load_state_dict(talk2dino_weights)
            A = t2d["linear_layer.weight"].float()       # [dino_dim, clip_dim]
            U, S, Vh = torch.linalg.svd(A, full_matrices=False)      # get_pseudo_inverse, embedding_utils.py:3-15
            S_pinv = torch.zeros_like(S)
            S_pinv[S > 1e-10] = 1.0 / S[S > 1e-10]
            inv_sd["talk2dino.A_pinv"] = (Vh.T @ torch.diag(S_pinv) @ U.T).contiguous()
            inv_sd["talk2dino.b"] = t2d["linear_layer.bias"].float()

        dec_kw = {}
        if viecap_sd is not None:
            n_layer = 1 + max(int(k.split(".")[3]) for k in viecap_sd if k.startswith("gpt.transformer.h."))
            dec_kw = dict(dec_layers=n_layer, dec_heads=12, max_steps=256,
                          dec_vocab=int(viecap_sd["gpt.transformer.wte.weight"].shape[0]),
                          dec_positions=int(viecap_sd["gpt.transformer.wpe.weight"].shape[0]))
        self.engine = Engine(embed_dim=self.embed_dim, depth=depth, num_heads=heads,
                             num_registers=self.num_global_tokens - 1, crop_dim=crop_dim, patch_size=patch_size,
                             pretrain_grid=int(math.isqrt(vit_sd["pos_embed"].shape[1] - 1)), prefix_size=prefix_size,
                             max_batch=max_batch, max_prefixes=max_prefixes, vit_dtype=vit_dtype,
                             device_index=self._device.index, readout_heads=self.num_attn_heads, readout_scale=self.scale,
                             vit_arch="clip" if is_clip else "dinov2", vit_out_dim=self.token_dim if is_clip else 0,
                             vit_ln_eps=1e-5 if is_clip else 1e-6, **dec_kw)
        self.engine.load_state_dict(vit_sd)
        if dec_sd is not None:                         # calculate_argmax_text without decoder weights: no decoder (model.py:165)
            self.engine.load_state_dict(dec_sd)        # strict=False like the reference (decap.py:214)
        if viecap_sd is not None:
            self.engine.load_state_dict({k: v for k, v in viecap_sd.items() if k.startswith(("mapping_network.", "gpt."))})
        if inv_sd:
            self.engine.load_state_dict(inv_sd)
        self.engine.finalize()
        if viecap_config is not None:
            from .viecap import VieCapHead
            self.viecap = VieCapHead(viecap_config, self.engine, clip_model_name)

        if support_memory_size > 0:
            if memory_bank is not None:
                bank, texts = load_memory_bank(memory_bank, want_texts=True)
                if self.text_dataset is None:
                    self.text_dataset = texts
            elif synthetic_seed is not None:
                bank = W.synth_bank(synthetic_seed + 3, support_memory_size, self.token_dim)
            else:
                raise FileNotFoundError("support_memory_size > 0 needs `memory_bank` (path or tensor): building the "
                                        "bank (CLIP text tower + datasets) is offline work outside this scope")
            if 'dinov2' not in dino_model:      # normalize_memory_embs (P/src/model.py:174; im2txtprojection.py:348-349): the CLIP ViTs
                bank = bank / bank.norm(dim=-1, keepdim=True)
            if bank.dim() != 2 or bank.shape[1] != self.token_dim:
                raise ValueError("memory bank is %s, the %s backbone needs [M, %d]" % (tuple(bank.shape), dino_model, self.token_dim))
            self.engine.set_memory_bank(bank)
            self.im_proj = self.engine
        else:
            self.im_proj = None
        if self.calculate_argmax_text:
            if self.im_proj is None or self.text_dataset is None:
                raise ValueError("calculate_argmax_text needs the memory bank AND its caption texts (the '<name>-text' dataset of "
                                 "the bank's .h5, or `memory_bank_texts`)")
            self.text_dataset = [t if isinstance(t, bytes) else str(t).encode("utf-8") for t in self.text_dataset]
        self.tokenizer = ClipDetokenizer()
        self.last_ids = None
        self.call_log = None            # tests: a list collects (decoder prefix, greedy ids) of every caption_tokens call
        self._defer = False
        self.dino = self.engine     # callers test `model.dino is not None`
        # one zero-size parameter so `next(model.parameters()).device` works as in the reference
        self._anchor = nn.Parameter(torch.zeros(0, device=self._device), requires_grad=False)

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config, device='cpu', online_texts=None):
        if type(config) is str:
            if os.path.exists(config):
                with open(config, 'r') as f:
                    config = yaml.safe_load(f)
            else:
                raise FileNotFoundError("config %r is not a local YAML file (HuggingFace download needs network)" % (config,))
        model = cls(
            projection_type=config.get('projection_type', 'coco'),
            decoder_weights=config.get('decap_weights', None),
            device=device,
            prefix_size=config['prefix_size'],
            linear_talk2dino=config.get('linear_talk2dino', False),
            support_memory_size=config['support_memory_size'],
            dino_model=config.get('dino_model', None),
            proxyclip_clipmodel=config.get('proxyclip_clipmodel', None),
            proxyclip_vfm=config.get('proxyclip_vfm', None),
            use_talk2dino_project=config.get('use_talk2dino_project', True),
            normalize=config.get('normalize', True),
            attention_type=config.get('attention_type', 'qkv'),
            talk2dino_config=config.get('talk2dino_config', None),
            talk2dino_weights=config.get('talk2dino_weights', None),
            resize_dim=config.get('resize_dim', 518),
            crop_dim=config.get('crop_dim', 518),
            talk2dino_attn_type=config.get('talk2dino_attn_type', 'qkv'),
            calculate_argmax_text=config.get('calculate_argmax_text', False),
            clip_model_name=config.get('clip_model_name', None),
            online_texts=online_texts,
            use_open_clip=config.get('use_open_clip', False),
            viecap_config=config.get('viecap', None),
            regionclip_config=config.get('regionclip_config', None),
            invite_config=config.get('invite_config', None),
            denseclip_config=config.get('denseclip_config', None),
            alphaclip_config=config.get('alphaclip_config', None),
            clipcap_config=config.get('clipcap', None),
            hf_repo_id=config.get('hf_repo_id', None),
            dino_weights=config.get('dino_weights', None),
            memory_bank=config.get('memory_bank', None),
            memory_bank_texts=config.get('memory_bank_texts', None),
            synthetic_seed=config.get('synthetic_seed', None),
            max_batch=config.get('max_batch', 16),
            max_prefixes=config.get('max_prefixes', 128),       # what ONE greedy decode serves (the engine's limit)
            vit_dtype=config.get('vit_dtype', 'fp16'),
        )
        model.to(device)
        return model

    def to(self, *args, **kwargs):
        """The weights live in the HIP library on the construction device: ``model.to(<that device>)`` (what the eval drivers
        do, eval_trace_captioning.py:185) returns self; any other device or a dtype change raises instead of silently
        leaving the model where it is."""
        device = kwargs.get("device")
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, (str, torch.device, int)):
                device = a
            elif isinstance(a, torch.Tensor):
                device, dtype = a.device, a.dtype
        if dtype is not None and dtype != torch.float32:
            raise RuntimeError("patchioner_amd: the model's precision is fixed when it is built (vit_dtype); .to(%s) is not supported" % dtype)
        if device is not None:
            d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
            idx = d.index if d.index is not None else (torch.cuda.current_device() if d.type == "cuda" else None)
            if d.type != "cuda" or idx != self._device.index:
                raise RuntimeError("patchioner_amd: the model was built on %s and cannot move to %s; build it there with "
                                   "Patchioner.from_config(config, device=...)" % (self._device, d))
        return self

    # ------------------------------------------------------------------------------------------
    def forward(self, imgs,
                get_cls_capt=True,
                get_avg_self_attn_capt=False,
                get_attn_heads_capt=False,
                get_patch_capts=False,
                get_register_capts=False,
                bboxes=None,
                traces=None,
                get_controllable_capts=False,
                bs_factor=4,
                gaussian_avg=False,
                gaussian_bbox_variance=0.5,
                get_avg_patch_capt=False,
                gaussian_img_variance=1,
                use_attn_map_for_bboxes=False,
                use_attention_tracing=False,
                double_DINO_for_bboxes=False,
                double_DINO_for_bboxes_return_type="avg",
                double_DINO_use_cls=False,
                cleaning_type=None,
                clean_after_projection=True,
                alpha=1.0,
                clean_from="cls",
                caption_bboxes_type: str = None,
                return_n_best_sims=None,
                compute_scores: bool = False
                ):
        assert clean_from in ["cls", "avg_self_attn"]
        assert cleaning_type in [None, "orthogonal_projection", "contrastive_mask"]
        if double_DINO_for_bboxes and double_DINO_for_bboxes_return_type not in ("cls", "avg", "gaussian_avg"):
            raise ValueError("double_DINO_for_bboxes_return_type must be 'cls', 'avg' or 'gaussian_avg'")
        if cleaning_type is not None and self.im_proj is None:
            # the reference calls self.im_proj.project unconditionally here (model.py:900-913)
            raise AttributeError("cleaning_type needs the memory-bank projector (support_memory_size > 0)")
        if return_n_best_sims is not None and not self.calculate_argmax_text:
            # caption_tokens returns the similarities only on the calculate_argmax_text path (model.py:1408-1411); without
            # it the reference fails while unpacking (model.py:1030, 1036)
            raise ValueError("return_n_best_sims needs a model built with calculate_argmax_text")
        if caption_bboxes_type is not None:
            return self.caption_bboxes(imgs, bboxes, caption_bboxes_type, compute_scores=compute_scores)

        eng = self.engine
        outs = {}
        bs = imgs.shape[0]
        # has_attention (model.py:864-865): only DINO backbones install the qkv hook.  Without it the reference never defines
        # self_attn / avg_self_attn_token / disentangled_self_attn and every option that reads them fails with
        # UnboundLocalError; clean_from == "avg_self_attn" falls back to the cls token (:886-890).
        has_attention = 'DINO' in self.backbone_type
        if double_DINO_for_bboxes and not has_attention:
            raise AttributeError("double_DINO_for_bboxes re-runs a DINOv2 block (bbox_utils.py:300-403): not a %s backbone" % self.backbone_type)
        tokens, qkv = eng.vit_forward(imgs, want_qkv=has_attention)
        G = self.num_global_tokens
        embed_dim = tokens.shape[-1]                 # model.py:924: the width of x_norm_patchtokens (512 behind the CLIP head)
        clean_avg = cleaning_type is not None and clean_from == "avg_self_attn" and has_attention
        if has_attention:
            self_attn, _, avg_self_attn_token, disentangled_self_attn = eng.cls_attention(
                qkv, tokens, want_maps=False, want_avg=get_avg_self_attn_capt or clean_avg, want_disentangled=get_attn_heads_capt)
        else:
            self_attn = avg_self_attn_token = disentangled_self_attn = None
            for flag, name in ((get_avg_self_attn_capt, "avg_self_attn_token"), (get_attn_heads_capt, "disentangled_self_attn"),
                               (use_attn_map_for_bboxes and bboxes is not None, "self_attn"),
                               (use_attention_tracing and traces is not None, "self_attn")):
                if flag:
                    raise UnboundLocalError("local variable %r referenced before assignment (the %s backbone has no "
                                            "attention hook, P/src/model.py:864-872)" % (name, self.backbone_type))
        if cleaning_type is not None:
            # P/src/model.py:879-922: every patch token goes through the memory-bank projection and ctx_cleaner (before
            # or after it); the cleaned, projected tokens REPLACE x_norm_patchtokens, and the patch / box captions
            # below skip their own projection.  (The reference's in-place normalisation of the tokens it hands to
            # project() is invisible afterwards: every later consumer projects, i.e. re-normalises, or uses the
            # replaced patch tokens.)
            n2 = self.num_patch_tokens
            patches = tokens[:, G:].contiguous()
            cf = avg_self_attn_token if clean_avg else tokens[:, 0].contiguous()
            if clean_after_projection:
                proj = eng.project_many(patches.view(-1, embed_dim), normalize=True).view(bs, n2, embed_dim)
                cleaned = eng.ctx_clean(proj, eng.project_many(cf.clone(), normalize=True), cleaning_type, alpha)
            else:
                cleaned = eng.ctx_clean(patches, cf, cleaning_type, alpha, normalize_inputs=True)
                cleaned = eng.project_many(cleaned.view(-1, embed_dim), normalize=True).view(bs, n2, embed_dim)
            tokens = torch.cat([tokens[:, :G], cleaned], dim=1)
        project_regions = cleaning_type is None

        def put(key, score_key, ret):
            if compute_scores is True:
                outs[key], outs[score_key] = ret
            else:
                outs[key] = ret

        if get_cls_capt:
            put('cls_capt', 'cls_capt_scores', self.caption_tokens(tokens[:, 0].contiguous(), compute_scores=compute_scores))
        if get_avg_self_attn_capt:
            put('avg_self_attn_capt', 'avg_self_attn_capt_scores',
                self.caption_tokens(avg_self_attn_token, compute_scores=compute_scores))
        if get_avg_patch_capt:
            put('avg_patch_capt', 'avg_patch_capt_scores',
                self.caption_tokens(self._region_means(tokens, gaussian_img_variance), compute_scores=compute_scores))
        if get_attn_heads_capt:
            ret = self.caption_tokens(disentangled_self_attn.view(-1, embed_dim), compute_scores=compute_scores)
            H = self.num_attn_heads
            caps = ret[0] if compute_scores is True else ret
            outs['attn_heads_capts'] = [caps[i * H:(i + 1) * H] for i in range(bs)]
            if compute_scores is True:
                outs['attn_heads_scores'] = [ret[1][i * H:(i + 1) * H] for i in range(bs)]
        if get_patch_capts:
            n_patches = self.num_patch_tokens
            ret = self.caption_tokens(tokens[:, G:].reshape(-1, embed_dim), project=project_regions, compute_scores=compute_scores)
            caps = ret[0] if compute_scores is True else ret
            outs['patch_tokens_capts'] = [caps[i * n_patches:(i + 1) * n_patches] for i in range(bs)]
            if compute_scores is True:
                outs['patch_tokens_scores'] = [ret[1][i * n_patches:(i + 1) * n_patches] for i in range(bs)]
        if get_register_capts:
            ret = self.caption_tokens(tokens[:, 1:G].reshape(-1, embed_dim), compute_scores=compute_scores)
            caps = ret[0] if compute_scores is True else ret
            outs['register_capts'] = [caps[i * 4:(i + 1) * 4] for i in range(bs)]
            if compute_scores is True:
                outs['register_scores'] = [ret[1][i * 4:(i + 1) * 4] for i in range(bs)]

        if bboxes is not None and not get_controllable_capts:
            bbox_bs = bs * bs_factor
            n_boxes = bboxes.shape[1]
            if double_DINO_for_bboxes:
                bbox_feats = self._bbox_feats_double_dino(tokens, bboxes, double_DINO_use_cls, double_DINO_for_bboxes_return_type,
                                                          gaussian_bbox_variance).view(-1, embed_dim)
            else:
                bbox_feats = self._bbox_feats(tokens, bboxes, gaussian_avg, gaussian_bbox_variance, False,
                                              self_attn if use_attn_map_for_bboxes else None).view(-1, embed_dim)
            # The reference captions the boxes in chunks of bbox_bs = bs * bs_factor (model.py:1000-1033), a memory
            # measure of its cache-less decode.  Projection and decode are row-independent, so ONE call gives the
            # same captions / scores; the engine splits at its own capacities (16 queries per bank pass, max_prefixes
            # per decode), e.g. 128 boxes decode as 2 x 64 instead of 4 x 32 prefixes.
            del bbox_bs
            ret = self.caption_tokens(bbox_feats, project=project_regions, return_n_best_sims=return_n_best_sims,
                                      compute_scores=compute_scores)
            if compute_scores is True:
                ret, scores = ret
                outs['bbox_scores'] = list(scores)
            if return_n_best_sims is not None:          # model.py:1023-1040: (captions, similarities) per chunk
                ret, sims = ret
                outs['bbox_sims'] = [list(sims)[i * n_boxes:(i + 1) * n_boxes] for i in range(bs)]
            outs['bbox_capts'] = [list(ret)[i * n_boxes:(i + 1) * n_boxes] for i in range(bs)]
            if compute_scores is True:
                outs['bbox_scores'] = [outs['bbox_scores'][i * n_boxes:(i + 1) * n_boxes] for i in range(bs)]
        elif bboxes is not None and get_controllable_capts:
            bbox_feats = self._bbox_feats(tokens, bboxes, gaussian_avg, gaussian_bbox_variance, True,
                                          self_attn if use_attn_map_for_bboxes else None)
            outs['set_controllable_capts'] = self.caption_tokens(bbox_feats)

        if traces is not None:
            # map_traces_to_grid + (grid * patches).mean((1,2))  (P/src/model.py:1049-1054): mean over n*n cells
            grids = eng.trace_grids(traces).view(bs, -1)
            if use_attention_tracing:
                grids = self_attn * grids
            trace_embeds = eng.region_reduce(tokens, grids, None, 1.0 / self.num_patch_tokens)
            outs['trace_capts'] = self.caption_tokens(trace_embeds)
        return outs

    # ------------------------------------------------------------------------------------------
    def _region_means(self, tokens, variance):
        """compute_region_means (P/src/model.py:45-94)."""
        bs, eng = tokens.shape[0], self.engine
        n = eng.n
        if variance == 0:
            opts = [n // 2] if n % 2 == 1 else [n // 2 - 1, n // 2]
            wmap = torch.zeros(bs, n, n)
            for i in range(bs):
                cy = random.choice(opts)
                cx = random.choice(opts)
                wmap[i, cy, cx] = 1.0
            wmap = wmap.view(bs, -1).to(eng.device)
        else:
            wmap = eng.gaussian_map(variance).unsqueeze(0).expand(bs, -1).contiguous()
        return eng.region_reduce(tokens, wmap, None, 1.0)

    def _center_choices(self, boxes_i, single):
        """host draw of the var==0 centre cell, same RNG call order as bbox_utils.py:62-71"""
        n = self.engine.n
        B, NB = boxes_i.shape[:2]
        out = torch.zeros(B, NB, 2, dtype=torch.int32)

        def span(a, length):
            lo, hi = slice(a, a + length + 1).indices(n)[:2]
            return max(0, hi - lo)

        bl = boxes_i.tolist()
        for i in range(B):
            for j in range(NB):
                x1, y1, w, h = bl[i][j]
                if single and x1 + y1 + w + h < 0:
                    continue
                hs, ws = span(y1, h), span(x1, w)
                cy = [hs // 2] if hs % 2 == 1 else [hs // 2 - 1, hs // 2]
                cx = [ws // 2] if ws % 2 == 1 else [ws // 2 - 1, ws // 2]
                out[i, j, 0] = random.choice(cy)
                out[i, j, 1] = random.choice(cx)
        return out

    def _bbox_feats(self, tokens, bboxes, gaussian_avg, variance, single, attention_map):
        """extract_bboxes_feats (P/src/bbox_utils.py:8-109); `bboxes //= patch_size` mutates the caller's tensor."""
        eng = self.engine
        bboxes //= self.patch_size
        boxes_i = bboxes.int().cpu()
        B, NB = boxes_i.shape[:2]
        cc = None
        attn = None
        if attention_map is not None:
            mode = 3
            attn = attention_map.clone()        # the reference hands a .cpu() copy, so the original map survives
        elif gaussian_avg:
            if variance == 0:
                mode, cc = 2, self._center_choices(boxes_i, single)
            else:
                mode = 1
        else:
            mode = 0
        weights, single_map = eng.bbox_weights(boxes_i, mode, variance, cc, attn, single_map=single)
        if single:
            return eng.region_reduce(tokens, single_map, None, 1.0)
        idx = torch.arange(B, dtype=torch.int32).repeat_interleave(NB)
        return eng.region_reduce(tokens, weights, idx, 1.0).view(B, NB, self.token_dim)

    def _bbox_feats_double_dino(self, tokens, bboxes, use_cls, return_type, variance):
        """extract_bboxes_feats_double_dino (P/src/bbox_utils.py:300-403).  The reference floor-divides a CLONE of the
        xywh tensor by the patch size and reads it as (x1, y1, x2, y2) with inclusive python slices; the last block runs
        on [cls | registers | region patches] ("cls" / "avg": pio_bbox_double_dino); "gaussian_avg" weights the block's
        INPUT patches with a gaussian normalised to sum 1 and never needs the block."""
        if return_type == "cls" and not use_cls:
            raise AssertionError("return_type 'cls' needs double_DINO_use_cls")       # reference: assert return_type != "cls"
        idx = bboxes.clone()
        idx //= self.patch_size
        idx = idx.int().cpu()
        B, NB = idx.shape[:2]
        n = self.engine.n
        slices = torch.zeros(B, NB, 4, dtype=torch.int32)
        for i in range(B):
            for j in range(NB):
                x1, y1, x2, y2 = (int(v) for v in idx[i, j])
                ys, ye, _ = slice(y1, y2 + 1).indices(n)
                xs, xe, _ = slice(x1, x2 + 1).indices(n)
                slices[i, j] = torch.tensor([ys, max(ye, ys), xs, max(xe, xs)])
        if return_type != "gaussian_avg":
            return self.engine.bbox_double_dino(tokens, slices, use_cls, return_type)
        weights = torch.zeros(B * NB, n, n)
        for r, (ys, ye, xs, xe) in enumerate(slices.view(-1, 4).tolist()):
            h_span, w_span = ye - ys, xe - xs
            if h_span == 0 or w_span == 0:
                continue                                                              # empty region: sum of nothing = zeros
            yc, xc = torch.meshgrid(torch.linspace(-1, 1, h_span), torch.linspace(-1, 1, w_span), indexing="ij")
            w = torch.exp(-(xc ** 2 + yc ** 2) / variance)
            weights[r, ys:ye, xs:xe] = w / w.sum()
        img = torch.arange(B, dtype=torch.int32).repeat_interleave(NB)
        return self.engine.region_reduce(tokens, weights.to(tokens.device), img, 1.0).view(B, NB, self.embed_dim)

    # ------------------------------------------------------------------------------------------
    def preprocess_images(self, images, no_crop: bool = False):
        """Device-side ``torch.stack([self.image_transforms(im) for im in images]).to(device)`` (``no_crop``:
        ``image_transforms_no_crop``): same floats, computed on the GPU from the raw pixels (Engine.preprocess)."""
        return self.engine.preprocess(images, self.resize_dim, self.crop_dim, no_crop=no_crop)

    def caption_bboxes(self, imgs, bboxes, capt_type='cls_capt', crop_boxes=False, compute_scores=False):
        """P/src/model.py:1356-1390: crop each box from the PIL image, transform the crop (``image_transforms`` when
        ``crop_boxes`` else ``image_transforms_no_crop``) and caption it with a whole forward pass, ``bs`` crops at a time.
        process_bboxes (P/src/bbox_utils.py:406-421) runs the resampling on the host; here only PIL's crop (a copy of the
        pixels, its float box rounded by PIL as in the reference) stays there and resize / centre-crop / normalisation run
        on the GPU (pio_preprocess: the same floats as the PIL transform, tests/test_gpu_preprocess.py)."""
        bs = len(imgs)
        n_bboxes = bboxes.shape[1]
        regions = []
        for img, img_boxes in zip(imgs, bboxes.tolist()):
            for x_min, y_min, w, h in img_boxes:
                regions.append(img.crop((x_min, y_min, x_min + w, y_min + h)))
        crops = self.preprocess_images(regions, no_crop=not crop_boxes)
        capts, scores = [], []
        for i in range(n_bboxes):
            start = i * bs
            end = start + bs if i < n_bboxes - 1 else crops.shape[0]
            out = self.forward(crops[start:end], get_cls_capt=capt_type == 'cls_capt',
                               get_avg_self_attn_capt=capt_type == 'avg_self_attn_capt')
            capts += out[capt_type]
            if compute_scores:      # as in the reference: forward() is not asked for scores, so this key is missing (KeyError)
                scores += out[f"{capt_type}_scores"]
        ret = {'bbox_capts': [capts[i * n_bboxes:(i + 1) * n_bboxes] for i in range(bs)]}
        if compute_scores:
            ret['bbox_scores'] = [scores[i * n_bboxes:(i + 1) * n_bboxes] for i in range(bs)]
        return ret

    def caption_tokens(self, dino_tokens, project=True, return_n_best_sims=None, compute_scores: bool = False):
        """P/src/model.py:1392-1423."""
        eng = self.engine
        if self.viecap is not None:
            if return_n_best_sims:
                raise Exception("return_n_best_sims is not supported with viecap")
            out = self.viecap.forward(dino_tokens, compute_scores=compute_scores)
            self.last_ids = self.viecap.last_ids      # [N, 64] ids of the greedy search: what dist.sharded_* gathers
            return out
        if self.im_proj is None:
            project = False
        x = dino_tokens
        if not isinstance(x, torch.Tensor):
            x = torch.tensor(x, dtype=torch.float)
        xd = x.to(device=eng.device, dtype=torch.float32).contiguous()
        if self.calculate_argmax_text:
            self.last_ids = None                       # no decoder on this path: nothing for dist.sharded_* to gather
            # model.py:1408-1411 -> im2txtprojection.py:367-375: the text of the most similar bank row, no decoder.  The row
            # index counts the rows kept at load while the texts are the file's unfiltered list (the reference's indexing).
            sims, rows = eng.topk_rows(xd, k=int(return_n_best_sims) if return_n_best_sims else 1)
            if isinstance(dino_tokens, torch.Tensor) and xd.data_ptr() != dino_tokens.data_ptr():
                dino_tokens.copy_(xd)           # normalised in place (im2txtprojection.py:368)
            captions = [self.text_dataset[int(r)].decode() for r in rows[:, 0].cpu().tolist()]
            if return_n_best_sims:
                captions = (captions, sims.cpu().tolist())
            return captions if compute_scores is False else (captions, [1.0] * len(captions))   # the reference's len(): 2 for a (texts, sims) pair
        if project:
            prefix = eng.project(xd, normalize=self.normalize)
            if isinstance(dino_tokens, torch.Tensor) and xd.data_ptr() != dino_tokens.data_ptr():
                dino_tokens.copy_(xd)           # quirk: the query is L2-normalised in place (im2txtprojection.py:368)
        else:
            prefix = xd
        if self.embed_inversion:
            prefix = eng.revert_transformation(prefix)
        ids, lp = eng.decode_greedy(prefix, steps=30, want_logprob=compute_scores)
        self.last_ids = ids
        if self.call_log is not None:
            self.call_log.append((prefix.detach().clone(), ids.detach().clone()))
        if self._defer:
            if compute_scores:
                raise NotImplementedError("forward_async does not defer score lists; call forward() for compute_scores")
            return _PendingCaptions(ids, self.tokenizer, self.decoding_method)
        outputs = self.tokenizer.batch_captions(ids.cpu().tolist(), decoding_method=self.decoding_method)
        if compute_scores:
            return outputs, torch.exp(lp.sum(dim=-1)).cpu().numpy().tolist()
        return outputs

    # ------------------------------------------------------------------------------------------
    def forward_async(self, imgs, stream: Optional["torch.cuda.Stream"] = None, **kwargs) -> "PendingForward":
        """Enqueue a whole forward on `stream` without waiting for the GPU: same arguments as ``forward`` (the
        flat caption outputs only: cls / avg_self_attn / avg_patch / trace / set_controllable captions), returns a
        handle whose ``result()`` gives exactly what ``forward`` would have returned.  Several model instances,
        each on its own stream, can keep several batches in flight (the decode of one batch is a chain of small
        latency-bound kernels that leaves most CUs idle, so it overlaps with the next batch's ViT)."""
        for k in ("bboxes", "get_attn_heads_capt", "get_patch_capts", "get_register_capts"):
            v = kwargs.get(k)
            if v is not None and v is not False:
                if not (k == "bboxes" and kwargs.get("get_controllable_capts")):
                    raise NotImplementedError("forward_async supports the flat caption outputs only (%s given)" % k)
        stream = stream or torch.cuda.current_stream()
        self._defer = True
        try:
            with torch.cuda.stream(stream):
                outs = self.forward(imgs, **kwargs)
                for v in outs.values():
                    if isinstance(v, _PendingCaptions):
                        v.start_copy()
                ev = torch.cuda.Event()
                ev.record(stream)
        finally:
            self._defer = False
        return PendingForward(outs, ev)

    def __len__(self):
        return self.engine.num_weights


class _PendingCaptions:
    """Greedy ids still on the device; resolved to the reference's list-of-strings on demand."""

    def __init__(self, ids_dev, tokenizer, decoding_method):
        self.ids_dev, self.tokenizer, self.decoding_method = ids_dev, tokenizer, decoding_method
        self.ids_host = None

    def start_copy(self):
        self.ids_host = torch.empty(self.ids_dev.shape, dtype=self.ids_dev.dtype).pin_memory()
        self.ids_host.copy_(self.ids_dev, non_blocking=True)

    def resolve(self):
        return self.tokenizer.batch_captions(self.ids_host.tolist(), decoding_method=self.decoding_method)


class PendingForward:
    def __init__(self, outs, event):
        self._outs, self._event = outs, event

    def done(self) -> bool:
        return self._event.query()

    def ids(self, key: str):
        """device tensor of the greedy ids behind output `key` (valid once result() or done())"""
        return self._outs[key].ids_dev

    def result(self) -> dict:
        self._event.synchronize()
        return {k: (v.resolve() if isinstance(v, _PendingCaptions) else v) for k, v in self._outs.items()}
