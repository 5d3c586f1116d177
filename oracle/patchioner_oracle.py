"""CPU oracle for the Patch-ioner hot path (TEST INFRASTRUCTURE -- never the product path).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It is a plain PyTorch-CPU fp32 restatement of the reference's algorithm, kept
deliberately *as the reference executes it* (no KV cache, logits for every position, the memory
bank re-normalised and streamed three times per call, python loops over boxes / trace points), so
that timing it stands in for "the reference CPU path" (BASELINE.md section 3).

Pinning status
--------------
* a4-a15 (attention read-out, region weighting, projection, decoder, routing): PINNED against the
  reference's own python modules executed in the build container through
  ``tools/oracle/refshim.py``; the resulting vectors are committed under ``tests/golden/`` together
  with the generator ``tools/oracle/gen_golden.py``.
* a2 (the DINOv2 ViT): the arithmetic lives in the third-party ``facebookresearch/dinov2``
  torch.hub package (unpinned, not vendored under /root/reference) -> "parity unpinned" against
  the reference itself.  ``DinoV2Oracle`` restates the published architecture and is cross-checked
  against the independent ``transformers.Dinov2WithRegistersModel`` port (fixture
  ``tests/golden/vit_hf_crosscheck.npz``).

Every function cites the reference file:line it follows (P/ = /root/reference/Patch-ioner/).
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# a2  DINOv2 ViT (third-party; public architecture restated)
# --------------------------------------------------------------------------------------------


class DinoV2Oracle:
    """``DinoVisionTransformer.forward(imgs, is_training=True)`` as called at P/src/model.py:783.

    ``w`` is a state dict in the hub model's own naming (``cls_token``, ``pos_embed``,
    ``register_tokens``, ``patch_embed.proj.*``, ``blocks.{i}.{norm1,attn.qkv,attn.proj,ls1,norm2,
    mlp.fc1,mlp.fc2,ls2}.*``, ``norm.*``).  Position interpolation follows the hub entry points: ``_reg`` models
    interpolate_antialias=True, interpolate_offset=0.0; the models without registers antialias=False, offset=0.1
    (``scale_factor = (n + 0.1) / M``, the kludge of facebookresearch/dino issue 8).
    """

    def __init__(self, w: Dict[str, torch.Tensor], num_heads: int, patch_size: int = 14,
                 num_register_tokens: int = 4, eps: float = 1e-6):
        self.w = {k: v.detach().float().cpu() for k, v in w.items()}
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.R = num_register_tokens
        self.eps = eps
        self.depth = 1 + max(int(k.split(".")[1]) for k in self.w if k.startswith("blocks."))
        self.D = self.w["cls_token"].shape[-1]
        self.last_qkv: Optional[torch.Tensor] = None  # what the reference's forward hook captures

    def pos_embed_for(self, n_h: int, n_w: int) -> torch.Tensor:
        pe = self.w["pos_embed"]
        N = pe.shape[1] - 1
        M = int(math.sqrt(N))
        assert M * M == N
        if n_h * n_w == N and n_h == n_w:
            return pe
        cls_pe, patch_pe = pe[:, :1], pe[:, 1:]
        grid = patch_pe.reshape(1, M, M, self.D).permute(0, 3, 1, 2)
        if self.R > 0:
            patch_pe = F.interpolate(grid, size=(n_h, n_w), mode="bicubic", antialias=True)
        else:
            patch_pe = F.interpolate(grid, scale_factor=(float(n_h + 0.1) / M, float(n_w + 0.1) / M), mode="bicubic",
                                     antialias=False)
            assert patch_pe.shape[-2:] == (n_h, n_w)
        patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, n_h * n_w, self.D)
        return torch.cat([cls_pe, patch_pe], dim=1)

    def tokens(self, imgs: torch.Tensor) -> torch.Tensor:
        B, _, H, W = imgs.shape
        p = self.patch_size
        x = F.conv2d(imgs.float(), self.w["patch_embed.proj.weight"], self.w["patch_embed.proj.bias"],
                     stride=p)
        n_h, n_w = x.shape[-2:]
        x = x.flatten(2).transpose(1, 2)
        x = torch.cat([self.w["cls_token"].expand(B, -1, -1), x], dim=1)
        x = x + self.pos_embed_for(n_h, n_w)
        if self.R:
            x = torch.cat([x[:, :1], self.w["register_tokens"].expand(B, -1, -1), x[:, 1:]], dim=1)
        return x

    def block(self, i: int, x: torch.Tensor, capture_qkv: bool = False) -> torch.Tensor:
        w, D, h = self.w, self.D, self.num_heads
        pre = "blocks.%d." % i
        B, T, _ = x.shape
        y = F.layer_norm(x, (D,), w[pre + "norm1.weight"], w[pre + "norm1.bias"], self.eps)
        qkv = F.linear(y, w[pre + "attn.qkv.weight"], w[pre + "attn.qkv.bias"])
        if capture_qkv:
            self.last_qkv = qkv
        q, k, v = qkv.reshape(B, T, 3, h, D // h).permute(2, 0, 3, 1, 4)
        attn = (q * (D // h) ** -0.5) @ k.transpose(-2, -1)
        attn = attn.softmax(dim=-1)
        y = (attn @ v).transpose(1, 2).reshape(B, T, D)
        y = F.linear(y, w[pre + "attn.proj.weight"], w[pre + "attn.proj.bias"])
        x = x + y * w[pre + "ls1.gamma"]
        y = F.layer_norm(x, (D,), w[pre + "norm2.weight"], w[pre + "norm2.bias"], self.eps)
        y = F.linear(y, w[pre + "mlp.fc1.weight"], w[pre + "mlp.fc1.bias"])
        y = F.gelu(y)  # exact erf GELU
        y = F.linear(y, w[pre + "mlp.fc2.weight"], w[pre + "mlp.fc2.bias"])
        return x + y * w[pre + "ls2.gamma"]

    def __call__(self, imgs: torch.Tensor) -> Dict[str, torch.Tensor]:
        x = self.tokens(imgs)
        for i in range(self.depth):
            x = self.block(i, x, capture_qkv=(i == self.depth - 1))
        xn = F.layer_norm(x, (self.D,), self.w["norm.weight"], self.w["norm.bias"], self.eps)
        R = self.R
        return {"x_norm_clstoken": xn[:, 0], "x_norm_regtokens": xn[:, 1:R + 1],
                "x_norm_patchtokens": xn[:, R + 1:], "x_prenorm": x}


class ClipViTOracle:
    """The OpenAI-CLIP vision tower as the reference runs it (P/src/model.py:358-392, 786-796):
    ``timm.create_model('vit_base_patch16_clip_224.openai', act_layer=QuickGELU)`` then
    ``output = dino.forward_features(imgs); output = dino.head(output)`` on EVERY token.

    timm is a third-party dependency that is absent here (R/requirements.txt lists ``timm`` unpinned); this restates the
    public ``timm.models.vision_transformer.VisionTransformer`` forward for the CLIP configuration (pre_norm=True,
    nn.LayerNorm eps 1e-5, class token with a position of its own, no LayerScale, qkv bias, bias-free patch conv,
    ``fc_norm`` absent) with the reference's QuickGELU (model.py:363-365).  Second opinion, like DinoV2Oracle's:
    ``transformers.CLIPVisionModelWithProjection`` (tests/golden/clip_vit_hf.npz, tools/oracle/gen_golden.py: gen_clip_hf).
    State-dict keys are timm's: cls_token, pos_embed, patch_embed.proj.weight, norm_pre.*, blocks.{i}.{norm1, attn.qkv,
    attn.proj, norm2, mlp.fc1, mlp.fc2}.*, norm.*, head.weight (+ head.bias, zeros in timm's converted checkpoint)."""

    has_attention = False          # model.py:864-865: no qkv hook on this backbone
    R = 0

    def __init__(self, w: Dict[str, torch.Tensor], num_heads: int, patch_size: int = 16, eps: float = 1e-5):
        self.w = {k: v.detach().float().cpu() for k, v in w.items()}
        self.num_heads, self.patch_size, self.eps = num_heads, patch_size, eps
        self.depth = 1 + max(int(k.split(".")[1]) for k in self.w if k.startswith("blocks."))
        self.Dv = self.w["cls_token"].shape[-1]
        self.D = self.w["head.weight"].shape[0]      # width of the tokens handed on (512 for ViT-B)
        self.last_qkv = None

    def pos_embed_for(self, n_h: int, n_w: int) -> torch.Tensor:
        """timm.layers.resample_abs_pos_embed (public; called from vision_transformer.checkpoint_filter_fn when
        ``timm.create_model(..., img_size=resize_dim)``, P/src/model.py:371, meets a checkpoint of another grid): the prefix
        (class) position is kept apart, the grid goes through F.interpolate(size=, mode='bicubic', antialias=True)."""
        pe = self.w["pos_embed"].reshape(1, -1, self.Dv)
        g = int(math.isqrt(pe.shape[1] - 1))
        if (n_h, n_w) == (g, g):
            return pe
        grid = pe[:, 1:].reshape(1, g, g, self.Dv).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(n_h, n_w), mode="bicubic", antialias=True)
        return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, n_h * n_w, self.Dv)], dim=1)

    def forward_features(self, imgs: torch.Tensor) -> torch.Tensor:
        w, D, h = self.w, self.Dv, self.num_heads
        B = imgs.shape[0]
        x = F.conv2d(imgs.float(), w["patch_embed.proj.weight"], w.get("patch_embed.proj.bias"), stride=self.patch_size)
        x = x.flatten(2).transpose(1, 2)
        x = torch.cat([w["cls_token"].reshape(1, 1, D).expand(B, -1, -1), x], dim=1)
        x = x + self.pos_embed_for(imgs.shape[-2] // self.patch_size, imgs.shape[-1] // self.patch_size)
        x = F.layer_norm(x, (D,), w["norm_pre.weight"], w["norm_pre.bias"], self.eps)
        T = x.shape[1]
        for i in range(self.depth):
            pre = "blocks.%d." % i
            y = F.layer_norm(x, (D,), w[pre + "norm1.weight"], w[pre + "norm1.bias"], self.eps)
            q, k, v = F.linear(y, w[pre + "attn.qkv.weight"], w[pre + "attn.qkv.bias"]).reshape(B, T, 3, h, D // h).permute(2, 0, 3, 1, 4)
            a = ((q * (D // h) ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
            y = (a @ v).transpose(1, 2).reshape(B, T, D)
            x = x + F.linear(y, w[pre + "attn.proj.weight"], w[pre + "attn.proj.bias"])
            y = F.layer_norm(x, (D,), w[pre + "norm2.weight"], w[pre + "norm2.bias"], self.eps)
            y = F.linear(y, w[pre + "mlp.fc1.weight"], w[pre + "mlp.fc1.bias"])
            y = y * torch.sigmoid(1.702 * y)                     # QuickGELU, model.py:363-365
            x = x + F.linear(y, w[pre + "mlp.fc2.weight"], w[pre + "mlp.fc2.bias"])
        return F.layer_norm(x, (D,), w["norm.weight"], w["norm.bias"], self.eps)

    def __call__(self, imgs: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = F.linear(self.forward_features(imgs), self.w["head.weight"], self.w.get("head.bias"))   # model.py:790
        return {"x_norm_clstoken": out[:, 0], "x_norm_regtokens": out[:, 1:1], "x_norm_patchtokens": out[:, 1:]}


# --------------------------------------------------------------------------------------------
# a4 / a5  CLS-row attention read-out and attention-weighted means
# --------------------------------------------------------------------------------------------


def process_self_attention(qkv_out: torch.Tensor, batch_size: int, num_tokens: int, num_attn_heads: int,
                           embed_dim: int, scale: float, num_global_tokens: int):
    """P/src/dino_extraction.py:24-34 (materialises the full [B,H,T,T] product, as the reference)."""
    qkv = qkv_out.reshape(batch_size, num_tokens, 3, num_attn_heads,
                          embed_dim // num_attn_heads).permute(2, 0, 3, 1, 4)
    q, k = qkv[0] * scale, qkv[1]
    attn = q @ k.transpose(-2, -1)
    maps = attn[:, :, 0, num_global_tokens:]
    self_attn = maps.mean(dim=1).softmax(dim=-1)
    return self_attn, maps


def attention_weighted_means(self_attn, maps, patch_tokens):
    """P/src/model.py:869-872."""
    avg_self_attn_token = (self_attn.unsqueeze(-1) * patch_tokens).mean(dim=1)
    maps_sm = maps.softmax(dim=-1)
    disentangled = (patch_tokens.unsqueeze(1) * maps_sm.unsqueeze(-1)).mean(dim=2)
    return avg_self_attn_token, disentangled


# --------------------------------------------------------------------------------------------
# a6  traces
# --------------------------------------------------------------------------------------------


def map_traces_to_grid(traces: Sequence[dict], n_patch: int) -> torch.Tensor:
    """P/src/bbox_utils.py:158-168 (double-precision ``int(x / (1.0/n))``, clamp, closed [0,1])."""
    grid = torch.zeros((n_patch, n_patch))
    patch_size = 1.0 / n_patch
    for tr in traces:
        x, y = tr["x"], tr["y"]
        if 0 <= x <= 1 and 0 <= y <= 1:
            gx, gy = int(x / patch_size), int(y / patch_size)
            grid[min(gy, n_patch - 1), min(gx, n_patch - 1)] += 1
    return grid


def trace_embeds(patch_tokens: torch.Tensor, traces: Sequence[Sequence[dict]],
                 self_attn: Optional[torch.Tensor] = None) -> torch.Tensor:
    """P/src/model.py:1049-1054: mean over n*n cells (NOT over the point count)."""
    bs, n2, D = patch_tokens.shape
    n = int(n2 ** 0.5)
    rel = torch.stack([map_traces_to_grid(t, n) for t in traces], dim=0)
    if self_attn is not None:
        rel = self_attn.view(rel.shape) * rel
    return (rel.unsqueeze(-1) * patch_tokens.view(bs, n, n, D)).mean(dim=(1, 2))


# --------------------------------------------------------------------------------------------
# a7 / a8  boxes and whole-image gaussian
# --------------------------------------------------------------------------------------------


def extract_bboxes_feats_double_dino(vit: "DinoV2Oracle", patch_embeddings: torch.Tensor, bboxes: torch.Tensor,
                                     cls_token: Optional[torch.Tensor], registers_tokens: Optional[torch.Tensor],
                                     patch_size: int, return_type: str = "cls", gaussian_bbox_variance: float = 0.5):
    """P/src/bbox_utils.py:300-403: per (image, box) re-run the LAST block on [cls | registers | region patches] of
    the final tokens.  Quirks kept: the xywh tensor is floor-divided by the patch size (on a clone) and then read as
    (x1, y1, x2, y2) with inclusive python slices; "gaussian_avg" weights the block's INPUT patches, normalised to
    sum 1 (an empty region gives zeros), "avg" is the mean of the block's OUTPUT region rows (NaN when empty)."""
    N, N_boxes = patch_embeddings.shape[0], bboxes.shape[1]
    g = int(patch_embeddings.shape[1] ** 0.5)
    D = patch_embeddings.shape[-1]
    idx = bboxes.clone()
    idx //= patch_size
    idx = idx.int()
    pe = patch_embeddings.view(N, g, g, D)
    if cls_token is not None:
        off = 5 if registers_tokens is not None else 1
    else:
        assert return_type != "cls"
        off = 0
    last = vit.depth - 1
    means = []
    for i in range(N):
        image_means = []
        for j in range(N_boxes):
            region_xy = pe[i, idx[i, j, 1]:idx[i, j, 3] + 1, idx[i, j, 0]:idx[i, j, 2] + 1, :]
            region = region_xy.reshape(1, -1, D)
            parts = []
            if cls_token is not None:
                parts.append(cls_token[i].reshape(1, 1, D))
                if registers_tokens is not None:
                    parts.append(registers_tokens[i].reshape(1, 4, D))
            parts.append(region)
            outputs = vit.block(last, torch.cat(parts, dim=1))
            rows = outputs[0, off:]
            if return_type == "gaussian_avg":
                h_span, w_span = region_xy.shape[:2]
                yc, xc = torch.meshgrid(torch.linspace(-1, 1, h_span), torch.linspace(-1, 1, w_span), indexing="ij")
                wts = torch.exp(-(xc ** 2 + yc ** 2) / gaussian_bbox_variance)
                wts = wts / wts.sum()
                mean = (region_xy * wts.unsqueeze(-1)).sum(dim=(0, 1))
            elif return_type == "avg":
                mean = rows.mean(dim=0)
            else:
                mean = outputs[0, 0]
            image_means.append(mean)
        means.append(torch.stack(image_means))
    return torch.stack(means)


def extract_bboxes_feats(patch_embeddings: torch.Tensor, bboxes: torch.Tensor, gaussian_avg=False,
                         gaussian_bbox_variance=0.5, get_single_embedding_per_image=False,
                         patch_size=14, attention_map: Optional[torch.Tensor] = None,
                         rng: Optional[random.Random] = None):
    """P/src/bbox_utils.py:8-109, including: in-place ``bboxes //= patch_size`` on the caller's
    tensor, inclusive slices with python clamping / negative wrap, in-place renormalisation of the
    attention slice (affects later overlapping boxes), NaN for empty regions."""
    N, N_boxes = patch_embeddings.shape[0], bboxes.shape[1]
    g = int(patch_embeddings.shape[1] ** 0.5)
    rng = rng or random
    bboxes //= patch_size
    bb = bboxes.int()
    pe = patch_embeddings.view(N, g, g, -1)
    if attention_map is not None:
        attention_map = attention_map.view(N, g, g)
    total = torch.zeros(N, g, g)
    x1, y1, w, h = bb.unbind(-1)
    x2, y2 = x1 + w, y1 + h
    means = []
    for i in range(N):
        image_means = []
        for j in range(N_boxes):
            if bb[i, j].sum().item() < 0 and get_single_embedding_per_image:
                continue
            ys, xs = slice(int(y1[i, j]), int(y2[i, j]) + 1), slice(int(x1[i, j]), int(x2[i, j]) + 1)
            region = pe[i, ys, xs, :]
            if attention_map is not None:
                pw = attention_map[i, ys, xs]
                pw /= pw.sum()
                total[i, ys, xs] += pw
                mean = (region * pw.unsqueeze(-1)).sum(dim=(0, 1))
            elif gaussian_avg:
                hs, ws = region.shape[:2]
                yc, xc = torch.meshgrid(torch.linspace(-1, 1, hs), torch.linspace(-1, 1, ws), indexing="ij")
                if gaussian_bbox_variance == 0:
                    pw = torch.zeros((hs, ws))
                    cy = [hs // 2] if hs % 2 == 1 else [hs // 2 - 1, hs // 2]
                    cx = [ws // 2] if ws % 2 == 1 else [ws // 2 - 1, ws // 2]
                    pw[rng.choice(cy), rng.choice(cx)] = 1.0
                else:
                    pw = torch.exp(-(xc ** 2 + yc ** 2) / gaussian_bbox_variance)
                    pw = pw / pw.sum()
                mean = (region * pw.unsqueeze(-1)).sum(dim=(0, 1))
                total[i, ys, xs] += pw
            else:
                hs, ws = region.shape[:2]
                total[i, ys, xs] += torch.ones(hs, ws) / (hs * ws)
                mean = region.mean(dim=(0, 1))
            image_means.append(mean)
        if not get_single_embedding_per_image:
            means.append(torch.stack(image_means))
    total /= total.sum(dim=(1, 2), keepdim=True)
    if not get_single_embedding_per_image:
        return torch.stack(means)
    return (total.unsqueeze(-1) * pe).sum(dim=(1, 2))


def compute_region_means(patch_embeddings: torch.Tensor, variance: float,
                         rng: Optional[random.Random] = None) -> torch.Tensor:
    """P/src/model.py:45-94."""
    N = patch_embeddings.shape[0]
    g = int(patch_embeddings.shape[1] ** 0.5)
    pe = patch_embeddings.view(N, g, g, -1)
    rng = rng or random
    lin = torch.linspace(-1, 1, g)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    if variance == 0:
        pw = torch.zeros(N, g, g)
        opts = [g // 2] if g % 2 == 1 else [g // 2 - 1, g // 2]
        for i in range(N):
            cy = rng.choice(opts)
            cx = rng.choice(opts)
            pw[i, cy, cx] = 1.0
    elif variance >= 100:
        pw = torch.full((N, g, g), 1 / (g * g))
    else:
        wts = torch.exp(-(xx ** 2 + yy ** 2) / variance)
        pw = (wts / wts.sum()).unsqueeze(0).expand(N, -1, -1)
    return (pe * pw.unsqueeze(-1)).sum(dim=(1, 2))


# --------------------------------------------------------------------------------------------
# a9 / a10  memory projection and pseudo-inverse inversion
# --------------------------------------------------------------------------------------------


def load_bank_rows(embs: torch.Tensor) -> torch.Tensor:
    """Im2TxtProjector.__init__ (im2txtprojection.py:343-345): rows of zero norm are dropped when the bank is loaded
    (the caption texts are NOT filtered along: project(return_argmax_text=True) indexes the unfiltered list)."""
    return embs[embs.norm(dim=-1) != 0]


def project(image_embedding: torch.Tensor, bank: torch.Tensor, temperature: float = 0.01,
            normalize: bool = False, return_n_best_sims: Optional[int] = None, return_argmax_text: bool = False,
            text_dataset=None):
    """P/src/decap/im2txtprojection/im2txtprojection.py:353-385 (three passes over the bank; the
    query is L2-normalised IN PLACE on the caller's tensor, line 368).  ``return_argmax_text`` (:371-375): the text of
    the most similar row instead of the projected embedding."""
    bank_n = bank / bank.norm(dim=-1, keepdim=True)
    image_embedding /= image_embedding.norm(dim=-1, keepdim=True)
    sim = image_embedding @ bank_n.T.float()
    if return_argmax_text:
        argmax_texts = [text_dataset[int(idx)].decode() for idx in sim.argmax(dim=-1)]
        if return_n_best_sims:
            return argmax_texts, sim.sort(dim=-1, descending=True).values[:, :return_n_best_sims].tolist()
        return argmax_texts
    sm = (sim / temperature).softmax(dim=-1)
    out = sm @ bank.float()
    if normalize:
        out /= out.norm(dim=-1, keepdim=True)
    if return_n_best_sims:
        return out, sim.sort(dim=-1, descending=True).values[:, :return_n_best_sims].tolist()
    return out


def ctx_cleaner(dirty_embeds: torch.Tensor, ctx_embed: torch.Tensor, cleaning_type="orthogonal_projection", alpha=1.0,
                epsilon=1e-6):
    """P/src/model.py:1425-1436: remove the context direction from [B, S, D] embeddings (ctx_embed [B, D])."""
    ctx = ctx_embed.unsqueeze(1)
    if cleaning_type == "orthogonal_projection":
        projection = (dirty_embeds @ ctx.transpose(-1, -2)) / (torch.norm(ctx, dim=-1, keepdim=True) ** 2)
        return dirty_embeds - alpha * projection * ctx
    if cleaning_type == "contrastive_mask":
        return dirty_embeds * (1 - (ctx / (torch.norm(ctx, p=2, dim=2, keepdim=True) + epsilon)))
    return None


def project_clip_txt(x: torch.Tensor, sd: dict, act=torch.tanh) -> torch.Tensor:
    """P/src/talk2dino/talk2dino.py:73-83: linear_layer, then act + hidden layer for every hidden layer (the bank builder's
    step between CLIP's text features and the stored rows, im2txtprojection.py:520-523)."""
    x = x.float() @ sd["linear_layer.weight"].t() + sd["linear_layer.bias"]
    k = 0
    while "hidden_layers.%d.weight" % k in sd:
        if act is not None:
            x = act(x)
        x = x @ sd["hidden_layers.%d.weight" % k].t() + sd["hidden_layers.%d.bias" % k]
        k += 1
    return x


def get_pseudo_inverse(A: torch.Tensor) -> torch.Tensor:
    """P/src/embedding_utils.py:3-15."""
    U, S, Vh = torch.linalg.svd(A, full_matrices=False)
    S_pinv = torch.zeros_like(S)
    nz = S > 1e-10
    S_pinv[nz] = 1.0 / S[nz]
    return Vh.T @ torch.diag(S_pinv) @ U.T


def revert_transformation(features, A_pinv, b):
    """P/src/embedding_utils.py:17-25."""
    return (features - b) @ A_pinv.t()


# --------------------------------------------------------------------------------------------
# a11 / a12  DeCap decoder (GPT-2 4L/4H/768, one-token prefix), greedy, NO KV cache
# --------------------------------------------------------------------------------------------


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


class DeCapOracle:
    """``DeCap`` (P/src/decap/decap.py:61-79) + the GPT2LMHeadModel math it instantiates
    (transformers==4.46.3 ``modeling_gpt2``: Conv1D = x @ W[in,out] + b, pre-LN blocks, eps 1e-5,
    gelu_new, causal softmax(QK^T / sqrt(d_head)), tied LM head).  ``w`` uses the checkpoint's own
    key names (``clip_project.model.0.*``, ``decoder.transformer.*``)."""

    def __init__(self, w: Dict[str, torch.Tensor], n_head: int = 4, eps: float = 1e-5):
        self.w = {k: v.detach().float().cpu() for k, v in w.items()}
        self.n_head = n_head
        self.eps = eps
        self.n_layer = 1 + max(int(k.split(".")[3]) for k in self.w if k.startswith("decoder.transformer.h."))
        self.E = self.w["decoder.transformer.wte.weight"].shape[1]

    def clip_project(self, x):
        return F.linear(x, self.w["clip_project.model.0.weight"], self.w["clip_project.model.0.bias"])

    def gpt2_logits(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        """Full forward over [N,S,E] -> logits [N,S,V] (all positions, as the reference executes)."""
        w, E, H = self.w, self.E, self.n_head
        N, S, _ = inputs_embeds.shape
        x = inputs_embeds + w["decoder.transformer.wpe.weight"][:S].unsqueeze(0)
        mask = torch.tril(torch.ones(S, S, dtype=torch.bool))
        for l in range(self.n_layer):
            p = "decoder.transformer.h.%d." % l
            y = F.layer_norm(x, (E,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], self.eps)
            qkv = y @ w[p + "attn.c_attn.weight"] + w[p + "attn.c_attn.bias"]
            q, k, v = qkv.split(E, dim=2)
            q = q.view(N, S, H, E // H).transpose(1, 2)
            k = k.view(N, S, H, E // H).transpose(1, 2)
            v = v.view(N, S, H, E // H).transpose(1, 2)
            att = (q @ k.transpose(-1, -2)) / math.sqrt(E // H)
            att = att.masked_fill(~mask, torch.finfo(att.dtype).min).softmax(dim=-1)
            y = (att @ v).transpose(1, 2).reshape(N, S, E)
            y = y @ w[p + "attn.c_proj.weight"] + w[p + "attn.c_proj.bias"]
            x = x + y
            y = F.layer_norm(x, (E,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], self.eps)
            y = gelu_new(y @ w[p + "mlp.c_fc.weight"] + w[p + "mlp.c_fc.bias"])
            y = y @ w[p + "mlp.c_proj.weight"] + w[p + "mlp.c_proj.bias"]
            x = x + y
        x = F.layer_norm(x, (E,), w["decoder.transformer.ln_f.weight"], w["decoder.transformer.ln_f.bias"],
                         self.eps)
        return x @ w["decoder.transformer.wte.weight"].t()

    def gpt2_logits_cached(self, inputs_embeds: torch.Tensor, past=None):
        """GPT2LMHeadModel(inputs_embeds=..., past_key_values=past, use_cache=True) as the ViECap greedy search calls it
        (P/src/viecap/search.py:152-164): the new positions attend to the cached keys / values and to themselves causally.
        -> (logits of the LAST new position [N, V], new past = list of (keys, values) [N, H, S_total, hd] per layer)."""
        w, E, H = self.w, self.E, self.n_head
        N, S, _ = inputs_embeds.shape
        S0 = 0 if past is None else past[0][0].shape[2]
        x = inputs_embeds + w["decoder.transformer.wpe.weight"][S0:S0 + S].unsqueeze(0)
        mask = torch.tril(torch.ones(S0 + S, S0 + S, dtype=torch.bool))[S0:]
        new_past = []
        for l in range(self.n_layer):
            p = "decoder.transformer.h.%d." % l
            y = F.layer_norm(x, (E,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], self.eps)
            qkv = y @ w[p + "attn.c_attn.weight"] + w[p + "attn.c_attn.bias"]
            q, k, v = qkv.split(E, dim=2)
            q = q.view(N, S, H, E // H).transpose(1, 2)
            k = k.view(N, S, H, E // H).transpose(1, 2)
            v = v.view(N, S, H, E // H).transpose(1, 2)
            if past is not None:
                k = torch.cat((past[l][0], k), dim=2)
                v = torch.cat((past[l][1], v), dim=2)
            new_past.append((k, v))
            att = (q @ k.transpose(-1, -2)) / math.sqrt(E // H)
            att = att.masked_fill(~mask, torch.finfo(att.dtype).min).softmax(dim=-1)
            y = (att @ v).transpose(1, 2).reshape(N, S, E)
            y = y @ w[p + "attn.c_proj.weight"] + w[p + "attn.c_proj.bias"]
            x = x + y
            y = F.layer_norm(x, (E,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], self.eps)
            y = gelu_new(y @ w[p + "mlp.c_fc.weight"] + w[p + "mlp.c_fc.bias"])
            y = y @ w[p + "mlp.c_proj.weight"] + w[p + "mlp.c_proj.bias"]
            x = x + y
        x = F.layer_norm(x[:, -1], (E,), w["decoder.transformer.ln_f.weight"], w["decoder.transformer.ln_f.bias"], self.eps)
        return x @ w["decoder.transformer.wte.weight"].t(), new_past

    fast = False        # tests with hundreds of prefixes set this: the same arithmetic through gpt2_logits_cached

    def logits_after(self, clip_features: torch.Tensor, history: torch.Tensor) -> torch.Tensor:
        """Logits [V] of the position that follows ``history`` (ids already decoded, possibly none) for ONE prefix."""
        emb = self.clip_project(clip_features.reshape(1, -1)).view(1, 1, -1)
        if history.numel():
            emb = torch.cat((emb, self.w["decoder.transformer.wte.weight"][history.long()].unsqueeze(0)), dim=1)
        return self.gpt2_logits(emb)[0, -1]

    def decode_ids(self, clip_features: torch.Tensor, entry_length: int = 30, cached: Optional[bool] = None):
        """P/src/decap/decap.py:116-155: 30 full forwards over the growing sequence; returns
        (ids [N,30] int64, per-token log-probs [N,30], top-2 logit margin [N,30]).
        ``cached`` (default: the ``fast`` attribute) keeps the keys / values of the positions already decoded instead of
        recomputing them every step as the reference does -- the same sums per position in the same order, 15x less CPU
        work at 30 steps; tests/test_oracle_golden.py holds both forms to the reference's golden ids."""
        emb = self.clip_project(clip_features).view(clip_features.shape[0], 1, -1)
        wte = self.w["decoder.transformer.wte.weight"]
        ids, lps, margins = [], [], []
        cached = self.fast if cached is None else cached
        past, new = None, emb
        for _ in range(entry_length):
            if cached:
                logits, past = self.gpt2_logits_cached(new, past)
            else:
                logits = self.gpt2_logits(emb)[:, -1, :]
            probs = F.softmax(logits, -1)
            nxt = torch.argmax(probs, -1).unsqueeze(1)
            lps.append(torch.log(probs).gather(1, nxt))
            top2 = logits.topk(2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]).unsqueeze(1))
            ids.append(nxt)
            new = wte[nxt]
            if not cached:
                emb = torch.cat((emb, new), dim=1)
        return torch.cat(ids, 1), torch.cat(lps, 1), torch.cat(margins, 1)


def ids_to_captions(ids, decode_fn, return_start_end_tokens=False) -> Optional[List[str]]:
    """P/src/decap/decap.py:162-181: one failure (KeyError for id >= 49408) voids the whole batch."""
    try:
        outs = []
        for row in ids:
            s = decode_fn([int(t) for t in row])
            s = s.split("<|endoftext|>")[0]
            if not return_start_end_tokens:
                s = s.replace("<|startoftext|>", "")
            else:
                s += "<|endoftext|>"
            outs.append(s)
        return outs
    except Exception:
        return None


# --------------------------------------------------------------------------------------------
# a14 / a15  routing: caption_tokens and forward (DINO + DeCap/CapDec configs)
# --------------------------------------------------------------------------------------------


class PatchionerOracle:
    """The reference ``Patchioner`` glue for the DINOv2 + DeCap/CapDec configs
    (P/src/model.py:718-1058, 1392-1423), over the oracle pieces above."""

    def __init__(self, vit: DinoV2Oracle, decoder: DeCapOracle, bank: Optional[torch.Tensor],
                 decode_fn, normalize: bool = True, crop_dim: int = 224, num_attn_heads: int = 16,
                 scale: float = 0.125, A_pinv=None, b=None):
        self.vit, self.decoder, self.bank = vit, decoder, bank
        self.decode_fn = decode_fn
        self.normalize = normalize
        self.patch_size = vit.patch_size
        self.num_global_tokens = 1 + vit.R
        self.num_tokens = self.num_global_tokens + crop_dim // self.patch_size * crop_dim // self.patch_size
        self.embed_dim = vit.D
        self.num_attn_heads, self.scale = num_attn_heads, scale
        self.A_pinv, self.b = A_pinv, b
        self.last_ids = None
        self.call_log = None            # tests: a list collects the greedy ids of every caption_tokens call

    def caption_tokens(self, tokens, project_flag=True, compute_scores=False):
        if self.bank is None:
            project_flag = False
        x = project(tokens, self.bank, normalize=self.normalize) if project_flag else tokens
        if self.A_pinv is not None:
            x = revert_transformation(x, self.A_pinv, self.b)
        ids, lps, _ = self.decoder.decode_ids(x)
        self.last_ids = ids
        if self.call_log is not None:
            self.call_log.append(ids.clone())
        if getattr(self, "prefix_log", None) is not None:      # tests: the fp32 path's decoder inputs, for the derived parity bound
            self.prefix_log.append(x.detach().clone())
        caps = ids_to_captions(ids.tolist(), self.decode_fn)
        if compute_scores:
            return caps, torch.exp(lps.sum(-1)).tolist()
        return caps

    def forward(self, imgs, get_cls_capt=True, get_avg_self_attn_capt=False, get_attn_heads_capt=False,
                get_patch_capts=False, get_register_capts=False, bboxes=None, traces=None,
                get_controllable_capts=False, bs_factor=4, gaussian_avg=False, gaussian_bbox_variance=0.5,
                get_avg_patch_capt=False, gaussian_img_variance=1, use_attn_map_for_bboxes=False,
                use_attention_tracing=False, compute_scores=False, cleaning_type=None, clean_after_projection=True,
                alpha=1.0, clean_from="cls"):
        outs = {}
        bs = imgs.shape[0]
        d = self.vit(imgs)
        patches = d["x_norm_patchtokens"]
        if getattr(self.vit, "has_attention", True):
            self_attn, maps = process_self_attention(self.vit.last_qkv, bs, self.num_tokens, self.num_attn_heads,
                                                     self.embed_dim, self.scale, self.num_global_tokens)
            avg_tok, disentangled = attention_weighted_means(self_attn, maps, patches)
        else:       # model.py:864-872: nothing defines these for a backbone without the qkv hook; their readers fail
            self_attn = maps = avg_tok = disentangled = None
            if clean_from != "cls":
                clean_from = "cls"                       # model.py:886-890: fallback to the cls token
        if cleaning_type is not None:
            # P/src/model.py:879-922.  project() normalises its argument IN PLACE: with clean_after_projection the
            # patch tokens and the clean-from token of the backbone output are left L2-normalised.
            cf = d["x_norm_clstoken"] if clean_from == "cls" else avg_tok
            cleaned = []
            for i in range(bs):
                p_i, c_i = patches[i:i + 1], cf[i:i + 1]
                if clean_after_projection:
                    cleaned.append(ctx_cleaner(project(p_i, self.bank, normalize=True), project(c_i, self.bank, normalize=True),
                                               cleaning_type=cleaning_type, alpha=alpha))
                else:
                    cleaned.append(project(ctx_cleaner(p_i / p_i.norm(dim=-1, keepdim=True), c_i / c_i.norm(dim=-1, keepdim=True),
                                                       cleaning_type=cleaning_type, alpha=alpha), self.bank, normalize=True))
            patches = torch.cat(cleaned, dim=0)
        noproj = cleaning_type is not None            # patch / box captions skip the projection after cleaning
        D = patches.shape[-1]

        def put(key, ret):
            if compute_scores:
                outs[key], outs[key + "_scores"] = ret
            else:
                outs[key] = ret

        if get_cls_capt:
            put("cls_capt", self.caption_tokens(d["x_norm_clstoken"], compute_scores=compute_scores))
        if get_avg_self_attn_capt:
            put("avg_self_attn_capt", self.caption_tokens(avg_tok, compute_scores=compute_scores))
        if get_avg_patch_capt:
            put("avg_patch_capt", self.caption_tokens(compute_region_means(patches, gaussian_img_variance),
                                                      compute_scores=compute_scores))
        # P/src/model.py:946-975: nested per-image lists of 16 head / n*n patch / 4 register captions
        def nested(key, score_key, toks, per, project_flag=True):
            ret = self.caption_tokens(toks, project_flag=project_flag, compute_scores=compute_scores)
            caps = ret[0] if compute_scores else ret
            outs[key] = [caps[i * per:(i + 1) * per] for i in range(bs)]
            if compute_scores:
                outs[score_key] = [ret[1][i * per:(i + 1) * per] for i in range(bs)]

        if get_attn_heads_capt:
            nested("attn_heads_capts", "attn_heads_scores", disentangled.reshape(-1, D), self.num_attn_heads)
        if get_patch_capts:
            nested("patch_tokens_capts", "patch_tokens_scores", patches.reshape(-1, D), patches.shape[1], project_flag=not noproj)
        if get_register_capts:
            nested("register_capts", "register_scores", d["x_norm_regtokens"].reshape(-1, D), 4)
        if bboxes is not None and not get_controllable_capts:
            n_boxes = bboxes.shape[1]
            amap = self_attn if use_attn_map_for_bboxes else None
            feats = extract_bboxes_feats(patches, bboxes, gaussian_avg=gaussian_avg,
                                         gaussian_bbox_variance=gaussian_bbox_variance,
                                         patch_size=self.patch_size, attention_map=amap).view(-1, D)
            bbox_bs = bs * bs_factor
            n_batch = math.ceil(feats.shape[0] / bbox_bs)
            caps, scores = [], []
            for i in range(n_batch):
                s = i * bbox_bs
                e = s + bbox_bs if i < n_batch - 1 else feats.shape[0]
                ret = self.caption_tokens(feats[s:e], project_flag=not noproj, compute_scores=compute_scores)
                if compute_scores:
                    caps.extend(ret[0]); scores.extend(ret[1])
                else:
                    caps.extend(ret)
            outs["bbox_capts"] = [caps[i * n_boxes:(i + 1) * n_boxes] for i in range(bs)]
            if compute_scores:
                outs["bbox_scores"] = [scores[i * n_boxes:(i + 1) * n_boxes] for i in range(bs)]
        elif bboxes is not None and get_controllable_capts:
            amap = self_attn if use_attn_map_for_bboxes else None
            feats = extract_bboxes_feats(patches, bboxes, gaussian_avg=gaussian_avg,
                                         gaussian_bbox_variance=gaussian_bbox_variance,
                                         get_single_embedding_per_image=True, patch_size=self.patch_size,
                                         attention_map=amap)
            outs["set_controllable_capts"] = self.caption_tokens(feats)
        if traces is not None:
            emb = trace_embeds(patches, traces, self_attn if use_attention_tracing else None)
            outs["trace_capts"] = self.caption_tokens(emb)
        return outs


# --------------------------------------------------------------------------------------------
# f1  ViECap head (P/src/viecap): mapping network, entity retrieval, hard prompt, greedy search
# --------------------------------------------------------------------------------------------


class ViECapOracle:
    """``VieCap.forward`` (P/src/viecap/entrypoint.py:98-153) for the GPT-2 / greedy-search configuration, with the pieces
    it calls restated: ``MappingNetwork`` (ClipCap.py:122-153 over :8-120), ``image_text_simiarlity`` / ``top_k_categories``
    (retrieval_categories.py:61-116), ``compose_discrete_prompts`` (utils.py:55-74), ``greedy_search`` (search.py:108-191,
    with the KV cache the reference itself uses there).  ``w`` = the
    checkpoint's state dict (``mapping_network.*``, ``gpt.*``); ``tokenizer`` = any object with ``encode`` / ``decode`` /
    ``pad_token_id``."""

    def __init__(self, w: Dict[str, torch.Tensor], tokenizer, entities_text: Sequence[str], texts_embeddings: torch.Tensor,
                 continuous_prompt_length: int = 10, clip_project_length: int = 10, temperature: float = 0.01, top_k: int = 3,
                 threshold: float = 0.2, using_hard_prompt: bool = False, soft_prompt_first: bool = False,
                 map_heads: int = 8, gpt_heads: int = 12, only_hard_prompt: bool = False):
        self.w = {k: v.detach().float().cpu() for k, v in w.items()}
        self.tok = tokenizer
        self.entities_text = list(entities_text)
        self.texts_embeddings = texts_embeddings.detach().float().cpu().clone()
        self.Lc, self.Lp = continuous_prompt_length, clip_project_length
        self.temperature, self.top_k, self.threshold = temperature, top_k, threshold
        self.using_hard_prompt, self.soft_prompt_first = using_hard_prompt, soft_prompt_first
        self.only_hard_prompt = only_hard_prompt                      # entrypoint.py:130-131: the word embeddings alone
        self.map_heads = map_heads
        self.gpt = DeCapOracle({("decoder." + k[4:]): v for k, v in self.w.items() if k.startswith("gpt.")}, n_head=gpt_heads)
        self.map_layers = 1 + max(int(k.split(".")[3]) for k in self.w if k.startswith("mapping_network.transformer.layers."))
        self.last = {}

    # ---- ClipCap.py:122-153
    def mapping_network(self, x: torch.Tensor) -> torch.Tensor:
        w = self.w
        E = w["mapping_network.prefix_const"].shape[1]
        h = F.linear(x, w["mapping_network.linear.weight"], w["mapping_network.linear.bias"]).view(x.shape[0], self.Lp, -1)
        prefix = w["mapping_network.prefix_const"].unsqueeze(0).expand(x.shape[0], -1, -1)
        q = torch.cat((h, prefix), dim=1)
        H = self.map_heads
        for l in range(self.map_layers):
            p = "mapping_network.transformer.layers.%d." % l
            y = F.layer_norm(q, (E,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)
            b, n, _ = y.shape
            queries = F.linear(y, w[p + "attn.to_queries.weight"]).reshape(b, n, H, E // H)
            kv = F.linear(y, w[p + "attn.to_keys_values.weight"]).reshape(b, n, 2, H, E // H)
            keys, values = kv[:, :, 0], kv[:, :, 1]
            att = torch.einsum("bnhd,bmhd->bnmh", queries, keys) * (E // H) ** -0.5
            att = att.softmax(dim=2)
            o = torch.einsum("bnmh,bmhd->bnhd", att, values).reshape(b, n, E)
            q = q + F.linear(o, w[p + "attn.project.weight"], w[p + "attn.project.bias"])
            y = F.layer_norm(q, (E,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-5)
            y = F.linear(F.relu(F.linear(y, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])), w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
            q = q + y
        return q[:, self.Lp:, :]

    # ---- retrieval_categories.py:61-116
    def entity_probs(self, feats: torch.Tensor) -> torch.Tensor:
        f = feats.float().clone()
        t = self.texts_embeddings
        f /= f.norm(dim=-1, keepdim=True)
        t /= t.norm(dim=-1, keepdim=True)            # in place on the stored embeddings, as the reference does per call
        return torch.softmax(f @ t.transpose(1, 0) / self.temperature, dim=-1)

    def detect(self, probs_row: torch.Tensor) -> List[str]:
        vals, idx = torch.topk(probs_row, k=self.top_k, dim=-1)
        out = []
        for j in range(self.top_k):
            if vals[j] < self.threshold:
                break
            out.append(self.entities_text[int(idx[j])])
        return out

    @staticmethod
    def prompt_text(entities: Sequence[str]) -> str:      # utils.py:55-74
        if len(entities) == 0:
            return "There are something in image."
        s = ""
        for e in entities:
            s += " " + e + ","
        return "There are" + s[:-1] + " in image."

    def forward(self, image_features: torch.Tensor):
        """-> captions (a str for a single feature, search.py:172-181); ``self.last``: soft prompt, entity probabilities,
        prompt token ids (padded), generated ids [N, 64], top-2 logit margins per step."""
        pad_id = self.tok.pad_token_id if self.tok.pad_token_id is not None else 0
        image_features /= image_features.norm(2, dim=-1, keepdim=True)           # in place (entrypoint.py:108)
        cont = self.mapping_network(image_features)
        wte = self.gpt.w["decoder.transformer.wte.weight"]
        tokens = None
        if self.using_hard_prompt:
            probs = self.entity_probs(image_features)
            rows = [self.tok.encode(self.prompt_text(self.detect(probs[i]))) for i in range(image_features.shape[0])]
            L = max(len(r) for r in rows)
            tokens = torch.full((len(rows), L), pad_id, dtype=torch.long)
            for i, r in enumerate(rows):
                tokens[i, :len(r)] = torch.tensor(r)
            disc = wte[tokens]
            if self.only_hard_prompt:                                  # entrypoint.py:130-131
                emb = disc
            else:
                emb = torch.cat((cont, disc), dim=1) if self.soft_prompt_first else torch.cat((disc, cont), dim=1)
            self.last["entity_probs"] = probs
        else:
            emb = cont
        eos = [self.tok.encode(e)[-1] for e in (".", " .")]
        ids, margins = [], []
        b = emb.shape[0]
        early = None
        logits, past = self.gpt.gpt2_logits_cached(emb)            # search.py:152-155: the prompt
        for step in range(64):                                    # search.py:150-181
            nxt = torch.argmax(logits, dim=-1, keepdim=True)
            top2 = logits.topk(2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]).unsqueeze(1))
            ids.append(nxt)
            if b == 1 and int(nxt) in eos and early is None:
                early = step                                      # the reference returns here; decoding on changes nothing it returns
            if step < 63:
                logits, past = self.gpt.gpt2_logits_cached(wte[nxt], past)
        ids, margins = torch.cat(ids, 1), torch.cat(margins, 1)
        self.last.update(cont=cont, prompt_tokens=tokens, ids=ids, margins=margins)
        if b == 1:
            r = ids[0].tolist()
            return self.tok.decode(r[:early + 1] if early is not None else r)
        out = []
        for r in ids.tolist():
            i = len(r) - 1
            for j, t in enumerate(r):
                if t in eos:
                    i = j
                    break
            out.append(self.tok.decode(r[:i + 1]))
        return out

    def prompt_embeddings(self, image_features: torch.Tensor) -> torch.Tensor:
        """entrypoint.py:108-135 alone: the prompt the search starts from ([N, P, E]; features normalised in place)."""
        pad_id = self.tok.pad_token_id if self.tok.pad_token_id is not None else 0
        image_features /= image_features.norm(2, dim=-1, keepdim=True)
        cont = self.mapping_network(image_features)
        if not self.using_hard_prompt:
            return cont
        wte = self.gpt.w["decoder.transformer.wte.weight"]
        probs = self.entity_probs(image_features)
        rows = [self.tok.encode(self.prompt_text(self.detect(probs[i]))) for i in range(image_features.shape[0])]
        L = max(len(r) for r in rows)
        tokens = torch.full((len(rows), L), pad_id, dtype=torch.long)
        for i, r in enumerate(rows):
            tokens[i, :len(r)] = torch.tensor(r)
        disc = wte[tokens]
        if self.only_hard_prompt:
            return disc
        return torch.cat((cont, disc), dim=1) if self.soft_prompt_first else torch.cat((disc, cont), dim=1)

    def beam_search(self, embeddings: torch.Tensor, beam_width: int = 5, max_len: int = 64, end_of_sentences=(".", " .")) -> List[str]:
        """``beam_search`` (P/src/viecap/search.py:193-285) for one prompt [1, P, E], statement by statement; the only change is
        that the language model keeps a KV cache (re-indexed by ``next_tokens_source``) instead of re-running every beam's whole
        sequence -- the same logits (the reference's greedy search itself uses the cache).  ``self.last_beam``: per beam, in the
        order of the last selection, (ids, length, score), the order returned, and per selection the margin between the last
        chosen and the first rejected candidate."""
        wte = self.gpt.w["decoder.transformer.wte.weight"]
        eos = [self.tok.encode(e)[-1] for e in end_of_sentences]
        scores = None
        tokens = None
        seq_lengths = torch.ones(beam_width)
        is_stopped = torch.zeros(beam_width, dtype=torch.bool)
        generated_new = embeddings.float()                       # the positions not yet run through the model
        past = None
        margins = []
        for i in range(max_len):
            logits, past = self.gpt.gpt2_logits_cached(generated_new, past)     # :244-245, last position
            logits = logits / 1.0                                               # temperature 1.0
            logits = logits.softmax(-1).log()                                   # :246
            if scores is None:
                top = logits.topk(beam_width + 1, -1)
                margins.append(float(top.values[0, beam_width - 1] - top.values[0, beam_width]))
                scores, next_tokens = logits.topk(beam_width, -1)               # :248
                past = [(k.expand(beam_width, *k.shape[1:]), v.expand(beam_width, *v.shape[1:])) for k, v in past]   # :249 generated.expand
                next_tokens, scores = next_tokens.permute(1, 0), scores.squeeze(0)
                tokens = next_tokens
            else:
                logits[is_stopped] = -float("inf")
                logits[is_stopped, 0] = 0
                scores_sum = scores[:, None] + logits
                seq_lengths[~is_stopped] += 1
                scores_sum_average = scores_sum / seq_lengths[:, None]
                top = scores_sum_average.view(-1).topk(beam_width + 1, -1)
                margins.append(float(top.values[beam_width - 1] - top.values[beam_width]))
                scores_sum_average, next_tokens = scores_sum_average.view(-1).topk(beam_width, -1)
                next_tokens_source = torch.div(next_tokens, scores_sum.shape[1], rounding_mode="trunc")
                seq_lengths = seq_lengths[next_tokens_source]
                next_tokens = next_tokens % scores_sum.shape[1]
                next_tokens = next_tokens.unsqueeze(1)
                tokens = tokens[next_tokens_source]
                tokens = torch.cat((tokens, next_tokens), dim=1)
                past = [(k[next_tokens_source], v[next_tokens_source]) for k, v in past]        # :267 generated[next_tokens_source]
                scores = scores_sum_average * seq_lengths
                is_stopped = is_stopped[next_tokens_source]
            generated_new = wte[next_tokens.squeeze()].view(beam_width, 1, -1)  # :271-274
            is_stopped = is_stopped + (next_tokens.eq(eos[0]) | next_tokens.eq(eos[1])).squeeze()
            if is_stopped.all():
                break
        scores = scores / seq_lengths
        output_list = tokens.numpy()
        texts = [self.tok.decode([int(t) for t in out[:int(n)]]) for out, n in zip(output_list, seq_lengths)]
        order = scores.argsort(descending=True)
        self.last_beam = dict(ids=[[int(t) for t in out[:int(n)]] for out, n in zip(output_list, seq_lengths)],
                              lengths=seq_lengths.clone(), scores=scores.clone(), order=order.tolist(), margins=margins)
        return [texts[i] for i in order]

    def compute_perplexity(self, sentences) -> List[float]:
        """``VieCap.compute_perplexity`` (P/src/viecap/entrypoint.py:155-172): each sentence is tokenised again and run
        through the language model with ``labels = input_ids``: loss = mean cross-entropy of token p+1 given tokens <= p
        over the L - 1 positions (GPT2LMHeadModel shifts by one), perplexity = exp(loss); one token: mean of nothing, NaN."""
        out = []
        for sentence in sentences:
            ids = torch.tensor(self.tok.encode(sentence), dtype=torch.long)
            if ids.numel() < 2:
                out.append(float("nan"))
                continue
            logits = self._logits_of_tokens(ids)
            loss = F.cross_entropy(logits[:-1], ids[1:])
            out.append(float(torch.exp(loss)))
        return out

    def _logits_of_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        """GPT2LMHeadModel(input_ids=ids).logits [L, V]: word + position embeddings, the causal stack, the tied head -- one
        full forward over the sentence, as the reference runs it."""
        wte = self.gpt.w["decoder.transformer.wte.weight"]
        return self.gpt.gpt2_logits(wte[ids].unsqueeze(0))[0]
