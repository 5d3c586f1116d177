"""Seeded synthetic weights in the reference checkpoints' own key names.

There is no network on the build or GPU boxes, so the DINOv2 hub weights
(P/src/model.py:342-343), the DeCap decoder checkpoint (P/src/decap/decap.py:188-222) and the
text memory bank (P/src/decap/im2txtprojection/im2txtprojection.py:387-407) are replaced by
deterministic synthetic tensors of the same shapes and names.  Real checkpoints
load through the same dict interface (see ``model.Patchioner.from_config``).

All generators use a CPU ``torch.Generator`` so the numbers are identical on every box.
"""
from __future__ import annotations

from typing import Dict

import torch

DINO_ARCHS = {
    # name fragment -> (embed_dim, depth, heads)
    "vits14": (384, 12, 6),
    "vitb14": (768, 12, 12),
    "vitl14": (1024, 24, 16),
}


def dino_arch(dino_model: str):
    for k, v in DINO_ARCHS.items():
        if k in dino_model:
            return v
    raise ValueError("unsupported DINOv2 model name %r" % (dino_model,))


def _tn(g, shape, std):
    t = torch.empty(shape)
    torch.nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std, generator=g)
    return t


def synth_dinov2(seed: int, dino_model: str = "dinov2_vitb14_reg", depth: int | None = None,
                 pretrain_grid: int = 37) -> Dict[str, torch.Tensor]:
    """State dict shaped like ``torch.hub.load('facebookresearch/dinov2', dino_model).state_dict()``."""
    D, full_depth, _ = dino_arch(dino_model)
    depth = depth or full_depth
    R = 4 if "reg" in dino_model else 0
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}
    w["cls_token"] = _tn(g, (1, 1, D), 0.02)
    w["pos_embed"] = _tn(g, (1, 1 + pretrain_grid * pretrain_grid, D), 0.02)
    if R:
        w["register_tokens"] = _tn(g, (1, R, D), 0.02)
    w["mask_token"] = torch.zeros(1, D)
    w["patch_embed.proj.weight"] = _tn(g, (D, 3, 14, 14), 0.02)
    w["patch_embed.proj.bias"] = _tn(g, (D,), 0.02)
    for i in range(depth):
        p = "blocks.%d." % i
        for ln in ("norm1", "norm2"):
            w[p + ln + ".weight"] = 0.5 + torch.rand(D, generator=g)
            w[p + ln + ".bias"] = _tn(g, (D,), 0.02)
        # q/k rows scaled up so the attention maps are peaky (as trained DINOv2 maps are)
        qkv = _tn(g, (3 * D, D), 0.02)
        qkv[: 2 * D] *= 2.5
        w[p + "attn.qkv.weight"] = qkv
        w[p + "attn.qkv.bias"] = _tn(g, (3 * D,), 0.02)
        w[p + "attn.proj.weight"] = _tn(g, (D, D), 0.02)
        w[p + "attn.proj.bias"] = _tn(g, (D,), 0.02)
        w[p + "ls1.gamma"] = 0.1 * (0.5 + torch.rand(D, generator=g))
        w[p + "mlp.fc1.weight"] = _tn(g, (4 * D, D), 0.02)
        w[p + "mlp.fc1.bias"] = _tn(g, (4 * D,), 0.02)
        w[p + "mlp.fc2.weight"] = _tn(g, (D, 4 * D), 0.02)
        w[p + "mlp.fc2.bias"] = _tn(g, (D,), 0.02)
        w[p + "ls2.gamma"] = 0.1 * (0.5 + torch.rand(D, generator=g))
    w["norm.weight"] = 0.5 + torch.rand(D, generator=g)
    w["norm.bias"] = _tn(g, (D,), 0.02)
    return w


CLIP_ARCHS = {
    # timm model name fragment -> (embed_dim, depth, heads, patch, head width)
    "vit_base_patch16_clip_224": (768, 12, 12, 16, 512),
    "vit_base_patch32_clip_224": (768, 12, 12, 32, 512),
    "vit_large_patch14_clip_224": (1024, 24, 16, 14, 768),
}


def clip_arch(name: str):
    for k, v in CLIP_ARCHS.items():
        if k in name:
            return v
    raise ValueError("unsupported OpenAI-CLIP timm model name %r" % (name,))


def synth_clip_vit(seed: int, name: str = "vit_base_patch16_clip_224.openai", depth: int | None = None,
                   img: int = 224) -> Dict[str, torch.Tensor]:
    """State dict shaped like ``timm.create_model(name, act_layer=QuickGELU).state_dict()`` (P/src/model.py:371): the OpenAI
    CLIP vision tower in timm's VisionTransformer layout -- bias-free patch conv, cls token, learned positions for
    [cls | patches], norm_pre, pre-LN blocks without LayerScale, final norm, bias-free head to the joint space."""
    D, full_depth, _, p, out = clip_arch(name)
    depth = depth or full_depth
    n = img // p
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}
    w["cls_token"] = _tn(g, (1, 1, D), 0.02)
    w["pos_embed"] = _tn(g, (1, 1 + n * n, D), 0.02)
    w["patch_embed.proj.weight"] = _tn(g, (D, 3, p, p), 0.02)
    w["norm_pre.weight"] = 0.5 + torch.rand(D, generator=g)
    w["norm_pre.bias"] = _tn(g, (D,), 0.02)
    for i in range(depth):
        q = "blocks.%d." % i
        for ln in ("norm1", "norm2"):
            w[q + ln + ".weight"] = 0.5 + torch.rand(D, generator=g)
            w[q + ln + ".bias"] = _tn(g, (D,), 0.02)
        qkv = _tn(g, (3 * D, D), 0.02)
        qkv[: 2 * D] *= 2.5
        w[q + "attn.qkv.weight"] = qkv
        w[q + "attn.qkv.bias"] = _tn(g, (3 * D,), 0.02)
        w[q + "attn.proj.weight"] = _tn(g, (D, D), 0.02) * 0.3        # no LayerScale: small branch outputs keep the stream tame
        w[q + "attn.proj.bias"] = _tn(g, (D,), 0.02)
        w[q + "mlp.fc1.weight"] = _tn(g, (4 * D, D), 0.02)
        w[q + "mlp.fc1.bias"] = _tn(g, (4 * D,), 0.02)
        w[q + "mlp.fc2.weight"] = _tn(g, (D, 4 * D), 0.02) * 0.3
        w[q + "mlp.fc2.bias"] = _tn(g, (D,), 0.02)
    w["norm.weight"] = 0.5 + torch.rand(D, generator=g)
    w["norm.bias"] = _tn(g, (D,), 0.02)
    w["head.weight"] = _tn(g, (out, D), 0.03)
    return w


def add_outlier_channels(w: Dict[str, torch.Tensor], channels=(7, 300, 611), block: int = 1, gain: float = 4000.0,
                         hidden_rows=(11, 2000), hidden_gain: float = 30.0) -> Dict[str, torch.Tensor]:
    """DINOv2-like massive activations for stress tests: the MLP branch of ``block`` is multiplied by ``gain`` on a few
    channels (from there on the fp32 residual stream carries values in the hundreds on those channels and every later
    LayerNorm sees a few dominant channels), and a few hidden units of every later MLP get ``hidden_gain``-times larger
    pre-activations (large fc2 operands).  Returns a modified copy."""
    w = dict(w)
    depth = 1 + max(int(k.split(".")[1]) for k in w if k.startswith("blocks."))
    g2 = w["blocks.%d.ls2.gamma" % block].clone()
    g2[list(channels)] *= gain
    w["blocks.%d.ls2.gamma" % block] = g2
    for i in range(block + 1, depth):
        f = w["blocks.%d.mlp.fc1.weight" % i].clone()
        f[list(hidden_rows)] *= hidden_gain
        w["blocks.%d.mlp.fc1.weight" % i] = f
    return w


def synth_decap(seed: int, prefix_size: int = 768, n_layer: int = 4, n_embd: int = 768,
                vocab: int = 50257, n_positions: int = 1024, tok_vocab: int = 49408,
                std: float = 0.02) -> Dict[str, torch.Tensor]:
    """State dict shaped like a DeCap checkpoint (``clip_project`` + ``decoder`` = GPT2LMHeadModel).

    Embedding rows >= ``tok_vocab`` are zero: the LM head is tied, so their logits are exactly 0 and
    a random-weight argmax never emits an id the CLIP tokenizer cannot decode.
    """
    g = torch.Generator().manual_seed(seed)
    E = n_embd

    def n(*shape):
        return torch.randn(*shape, generator=g) * std

    w: Dict[str, torch.Tensor] = {}
    w["clip_project.model.0.weight"] = n(E, prefix_size)
    w["clip_project.model.0.bias"] = n(E)
    wte = n(vocab, E)
    wte[tok_vocab:] = 0
    w["decoder.transformer.wte.weight"] = wte
    w["decoder.transformer.wpe.weight"] = n(n_positions, E)
    for l in range(n_layer):
        p = "decoder.transformer.h.%d." % l
        w[p + "ln_1.weight"] = 0.5 + torch.rand(E, generator=g)
        w[p + "ln_1.bias"] = n(E)
        w[p + "attn.c_attn.weight"] = n(E, 3 * E)      # Conv1D: [in, out]
        w[p + "attn.c_attn.bias"] = n(3 * E)
        w[p + "attn.c_proj.weight"] = n(E, E)
        w[p + "attn.c_proj.bias"] = n(E)
        w[p + "ln_2.weight"] = 0.5 + torch.rand(E, generator=g)
        w[p + "ln_2.bias"] = n(E)
        w[p + "mlp.c_fc.weight"] = n(E, 4 * E)
        w[p + "mlp.c_fc.bias"] = n(4 * E)
        w[p + "mlp.c_proj.weight"] = n(4 * E, E)
        w[p + "mlp.c_proj.bias"] = n(E)
    w["decoder.transformer.ln_f.weight"] = 0.5 + torch.rand(E, generator=g)
    w["decoder.transformer.ln_f.bias"] = n(E)
    w["decoder.lm_head.weight"] = wte  # tied
    return w


def synth_bank(seed: int, rows: int, dim: int = 768) -> torch.Tensor:
    """Text-embedding memory bank ``[rows, dim] ~ N(0,1)`` (un-normalised, as the DINOv2 configs keep it,
    P/src/model.py:174)."""
    g = torch.Generator().manual_seed(seed)
    out = torch.empty(rows, dim)
    step = 65536
    for s in range(0, rows, step):
        e = min(rows, s + step)
        out[s:e] = torch.randn(e - s, dim, generator=g)
    return out


def synth_images(seed: int, batch: int, size: int = 224) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, size, size, generator=g)


# ---------------------------------------------------------------------------------------------------------------------
# ViECap head (P/src/viecap): seeded stand-ins for the checkpoint, the GPT-2 tokenizer files and the entity vocabulary
# ---------------------------------------------------------------------------------------------------------------------
def synth_viecap(seed: int, clip_hidden_size: int = 768, n_layer: int = 12, n_embd: int = 768, vocab: int = 50257,
                 n_positions: int = 1024, tok_vocab: int = 50257, prompt_len: int = 10, project_len: int = 10,
                 map_layers: int = 8, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """State dict shaped like ``ClipCaptionModel.state_dict()`` (P/src/viecap/ClipCap.py:155-200): ``mapping_network.*``
    (Linear C -> project_len x 768, prefix_const, 8 transformer layers with bias-free q / kv projections and a ratio-2 ReLU
    MLP) and ``gpt.*`` (GPT2LMHeadModel, GPT2Config() sizes).  Embedding rows >= ``tok_vocab`` are zero (tied head: logit 0),
    so a random-weight arg-max never leaves the tokenizer's vocabulary."""
    g = torch.Generator().manual_seed(seed)
    E = n_embd

    def n(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    w: Dict[str, torch.Tensor] = {}
    w["mapping_network.linear.weight"] = n(project_len * E, clip_hidden_size, s=0.05)
    w["mapping_network.linear.bias"] = n(project_len * E)
    w["mapping_network.prefix_const"] = n(prompt_len, E, s=0.5)
    for l in range(map_layers):
        p = "mapping_network.transformer.layers.%d." % l
        w[p + "norm1.weight"] = 0.5 + torch.rand(E, generator=g)
        w[p + "norm1.bias"] = n(E)
        w[p + "attn.to_queries.weight"] = n(E, E, s=0.04)
        w[p + "attn.to_keys_values.weight"] = n(2 * E, E, s=0.04)
        w[p + "attn.project.weight"] = n(E, E)
        w[p + "attn.project.bias"] = n(E)
        w[p + "norm2.weight"] = 0.5 + torch.rand(E, generator=g)
        w[p + "norm2.bias"] = n(E)
        w[p + "mlp.fc1.weight"] = n(2 * E, E)
        w[p + "mlp.fc1.bias"] = n(2 * E)
        w[p + "mlp.fc2.weight"] = n(E, 2 * E)
        w[p + "mlp.fc2.bias"] = n(E)
    wte = n(vocab, E)
    wte[tok_vocab:] = 0
    w["gpt.transformer.wte.weight"] = wte
    w["gpt.transformer.wpe.weight"] = n(n_positions, E)
    for l in range(n_layer):
        p = "gpt.transformer.h.%d." % l
        w[p + "ln_1.weight"] = 0.5 + torch.rand(E, generator=g)
        w[p + "ln_1.bias"] = n(E)
        w[p + "attn.c_attn.weight"] = n(E, 3 * E)      # Conv1D: [in, out]
        w[p + "attn.c_attn.bias"] = n(3 * E)
        w[p + "attn.c_proj.weight"] = n(E, E)
        w[p + "attn.c_proj.bias"] = n(E)
        w[p + "ln_2.weight"] = 0.5 + torch.rand(E, generator=g)
        w[p + "ln_2.bias"] = n(E)
        w[p + "mlp.c_fc.weight"] = n(E, 4 * E)
        w[p + "mlp.c_fc.bias"] = n(4 * E)
        w[p + "mlp.c_proj.weight"] = n(4 * E, E)
        w[p + "mlp.c_proj.bias"] = n(E)
    w["gpt.transformer.ln_f.weight"] = 0.5 + torch.rand(E, generator=g)
    w["gpt.transformer.ln_f.bias"] = n(E)
    w["gpt.lm_head.weight"] = wte  # tied
    return w


SYNTH_ENTITIES = ["person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
                  "fire hydrant", "stop sign", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra",
                  "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "kite", "baseball bat"]


def synth_entity_embeddings(seed: int, count: int, dim: int) -> torch.Tensor:
    return torch.randn(count, dim, generator=torch.Generator().manual_seed(seed))


def synth_bpe(seed: int = 0, merges: int = 400):
    """A small byte-level BPE vocabulary in GPT-2's file format (vocab dict, ranked merges) learnt on the words the ViECap
    hard prompt is made of ("There are ... in image.", the entity names) plus seeded filler words: the real ``vocab.json`` /
    ``merges.txt`` are not on this image.  -> (vocab: token -> id, merges: list of pairs); ids 0..255 are the byte tokens."""
    import random
    from .viecap import ByteLevelBPE, _bytes_to_unicode
    import regex
    rng = random.Random(seed)
    b2u = _bytes_to_unicode()
    words = ["There are something in image.", "There are " + ", ".join(SYNTH_ENTITIES) + " in image.", " .", "."]
    letters = "abcdefghijklmnopqrstuvwxyz"
    for _ in range(300):
        words.append(" " + "".join(rng.choice(letters) for _ in range(rng.randint(2, 8))))
    pat = regex.compile(ByteLevelBPE.PATTERN)
    corpus = []
    for t in words:
        for piece in pat.findall(t):
            corpus.append(["".join(b2u[b] for b in piece.encode("utf-8"))][0])
    seqs = [list(w) for w in corpus]
    vocab = {b2u[b]: b for b in range(256)}
    merge_list = []
    for _ in range(merges):
        counts = {}
        for sq in seqs:
            for a, b in zip(sq, sq[1:]):
                counts[(a, b)] = counts.get((a, b), 0) + 1
        if not counts:
            break
        best = max(sorted(counts), key=lambda p: counts[p])
        if counts[best] < 2:
            break
        merge_list.append(best)
        vocab[best[0] + best[1]] = len(vocab)
        for sq in seqs:
            i = 0
            while i < len(sq) - 1:
                if sq[i] == best[0] and sq[i + 1] == best[1]:
                    sq[i:i + 2] = [best[0] + best[1]]
                else:
                    i += 1
    return vocab, merge_list
