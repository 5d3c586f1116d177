"""Seeded input builders shared by ``tools/oracle/gen_golden.py`` (which runs the REFERENCE on them in
the build container and commits the outputs under ``tests/golden/``) and by the tests (which rebuild
the same inputs and compare the oracle / the HIP path with those committed outputs).

Pure data generation: nothing here imports the reference or the oracle.
"""
import math

import numpy as np
import torch


def randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ---- A: CLS-attention read-out ------------------------------------------------------------------
ATTN = dict(B=2, n=16, D=768, heads=16, scale=0.125, G=5, seed_qkv=11, seed_patch=12)


def attn_inputs():
    c = ATTN
    T = c["G"] + c["n"] ** 2
    return randn(c["seed_qkv"], c["B"], T, 3 * c["D"]), randn(c["seed_patch"], c["B"], c["n"] ** 2, c["D"])


# ---- B: traces ----------------------------------------------------------------------------------
def trace_cases():
    """list of (n_patch, [ {x,y}, ... ])"""
    rng = np.random.RandomState(5)
    cases = []
    # closed interval ends, outside points, clamping
    cases.append((16, [dict(x=0.0, y=0.0), dict(x=1.0, y=1.0), dict(x=1.2, y=0.5), dict(x=-0.1, y=0.3),
                       dict(x=0.5, y=0.25), dict(x=0.9999999, y=0.0625), dict(x=0.0625, y=0.9999999),
                       dict(x=0.5, y=1.0000001)]))
    # n = 37: 1/37 is inexact, int(x / (1/37)) differs from int(x*37) for some k/37
    cases.append((37, [dict(x=k / 37, y=(37 - k) / 37) for k in range(38)]))
    cases.append((37, [dict(x=float(a), y=float(b), t=i) for i, (a, b) in enumerate(rng.rand(64, 2))]))
    cases.append((16, [dict(x=float(a), y=float(b)) for a, b in rng.rand(200, 2) * 1.2 - 0.1]))
    cases.append((10, [dict(x=k / 10, y=k / 10) for k in range(11)]))
    cases.append((16, []))
    return cases


def block_trace(cx, cy, n=16, side=4):
    """16-point trace on a side x side patch block = the metric's "16-patch region" (SURVEY 8d C2)."""
    return [dict(x=(cx + i + 0.5) / n, y=(cy + j + 0.5) / n) for j in range(side) for i in range(side)]


# ---- C: boxes -----------------------------------------------------------------------------------
BOX = dict(N=3, n=16, D=64, seed_patch=21, seed_attn=22, patch_size=14)


def box_patches():
    c = BOX
    return randn(c["seed_patch"], c["N"], c["n"] ** 2, c["D"])


def box_attn():
    c = BOX
    return torch.softmax(randn(c["seed_attn"], c["N"], c["n"] ** 2), dim=-1)


def boxes_regular():
    """xywh in crop-pixel coordinates: interior, fractional, touching / exceeding the border, 1-pixel pad."""
    return torch.tensor([
        [[14.0, 28.0, 42.0, 42.0], [0.0, 0.0, 1.0, 1.0], [100.5, 60.2, 80.7, 33.3], [200.0, 200.0, 60.0, 60.0], [0.0, 0.0, 223.0, 223.0]],
        [[70.0, 14.0, 13.0, 100.0], [210.0, 0.0, 13.9, 223.9], [56.0, 56.0, 0.0, 0.0], [0.0, 0.0, 1.0, 1.0], [30.0, 150.0, 120.0, 40.0]],
        [[1.0, 1.0, 221.0, 12.0], [112.0, 112.0, 111.0, 111.0], [13.9, 13.9, 14.1, 14.1], [0.0, 0.0, 1.0, 1.0], [180.0, 20.0, 30.0, 170.0]],
    ])


def boxes_with_dummies():
    b = boxes_regular()
    b[0, 3] = -1.0
    b[1, 1] = -1.0
    b[1, 4] = -1.0
    b[2, 0] = -1.0
    return b


def boxes_odd_spans():
    """spans (w//14+1, h//14+1) all odd -> the var==0 centre pick is deterministic."""
    return torch.tensor([
        [[14.0, 14.0, 28.0, 28.0], [0.0, 0.0, 1.0, 1.0], [56.0, 28.0, 56.0, 84.0]],
        [[28.0, 70.0, 84.0, 28.0], [140.0, 140.0, 28.0, 56.0], [0.0, 0.0, 1.0, 1.0]],
        [[98.0, 14.0, 0.0, 112.0], [0.0, 0.0, 1.0, 1.0], [42.0, 42.0, 112.0, 112.0]],
    ])


# ---- D: whole-image gaussian --------------------------------------------------------------------
REGION_VARIANCES = [0.5, 1, 2.5, 100, 1000]


# ---- E: memory projection -----------------------------------------------------------------------
PROJ = dict(M=4096, D=768, N=8, seed_bank=31, seed_q=32)


def proj_inputs(clustered=False):
    c = PROJ
    bank = randn(c["seed_bank"], c["M"], c["D"])
    q = randn(c["seed_q"], c["N"], c["D"])
    if clustered:
        # text-like bank: rows concentrate around 32 centres with varying norms; queries near centres
        centres = randn(c["seed_bank"] + 100, 32, c["D"])
        idx = torch.arange(c["M"]) % 32
        bank = (centres[idx] + 0.35 * bank) * (0.5 + torch.rand(c["M"], 1, generator=torch.Generator().manual_seed(7)))
        bank[5] = 0.0 * bank[5] + 1e-3 * bank[6]        # a tiny-norm row (zero rows are dropped at load)
        q = centres[:c["N"]] * 3.0 + q
    return bank, q


# ---- F: decoder ---------------------------------------------------------------------------------
DEC = dict(seed_w=3, N=16, seed_x=41)


def decoder_prefixes(kind):
    x = randn(DEC["seed_x"], DEC["N"], 768)
    if kind == "unit":          # DeCap config: projector output is L2-normalised
        return x / x.norm(dim=-1, keepdim=True)
    if kind == "raw":           # CapDec config: raw patch feature, no normalisation
        return x
    raise ValueError(kind)


# ---- H: end to end ------------------------------------------------------------------------------
E2E = dict(B=4, depth=2, seed_vit=51, seed_dec=3, seed_img=0, seed_bank=61, M=2048, crop=224)


def e2e_traces():
    rng = np.random.RandomState(2)
    out = []
    for b in range(E2E["B"]):
        cx, cy = rng.randint(0, 13, size=2)
        out.append(block_trace(int(cx), int(cy)))
    return out


def e2e_boxes():
    rng = np.random.RandomState(3)
    B = E2E["B"]
    xy = rng.randint(0, 13, size=(B, 3, 2)) * 14.0
    wh = rng.randint(1, 8, size=(B, 3, 2)) * 14.0
    b = np.concatenate([xy, wh], axis=-1).astype(np.float32)
    b[:, 2] = [0.0, 0.0, 1.0, 1.0]      # dense-captioning pad box (eval_densecap.py:332-333)
    return torch.tensor(b)


# ---- image transforms -------------------------------------------------------------------
# (w, h, new_w, new_h): down- and up-scaling, odd sizes, extreme aspect, identity on one axis
PREP_RESIZE_CASES = [(64, 48, 30, 22), (50, 37, 29, 22), (10, 8, 28, 22), (33, 50, 22, 33), (22, 22, 22, 22),
                     (5, 100, 22, 440), (96, 72, 22, 22), (7, 5, 22, 22), (40, 30, 40, 17)]
# (w, h, resize_dim, crop_dim) for the whole transform, incl. resize_dim < crop_dim (zero padding, odd differences)
PREP_TRANSFORM_CASES = [(640, 480, 224, 224), (375, 500, 256, 224), (100, 300, 224, 224), (224, 224, 224, 224),
                        (900, 200, 518, 518), (60, 50, 224, 224), (200, 150, 160, 224), (201, 150, 160, 224),
                        (150, 313, 100, 224), (300, 100, 37, 224), (1023, 767, 224, 224)]


def prep_image(seed, w, h):
    """Seeded RGB uint8 image with smooth structure plus noise (exercises clipping at 0 / 255)."""
    rng = np.random.RandomState(1000 + seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127.5 + 127.5 * np.sin(xx[..., None] / (3.0 + np.arange(3)) + yy[..., None] / (5.0 - np.arange(3)))
    img = base + rng.randint(-90, 91, size=(h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


# ---- double-DINO boxes --------------------------------------------------------------------
DDINO = dict(B=3, depth=2, seed_w=77, seed_tok=78, variance=0.5)


def ddino_tokens():
    """final (normed) tokens [B, 261, 768] fed to extract_bboxes_feats_double_dino: unit-scale like a LayerNorm output"""
    return randn(DDINO["seed_tok"], DDINO["B"], 261, 768)


# ---- ViT-S read-out: 6 heads of 64 channels (P/src/model.py:336) -------------------------------------------
ATTN_VITS = dict(B=2, n=16, D=384, heads=6, scale=0.125, G=5, seed_qkv=13, seed_patch=14)


def attn_vits_inputs():
    c = ATTN_VITS
    T = c["G"] + c["n"] ** 2
    return randn(c["seed_qkv"], c["B"], T, 3 * c["D"]), randn(c["seed_patch"], c["B"], c["n"] ** 2, c["D"])


# ---- ctx_cleaner (P/src/model.py:1425-1436) ----------------------------------------------------------------
CTX = dict(B=3, S=9, D=96, seed=31)


def ctx_inputs():
    c = CTX
    return randn(c["seed"], c["B"], c["S"], c["D"]), randn(c["seed"] + 1, c["B"], c["D"])


# ---- memory-bank file (HDF5 as the reference writes it) -----------------------------------------
H5BANK = dict(name="coco", rows=40, dim=768, seed=77, zero_rows=(7, 23))


def h5_bank_case():
    """(embeddings [40, 768] float32 with two all-zero rows, 40 captions incl. non-ASCII and an empty one)."""
    c = H5BANK
    emb = randn(c["seed"], c["rows"], c["dim"]).numpy().astype(np.float32)
    for r in c["zero_rows"]:
        emb[r] = 0.0
    words = ["a dog", "two cats on a sofa", "caf\u00e9 au lait", "\u5c71\u306e\u4e0a\u306e\u96ea", "a man riding a wave on top of a surfboard",
             "", "na\u00efve r\u00e9sum\u00e9 \u2014 d\u00e9j\u00e0 vu", "x" * 300]
    texts = ["%s #%d" % (words[i % len(words)], i) if words[i % len(words)] else "" for i in range(c["rows"])]
    return emb, texts


# ---- ViECap head --------------------------------------------------------------------------------
VIECAP = dict(seed_w=301, seed_ent=302, seed_x=303, seed_bpe=0, C=768, gpt_layers=12, N=6, temperature=0.01, top_k=3, threshold=0.4)


def viecap_case():
    """(weights, tokenizer vocabulary + merges, entity names, entity embeddings, features [6, 768]) of the ViECap fixture:
    hard prompts of different lengths -- one entity, two entities, a two-word entity, none (the 'something' prompt) -- and
    plain noise."""
    from patchioner_amd import weights as W
    c = VIECAP
    vocab, merges = W.synth_bpe(c["seed_bpe"])
    w = W.synth_viecap(c["seed_w"], clip_hidden_size=c["C"], n_layer=c["gpt_layers"], tok_vocab=len(vocab))
    ents = list(W.SYNTH_ENTITIES)
    emb = W.synth_entity_embeddings(c["seed_ent"], len(ents), c["C"])
    e = emb / emb.norm(dim=-1, keepdim=True)
    x = randn(c["seed_x"], c["N"], c["C"]) * 0.02
    x[0] += e[3]                                  # one entity
    x[1] = 3.0 * (1.002 * e[10] + e[20])          # two entities at nearly equal cosine (0.52 / 0.48): both pass the 0.4 threshold,
                                                  # and their order does not hang on the last bit of a dot product
    x[2] += e[9]                                  # a two-word entity (longer hard prompt)
    x[3] += 2.0 * e[15] + 1.9 * e[16]
    x[4] = e.sum(0)                               # every entity about equally (un)likely: nothing passes -> "something"
    return w, (vocab, merges), ents, emb, x
