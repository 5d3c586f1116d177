"""f4: the memory-bank builder (patchioner_amd/bank.py = the tail of Im2TxtProjector._build_support_memory,
P/src/decap/im2txtprojection/im2txtprojection.py:511-555): Talk2DINO's text projection on the device vs the oracle, the
.h5 it writes read back through the model's own loader, and the projection through the freshly built bank."""
import os

import numpy as np
import pytest
import torch

from patchioner_amd import h5lite, weights as W
from patchioner_amd.bank import build_memory_bank
from patchioner_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import patchioner_oracle as O
    return O


def _t2d(seed, cin=512, d=768, hidden=True):
    g = torch.Generator().manual_seed(seed)
    sd = {"linear_layer.weight": torch.randn(d, cin, generator=g) * 0.05, "linear_layer.bias": torch.randn(d, generator=g) * 0.02}
    if hidden:
        sd["hidden_layers.0.weight"] = torch.randn(d, d, generator=g) * 0.04
        sd["hidden_layers.0.bias"] = torch.randn(d, generator=g) * 0.02
    return sd


def _encode_text(texts, cin=512):
    """a stand-in for clip_model.encode_text(tokenize(texts)): a deterministic feature per caption"""
    rows = []
    for t in texts:
        g = torch.Generator().manual_seed(sum(t.encode()) * 7919 + len(t))
        rows.append(torch.randn(cin, generator=g))
    return torch.stack(rows)


@pytest.mark.parametrize("act,hidden", [("tanh", True), ("relu", True), ("sigmoid", True), (None, True), ("tanh", False)])
def test_text_projection_vs_oracle(O, act, hidden):
    e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=1)
    try:
        sd = _t2d(3, hidden=hidden)
        x = torch.randn(1003, 512, generator=torch.Generator().manual_seed(4))      # not a multiple of the 64-row tile
        got = e.text_project(x, sd["linear_layer.weight"], sd["linear_layer.bias"], sd.get("hidden_layers.0.weight"),
                             sd.get("hidden_layers.0.bias"), act=act).cpu()
        fn = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, None: None}[act]
        want = O.project_clip_txt(x, sd, fn)
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=1e-5)
    finally:
        e.close()


def test_built_bank_file_round_trips_through_the_model(O, tmp_path):
    from patchioner_amd.model import Patchioner, load_memory_bank
    texts = ["a %s %s on the %s" % (a, b, c) for a in ("red", "blue", "tall", "small", "wet") for b in ("dog", "cat", "bus", "tree")
             for c in ("grass", "road", "table", "roof", "beach", "snow", "hill")]                       # 140 captions
    sd = _t2d(9)
    e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=1)
    try:
        path = build_memory_bank(e, texts, _encode_text, str(tmp_path), "coco_captions", None, 150, talk2dino_state_dict=sd,
                                 batch_size=64)
    finally:
        e.close()
    assert os.path.basename(path) == "coco_captions_text_embeddings-B16-ViT-B.16-150.h5"       # im2txtprojection.py:234
    assert h5lite.dataset_names(path) == ["coco_captions-embeddings", "coco_captions-text"]
    bank, t = load_memory_bank(path, want_texts=True)
    assert bank.shape == (150, 768) and [x.decode() for x in t] == texts + [""] * 10
    want = O.project_clip_txt(_encode_text(texts), sd)
    np.testing.assert_allclose(bank[:140].numpy(), want.numpy(), rtol=2e-5, atol=1e-5)
    assert float(bank[140:].abs().max()) == 0.0            # rows never written (:546-553); the loader's norm filter drops them
    # the model opens the file it was pointed at and projects through it like the oracle does through the same rows
    m = Patchioner.from_config({"decap_weights": W.synth_decap(3), "prefix_size": 768, "support_memory_size": 150,
                                "dino_model": "dinov2_vitb14_reg", "memory_bank": path, "resize_dim": 224, "crop_dim": 224,
                                "dino_weights": W.synth_dinov2(1, "dinov2_vitb14_reg", depth=1), "calculate_argmax_text": True},
                               "cuda")
    q = torch.randn(5, 768, generator=torch.Generator().manual_seed(2))
    kept = O.load_bank_rows(bank)
    assert kept.shape[0] == 140
    got = m.engine.project(q.clone().cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), O.project(q.clone(), kept).numpy(), rtol=2e-4, atol=2e-6)
    caps = m.caption_tokens(want[[7, 77]].clone().cuda())
    assert caps == [texts[7], texts[77]]                 # arg-max text of a bank row is its own caption
