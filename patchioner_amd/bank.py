"""Building the text memory bank -- the offline half of ``Im2TxtProjector`` (P/src/decap/im2txtprojection/
im2txtprojection.py:448-560): captions -> CLIP text features -> Talk2DINO's text projection -> ``<name>-embeddings`` /
``<name>-text`` in an HDF5 file whose name encodes the configuration (:100-170, :234).

What runs where:
  * the CLIP text tower is NOT part of this package (bank build needs CLIP + the caption datasets); the
    caller hands in ``encode_text(list of str) -> [n, clip_dim] float tensor`` -- ``lambda t: clip_model.encode_text(
    clip.tokenize(t))`` in the reference's terms (:519-520);
  * ``project_clip_txt`` (talk2dino.py:73-83) runs on the MI355X through ``pio_text_project`` (exact fp32 MFMA GEMMs);
  * the file is written by ``h5lite.write_bank`` (no h5py on the image) in the layout h5py gives the reference's
    ``create_dataset`` calls, and read back by ``h5lite.read_datasets`` / ``Patchioner(memory_bank=...)``.

Reference behaviour kept: batches of ``batch_size`` texts (:511-524); the datasets are created with SUPPORT_MEMORY_SIZE
rows whatever the number of texts (:546-547), so a short corpus leaves zero rows / empty strings behind -- the loader's
``norm != 0`` filter (:343-345) drops them again; random sampling of the corpus is the caller's (``random.sample``, :456).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import h5lite

PROJECTION_TYPES = {            # ProjectionType values (im2txtprojection.py:18-24): the dataset name inside the file
    "coco_captions", "ms_marco_queries_a", "cc3m_blip_captions", "vg_captions", "vg_dense_captions_test", "online_texts",
}


def filename_components(projection_type="coco_captions", use_talk2dino=True, linear_talk2dino=False, talk2dino_attn_type="qkv",
                        memory_bank_name=None, use_open_clip=False) -> Tuple[str, str, str, str]:
    """(prefix, dataset_name, talk2dino_attn_type_str, postfix) -- ``_build_filename_components`` (:80-170) for the
    configurations this package serves (Talk2DINO / plain CLIP / OpenCLIP banks; the RegionCLIP / INViTE / DenseClip
    prefixes belong to backbones outside the path)."""
    if use_talk2dino:
        prefix, postfix = "", ("-B16" if use_talk2dino is True else use_talk2dino)
        if linear_talk2dino:
            postfix += "-linear"
    else:
        prefix, postfix = "clip-", ""
    attn = "" if talk2dino_attn_type == "qkv" else "_%s" % talk2dino_attn_type
    if isinstance(projection_type, str):
        if projection_type in PROJECTION_TYPES:
            dataset_name = projection_type
        elif os.path.exists(projection_type):
            base = os.path.basename(projection_type).lower()
            dataset_name = ("coco_captions" if "karpathy" in base or "coco" in base else "vg_captions" if "vg" in base
                            else "ms_marco_queries_a" if "marco" in base else "coco_captions")
        else:
            dataset_name = projection_type
    elif memory_bank_name is not None:
        dataset_name = memory_bank_name
    else:
        dataset_name = "coco_captions" if use_talk2dino else "coco"
    if use_open_clip:
        postfix += "-open_clip"
    return prefix, dataset_name, attn, postfix


def memory_bank_filename(projection_type="coco_captions", clip_modelname=None, support_memory_size=591753, use_talk2dino=True,
                         **kw) -> Tuple[str, str]:
    """-> (file name, dataset name).  ``{prefix}{dataset}_text_embeddings{attn}{postfix}-{clip model, / -> .}-{size}.h5`` (:234);
    the default CLIP model is ViT-B/16 with Talk2DINO and ViT-B/32 without (:289-299)."""
    if clip_modelname is None:
        clip_modelname = "ViT-B/16" if use_talk2dino else "ViT-B/32"
    prefix, dataset_name, attn, postfix = filename_components(projection_type, use_talk2dino, **kw)
    return (prefix + "%s_text_embeddings%s%s-%s-%d.h5" % (dataset_name, attn, postfix, clip_modelname.replace("/", "."),
                                                          support_memory_size), dataset_name)


def talk2dino_text_weights(state_dict) -> dict:
    """linear_layer / hidden_layers.0 of a ProjectionLayer checkpoint; ``linear_layer2`` is the old name of the hidden layer
    (talk2dino.py:84-90)."""
    sd = {k: v for k, v in state_dict.items()}
    if "linear_layer2.weight" in sd:
        sd["hidden_layers.0.weight"] = sd.pop("linear_layer2.weight")
        sd["hidden_layers.0.bias"] = sd.pop("linear_layer2.bias")
    if any(k.startswith("hidden_layers.1.") for k in sd):
        raise NotImplementedError("ProjectionLayer with more than one hidden layer")
    return {"w1": sd["linear_layer.weight"].float(), "b1": sd["linear_layer.bias"].float(),
            "w2": sd["hidden_layers.0.weight"].float() if "hidden_layers.0.weight" in sd else None,
            "b2": sd["hidden_layers.0.bias"].float() if "hidden_layers.0.bias" in sd else None}


def build_memory_bank(engine, texts: Sequence[str], encode_text: Callable[[List[str]], torch.Tensor], out_dir: str,
                      projection_type="coco_captions", clip_modelname=None, support_memory_size: Optional[int] = None,
                      talk2dino_state_dict=None, act="tanh", batch_size: int = 1000, **name_kw) -> str:
    """``_build_support_memory`` (:448-560) for a list of captions.  ``talk2dino_state_dict`` None = a plain CLIP bank
    (use_talk2dino False).  Returns the path of the .h5 written into ``out_dir``."""
    texts = list(texts)
    use_t2d = talk2dino_state_dict is not None
    size = len(texts) if support_memory_size is None else int(support_memory_size)
    if len(texts) > size:
        raise ValueError("%d texts do not fit a support memory of %d rows" % (len(texts), size))
    fname, dataset = memory_bank_filename(projection_type, clip_modelname, size, use_talk2dino=use_t2d, **name_kw)
    w = talk2dino_text_weights(talk2dino_state_dict) if use_t2d else None
    rows = []
    for s in range(0, len(texts), batch_size):
        f = encode_text(texts[s:s + batch_size])
        if f.shape[0] != len(texts[s:s + batch_size]):
            raise ValueError("encode_text returned %d rows for %d texts" % (f.shape[0], len(texts[s:s + batch_size])))
        f = f.float()
        if use_t2d:
            f = engine.text_project(f, w["w1"], w["b1"], w["w2"], w["b2"], act=act)
        rows.append(f.cpu())
    feats = torch.cat(rows).numpy() if rows else np.zeros((0, 0), np.float32)
    emb = np.zeros((size, feats.shape[1]), np.float32)
    emb[:len(texts)] = feats
    path = os.path.join(out_dir, fname)
    h5lite.write_bank(path, dataset, emb, texts + [""] * (size - len(texts)))
    return path
