"""Import harness for the *reference* Patch-ioner python modules (container-only tooling).

This file never ships to the GPU box's test/bench path: it is used only by
``tools/oracle/gen_golden.py`` (to generate the committed fixtures under
``tests/golden/``) and by the optional container-only cross-check tests that are
skipped when ``/root/reference`` is absent.  It copies no reference source; it only
arranges ``sys.modules`` so that the reference's own files import here:

* the reference's ``src/__init__.py`` imports ``model.py`` which needs ``timm`` and
  ``torchvision`` (absent in this image) -> register an empty package ``refsrc`` whose
  ``__path__`` points at the reference ``src`` directory so sub-modules import directly;
* absent non-arithmetic dependencies (dotenv, ftfy, h5py, timm, torchvision.transforms,
  clip) are stubbed -- none of them performs arithmetic on the hot path;
* ``decoder_config.pkl`` was pickled by transformers 4.x -> rebuilt as a fresh
  ``GPT2Config`` with the same public fields and re-pickled into a temp file.
"""
import os
import pickle
import sys
import tempfile
import types

REF_ROOT = os.environ.get("PIO_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "Patch-ioner", "src")

_loaded = {}


def available() -> bool:
    return os.path.isdir(REF_SRC)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Callable:
    """Stand-in for torchvision transform objects (constructed, never applied)."""

    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k

    def __call__(self, x):
        raise RuntimeError("torchvision transform stub was applied; feed tensors instead")


def load():
    """Returns a namespace with the reference modules (decap, model, bbox_utils, ...)."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_SRC)

    import torch
    import transformers  # noqa: F401  (must precede the torchvision stub)
    import transformers.models.gpt2.modeling_gpt2  # noqa: F401

    if not hasattr(transformers, "AdamW"):
        transformers.AdamW = torch.optim.AdamW

    if "dotenv" not in sys.modules:
        _stub("dotenv", load_dotenv=lambda *a, **k: None)
    if "ftfy" not in sys.modules:
        _stub("ftfy", fix_text=lambda s: s)
    if "h5py" not in sys.modules:
        _stub("h5py")
    if "timm" not in sys.modules:
        _stub("timm")
    if "clip" not in sys.modules:
        _stub("clip")
    try:
        import torchvision.transforms  # noqa: F401
    except Exception:
        tv = _stub("torchvision")
        tv.__path__ = []
        T = _stub(
            "torchvision.transforms",
            Compose=_Callable, Resize=_Callable, CenterCrop=_Callable, ToTensor=_Callable,
            Normalize=_Callable,
            InterpolationMode=types.SimpleNamespace(BICUBIC="bicubic", BILINEAR="bilinear"),
        )
        tv.transforms = T

    pkg = types.ModuleType("refsrc")
    pkg.__path__ = [REF_SRC]
    sys.modules["refsrc"] = pkg

    # modules of out-of-scope backbones that model.py imports at module import time
    _stub("refsrc.dinotxt_utils", get_tokenizer=lambda *a, **k: None)
    px = _stub("refsrc.proxyclip")
    px.__path__ = []
    _stub("refsrc.proxyclip.proxyclip", ProxyCLIP=object)

    import importlib

    decap = importlib.import_module("refsrc.decap.decap")

    # decoder config: re-pickle for the installed transformers
    from transformers import GPT2Config

    with open(os.path.join(REF_SRC, "decap", "decoder_config.pkl"), "rb") as f:
        try:
            old = pickle.load(f)
            fields = {k: v for k, v in old.__dict__.items() if not k.startswith("_")}
        except Exception:
            fields = dict(vocab_size=50257, n_positions=1024, n_embd=768, n_layer=4, n_head=4,
                          activation_function="gelu_new", layer_norm_epsilon=1e-5)
    keep = ("vocab_size", "n_positions", "n_embd", "n_layer", "n_head", "n_inner",
            "activation_function", "resid_pdrop", "embd_pdrop", "attn_pdrop",
            "layer_norm_epsilon", "initializer_range", "scale_attn_weights", "use_cache",
            "bos_token_id", "eos_token_id", "scale_attn_by_inverse_layer_idx",
            "reorder_and_upcast_attn", "tie_word_embeddings")
    cfg = GPT2Config(**{k: fields[k] for k in keep if k in fields})
    tmp = tempfile.NamedTemporaryFile(suffix=".pkl", delete=False)
    pickle.dump(cfg, tmp)
    tmp.close()
    decap.DECAP_DECODER_CONFIG_PATH = tmp.name

    _loaded["decap"] = decap
    _loaded["gpt2_config"] = cfg
    _loaded["bbox_utils"] = importlib.import_module("refsrc.bbox_utils")
    _loaded["dino_extraction"] = importlib.import_module("refsrc.dino_extraction")
    _loaded["embedding_utils"] = importlib.import_module("refsrc.embedding_utils")
    _loaded["im2txt"] = importlib.import_module("refsrc.decap.im2txtprojection.im2txtprojection")
    _loaded["talk2dino"] = importlib.import_module("refsrc.talk2dino.talk2dino")
    _loaded["tokenizer"] = importlib.import_module("refsrc.clip.simple_tokenizer")
    _loaded["model"] = importlib.import_module("refsrc.model")
    return types.SimpleNamespace(**_loaded)


if __name__ == "__main__":
    ns = load()
    print("loaded:", sorted(_loaded))
    print(ns.gpt2_config)
