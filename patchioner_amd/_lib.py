"""ctypes binding of libpatchioner_hip.so (the C ABI declared in include/patchioner_hip.h).

This is the stub a maintainer of the reference would add: plain pointers and
sizes, device pointers taken from torch tensors with ``.data_ptr()``, the stream from
``torch.cuda.current_stream().cuda_stream``.  There is NO fallback: if the library is missing or a call
fails, a ``PioError`` is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_void_p
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIO_LIB_PATH") or os.path.join(_HERE, "libpatchioner_hip.so")   # PIO_LIB_PATH: A/B builds of the same ABI (tools/microbench)


class PioError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libpatchioner_hip error %d: %s" % (code, msg))
        self.code = code


PIO_ERR_UNKNOWN_WEIGHT = -4


class PioConfig(ctypes.Structure):
    _fields_ = [
        ("embed_dim", c_int32), ("depth", c_int32), ("num_heads", c_int32), ("patch_size", c_int32),
        ("num_registers", c_int32), ("pretrain_grid", c_int32), ("crop_dim", c_int32), ("vit_ln_eps", c_float),
        ("readout_heads", c_int32), ("readout_scale", c_float),
        ("dec_layers", c_int32), ("dec_heads", c_int32), ("dec_embd", c_int32), ("dec_vocab", c_int32),
        ("dec_positions", c_int32), ("prefix_size", c_int32), ("dec_ln_eps", c_float),
        ("max_batch", c_int32), ("max_prefixes", c_int32), ("max_steps", c_int32),
        ("vit_operand_type", c_int32), ("device", c_int32),
        ("vit_arch", c_int32), ("vit_out_dim", c_int32),
    ]


# every symbol include/patchioner_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "pio_last_error": (c_char_p, []),
    "pio_version": (c_char_p, []),
    "pio_stream_create": (c_int32, [c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "pio_stream_destroy": (c_int32, [c_void_p]),
    "pio_create": (c_int32, [POINTER(PioConfig), POINTER(c_void_p)]),
    "pio_destroy": (c_int32, [c_void_p]),
    "pio_clone_decoder": (c_int32, [c_void_p, POINTER(c_void_p)]),
    "pio_load_weight": (c_int32, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int32]),
    "pio_finalize_weights": (c_int32, [c_void_p]),
    "pio_set_memory_bank": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, POINTER(c_int64)]),
    "pio_set_memory_bank_device": (c_int32, [c_void_p, c_void_p, c_int64, c_int32]),
    "pio_vit_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "pio_cls_attention": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "pio_trace_grids": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "pio_bbox_weights": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                   c_void_p, c_int32, c_void_p, c_void_p]),
    "pio_region_reduce": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_float, c_void_p,
                                    c_void_p]),
    "pio_gaussian_map": (c_int32, [c_void_p, c_float, c_void_p, c_void_p]),
    "pio_mem_project": (c_int32, [c_void_p, c_void_p, c_int32, c_float, c_int32, c_void_p, c_int32, c_void_p,
                                  c_void_p]),
    "pio_viecap_prompt_length": (c_int32, [c_void_p]),
    "pio_viecap_set_entities": (c_int32, [c_void_p, c_void_p, c_int32, c_int32]),
    "pio_viecap_mapping": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "pio_viecap_entity_logits": (c_int32, [c_void_p, c_void_p, c_int32, c_float, c_void_p, c_void_p]),
    "pio_lm_score": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "pio_viecap_decode": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "pio_viecap_build_prompt": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "pio_lm_prefill": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "pio_lm_advance": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "pio_beam_select": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "pio_mem_topk": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "pio_text_project": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32,
                                   c_void_p, c_void_p, c_void_p]),
    "pio_revert_transformation": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "pio_decode_greedy": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "pio_profile_enable": (c_int32, [c_void_p, c_int32]),
    "pio_profile_read": (c_int32, [c_void_p, c_int32, POINTER(c_double), POINTER(c_int64), POINTER(c_double),
                                   POINTER(c_double)]),
    "pio_bbox_double_dino": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "pio_ctx_clean": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, c_void_p,
                                c_void_p]),
    "pio_preprocess": (c_int32, [c_void_p, c_void_p, POINTER(c_int64), POINTER(c_int32), c_int32, c_int32, c_int32, c_int32,
                                 c_void_p, c_void_p]),
    "pio_num_tokens": (c_int32, [c_void_p]),
    "pio_grid_side": (c_int32, [c_void_p]),
    "pio_bank_rows": (c_int64, [c_void_p]),
    "pio_host_interpolate_pos_embed": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "pio_host_interpolate_pos_embed_plain": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_double, c_void_p]),
    "pio_host_pil_ksize": (c_int32, [c_int32, c_int32]),
    "pio_host_pil_table": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
}

_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Loads the library (once).  ``import torch`` first so both share torch's bundled HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PioError(-100, "%s not found: run `python patchioner_amd/build.py` (hipcc, gfx950); "
                             "there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads libamdhip64 with the soname our library asks for)
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise PioError(rc, load().pio_last_error().decode("utf-8", "replace"))


def ptr(t) -> Optional[int]:
    """Device/host pointer of a contiguous torch tensor (None stays NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor crossing the C ABI must be contiguous"
    return t.data_ptr()
