"""Thin python owner of one libpatchioner_hip handle: torch tensors in, torch tensors out.

PyTorch here only provides device memory (tensors), the current stream and (in ``dist.py``)
``torch.distributed``; every computation is a C-ABI call into the HIP library.  One Engine per process /
GPU; not re-entrant (same as the reference, whose hooks write module-level globals,
P/src/dino_extraction.py:7, P/src/model.py:30).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import PioConfig, PioError, check, ptr


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class Engine:
    def __init__(self, *, embed_dim: int, depth: int, num_heads: int, num_registers: int, crop_dim: int,
                 patch_size: int = 14, pretrain_grid: int = 37, prefix_size: int = 768, dec_layers: int = 4,
                 dec_heads: int = 4, dec_embd: int = 768, dec_vocab: int = 50257, dec_positions: int = 1024,
                 max_batch: int = 16, max_prefixes: int = 64, max_steps: int = 30, vit_dtype: str = "fp16",
                 device_index: int = 0, readout_heads: int = 16, readout_scale: float = 0.125, vit_arch: str = "dinov2",
                 vit_out_dim: int = 0, vit_ln_eps: float = 1e-6):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise PioError(-101, "no HIP device visible: the captioning path has no CPU fallback")
        self.device = torch.device("cuda", device_index)
        cfg = PioConfig(
            embed_dim=embed_dim, depth=depth, num_heads=num_heads, patch_size=patch_size,
            num_registers=num_registers, pretrain_grid=pretrain_grid, crop_dim=crop_dim, vit_ln_eps=vit_ln_eps,
            readout_heads=readout_heads, readout_scale=readout_scale, dec_layers=dec_layers, dec_heads=dec_heads,
            dec_embd=dec_embd, dec_vocab=dec_vocab, dec_positions=dec_positions, prefix_size=prefix_size,
            dec_ln_eps=1e-5, max_batch=max_batch, max_prefixes=max_prefixes, max_steps=max_steps,
            vit_operand_type={"fp16": 0, "bf16": 1, "fp32": 2}[vit_dtype], device=device_index,
            vit_arch={"dinov2": 0, "clip": 1}[vit_arch], vit_out_dim=vit_out_dim)
        self.cfg = cfg
        h = ctypes.c_void_p()
        check(self.lib.pio_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h
        # D = width of the tokens the backbone returns (the CLIP ViT's head maps embed_dim -> vit_out_dim); Dv = its own width
        self.D, self.G, self.Dv = (vit_out_dim or embed_dim), 1 + num_registers, embed_dim
        self.vit_arch = vit_arch
        self.n = self.lib.pio_grid_side(h)
        self.T = self.lib.pio_num_tokens(h)
        self.n2 = self.n * self.n
        self.prefix_size = prefix_size
        self.max_batch, self.max_prefixes, self.max_steps = max_batch, max_prefixes, max_steps
        self.num_weights = 0
        self.bank_rows = 0
        self.bank_dim = 0
        self._finalized = False
        self._stage_ring = [dict(host=None, dev=None, event=None) for _ in range(8)]
        self._stage_next = 0

    def close(self):
        """Destroys the handle.  An engine that still has live decoder clones is refused by the library (the clones borrow
        its weights): the handle is kept and the error raised, so that nothing leaks silently."""
        if getattr(self, "h", None) is not None and self.h.value:
            check(self.lib.pio_destroy(self.h))
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone_decoder(self) -> "DecoderClone":
        """A second greedy decoder on this engine's weights (pio_clone_decoder): own KV caches / scratch / graphs, so its
        decode_greedy may run concurrently with this engine's on another stream.  Close it before this engine."""
        return DecoderClone(self)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False) -> List[str]:
        """Upload fp32 tensors under their checkpoint keys; returns the keys the library does not know."""
        unknown = []
        for k, v in sd.items():
            t = v.detach().to(dtype=torch.float32, device="cpu").contiguous()
            shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            rc = self.lib.pio_load_weight(self.h, k.encode(), ptr(t), shape, t.dim())
            if rc == _lib.PIO_ERR_UNKNOWN_WEIGHT:
                unknown.append(k)
                continue
            check(rc)
            self.num_weights += t.numel()
        if strict and unknown:
            raise PioError(_lib.PIO_ERR_UNKNOWN_WEIGHT, "unexpected keys: %s" % unknown[:8])
        return unknown

    def finalize(self):
        check(self.lib.pio_finalize_weights(self.h))
        self._finalized = True

    def set_memory_bank(self, bank: torch.Tensor) -> int:
        bank = bank.detach().to(torch.float32).contiguous()
        if bank.is_cuda:
            # the reference drops zero-norm rows when it loads the bank (im2txtprojection.py:343-345); a zero row would
            # give inv_norm = inf and NaN similarities
            keep = bank.norm(dim=-1) != 0
            if not bool(keep.all()):
                bank = bank[keep].contiguous()
            check(self.lib.pio_set_memory_bank_device(self.h, ptr(bank), bank.shape[0], bank.shape[1]))
        else:
            kept = ctypes.c_int64(0)
            check(self.lib.pio_set_memory_bank(self.h, ptr(bank), bank.shape[0], bank.shape[1], ctypes.byref(kept)))
        self.bank_rows = int(self.lib.pio_bank_rows(self.h))
        self.bank_dim = int(bank.shape[1])
        return self.bank_rows

    # ------------------------------------------------------------------ measurement
    PROFILE_CLASSES = {"vit_gemm": 0, "vit_attention": 1, "vit_layernorm": 2, "mem_project": 3, "decode": 4}

    def profile_enable(self, on: bool = True):
        check(self.lib.pio_profile_enable(self.h, 1 if on else 0))

    def profile_read(self) -> Dict[str, dict]:
        out = {}
        for name, cls in self.PROFILE_CLASSES.items():
            ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            check(self.lib.pio_profile_read(self.h, cls, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl),
                                            ctypes.byref(by)))
            out[name] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
        return out

    # ------------------------------------------------------------------ a2/a3 backbone
    def _dev(self, t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        return t.to(device=self.device, dtype=dtype).contiguous()

    def vit_forward(self, imgs: torch.Tensor, want_qkv: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        imgs = self._dev(imgs)
        B = imgs.shape[0]
        crop = self.cfg.crop_dim
        if tuple(imgs.shape[1:]) != (3, crop, crop):
            # the reference's reshape in process_self_attention fails for any other size
            raise ValueError("images must be [B,3,%d,%d], got %s" % (crop, crop, tuple(imgs.shape)))
        tokens = torch.empty(B, self.T, self.D, device=self.device, dtype=torch.float32)
        if want_qkv and self.vit_arch != "dinov2":
            raise PioError(_lib.PIO_ERR_INVALID_ARG, "the CLIP ViT exposes no qkv hook (P/src/model.py:864-865)")
        qkv = torch.empty(B, self.T, 3 * self.Dv, device=self.device, dtype=torch.float32) if want_qkv else None
        for s in range(0, B, self.max_batch):
            e = min(B, s + self.max_batch)
            check(self.lib.pio_vit_forward(self.h, ptr(imgs[s:e]), e - s, ptr(tokens[s:e]),
                                           ptr(qkv[s:e]) if want_qkv else None, _stream()))
        return tokens, qkv

    # ------------------------------------------------------------------ a4/a5 read-out
    def cls_attention(self, qkv: torch.Tensor, tokens: torch.Tensor, want_maps=False, want_avg=False,
                      want_disentangled=False):
        B = qkv.shape[0]
        Hr = self.cfg.readout_heads
        self_attn = torch.empty(B, self.n2, device=self.device, dtype=torch.float32)
        maps = torch.empty(B, Hr, self.n2, device=self.device, dtype=torch.float32) if want_maps else None
        avg = torch.empty(B, self.D, device=self.device, dtype=torch.float32) if want_avg else None
        dis = torch.empty(B, Hr, self.D, device=self.device, dtype=torch.float32) if want_disentangled else None
        for s in range(0, B, self.max_batch):
            e = min(B, s + self.max_batch)
            sl = slice(s, e)
            check(self.lib.pio_cls_attention(
                self.h, ptr(qkv[sl]), ptr(tokens[sl]), e - s, ptr(self_attn[sl]),
                ptr(maps[sl]) if want_maps else None, ptr(avg[sl]) if want_avg else None,
                ptr(dis[sl]) if want_disentangled else None, _stream()))
        return self_attn, maps, avg, dis

    # ------------------------------------------------------------------ a6 traces
    def trace_grids(self, traces: Sequence[Sequence[dict]]) -> torch.Tensor:
        """map_traces_to_grid for a batch: points packed CSR-style into ONE pinned staging buffer (x, y as
        float64 followed by the int32 offsets) and sent with one asynchronous copy; the counting runs on
        the device.  A ring of 8 staging slots (each guarded by an event) lets the host run several batches
        ahead of the GPU without overwriting a buffer whose copy has not executed yet."""
        B = len(traces)
        xy = np.array([(p["x"], p["y"]) for tr in traces for p in tr], dtype=np.float64).reshape(-1, 2)
        npts = xy.shape[0]
        nbytes = npts * 16 + (B + 1) * 4
        slot = self._stage_ring[self._stage_next % len(self._stage_ring)]
        self._stage_next += 1
        if slot["event"] is not None:
            slot["event"].synchronize()
        if slot["host"] is None or slot["host"].numel() < nbytes:
            cap = max(1 << 16, 2 * nbytes)
            slot["host"] = torch.empty(cap, dtype=torch.uint8).pin_memory()
            slot["dev"] = torch.empty(cap, dtype=torch.uint8, device=self.device)
        host = slot["host"].numpy()
        host[: npts * 16].view(np.float64)[:] = xy.reshape(-1)
        offs = host[npts * 16: nbytes].view(np.int32)
        offs[0] = 0
        np.cumsum([len(tr) for tr in traces], out=offs[1:])
        slot["dev"][:nbytes].copy_(slot["host"][:nbytes], non_blocking=True)
        base = slot["dev"].data_ptr()
        grids = torch.empty(B, self.n, self.n, device=self.device, dtype=torch.float32)
        check(self.lib.pio_trace_grids(self.h, base if npts else None, base + npts * 16, B, npts, ptr(grids), _stream()))
        if slot["event"] is None:
            slot["event"] = torch.cuda.Event()
        slot["event"].record(torch.cuda.current_stream())
        return grids

    # ------------------------------------------------------------------ a7 boxes
    def bbox_weights(self, boxes_i32: torch.Tensor, mode: int, variance: float = 0.5,
                     center_choice: Optional[torch.Tensor] = None, attn: Optional[torch.Tensor] = None,
                     single_map: bool = False):
        """boxes_i32 [B,NB,4] int32 (already // patch_size); returns (weights [B,NB,n2], single [B,n2] | None)."""
        boxes_i32 = self._dev(boxes_i32, torch.int32)
        B, NB = boxes_i32.shape[:2]
        weights = torch.empty(B, NB, self.n2, device=self.device, dtype=torch.float32)
        single = torch.empty(B, self.n2, device=self.device, dtype=torch.float32) if single_map else None
        cc = self._dev(center_choice, torch.int32) if center_choice is not None else None
        check(self.lib.pio_bbox_weights(self.h, ptr(boxes_i32), B, NB, mode, float(variance), ptr(cc), ptr(attn),
                                        ptr(weights), 1 if single_map else 0, ptr(single), _stream()))
        return weights, single

    def preprocess(self, images: Sequence, resize_dim: int, crop_dim: int, no_crop: bool = False) -> torch.Tensor:
        """image_transforms / image_transforms_no_crop (P/src/model.py:347-357) for a list of RGB images of different
        sizes (PIL images or uint8 [H][W][3] arrays / tensors) -> fp32 [B][3][S][S] on the device, bit-exact to the
        host PIL pipeline.  The raw pixels cross PCIe once (pinned staging, one asynchronous copy); resize, crop and
        normalisation run on the GPU (pio_preprocess)."""
        arrs = []
        for im in images:
            if isinstance(im, torch.Tensor):
                a = im.detach().cpu().numpy()
            elif isinstance(im, np.ndarray):
                a = im
            else:
                a = np.asarray(im.convert("RGB"))
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("preprocess expects RGB uint8 [H][W][3] images")
            arrs.append(np.ascontiguousarray(a))
        B = len(arrs)
        sizes = np.array([(a.shape[1], a.shape[0]) for a in arrs], dtype=np.int32)
        nbytes = [a.size for a in arrs]
        offsets = np.zeros(B, dtype=np.int64)
        offsets[1:] = np.cumsum([(n + 15) // 16 * 16 for n in nbytes])[:-1]
        total = int(offsets[-1] + nbytes[-1])
        slot = self._stage_ring[self._stage_next % len(self._stage_ring)]
        self._stage_next += 1
        if slot["event"] is not None:
            slot["event"].synchronize()
        if slot["host"] is None or slot["host"].numel() < total:
            cap = max(1 << 16, 2 * total)
            slot["host"] = torch.empty(cap, dtype=torch.uint8).pin_memory()
            slot["dev"] = torch.empty(cap, dtype=torch.uint8, device=self.device)
        host = slot["host"].numpy()
        for a, o, n in zip(arrs, offsets, nbytes):
            host[o:o + n] = a.reshape(-1)
        slot["dev"][:total].copy_(slot["host"][:total], non_blocking=True)
        S = resize_dim if no_crop else crop_dim
        out = torch.empty(B, 3, S, S, device=self.device, dtype=torch.float32)
        check(self.lib.pio_preprocess(self.h, slot["dev"].data_ptr(), offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                      sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), B, int(resize_dim), int(crop_dim),
                                      1 if no_crop else 0, ptr(out), _stream()))
        if slot["event"] is None:
            slot["event"] = torch.cuda.Event()
        slot["event"].record(torch.cuda.current_stream())
        return out

    def region_reduce(self, tokens: torch.Tensor, weights: torch.Tensor, img_index: Optional[torch.Tensor],
                      scale: float) -> torch.Tensor:
        """out[r] = scale * sum_p weights[r,p] * patch_tokens[img_index[r], p]"""
        weights = self._dev(weights.reshape(-1, self.n2))
        R = weights.shape[0]
        idx = self._dev(img_index, torch.int32) if img_index is not None else None
        out = torch.empty(R, self.D, device=self.device, dtype=torch.float32)
        check(self.lib.pio_region_reduce(self.h, ptr(tokens), tokens.shape[0], ptr(weights), ptr(idx), R, float(scale),
                                         ptr(out), _stream()))
        return out

    def bbox_double_dino(self, tokens: torch.Tensor, slices: torch.Tensor, use_cls: bool, return_type: str) -> torch.Tensor:
        """extract_bboxes_feats_double_dino, return types "cls" / "avg" (P/src/bbox_utils.py:300-403): the last ViT block
        re-run on [cls | registers | region patches] of every box.  ``slices`` int32 [B, NB, 4] = python-normalised
        (y_start, y_end, x_start, x_end) patch-grid slices.  -> [B, NB, D]."""
        B, NB = slices.shape[:2]
        sl = self._dev(slices.reshape(-1, 4), torch.int32)
        out = torch.empty(B * NB, self.D, device=self.device, dtype=torch.float32)
        check(self.lib.pio_bbox_double_dino(self.h, ptr(tokens), ptr(sl), B, NB, 1 if use_cls else 0,
                                            {"cls": 0, "avg": 1}[return_type], ptr(out), _stream()))
        return out.view(B, NB, self.D)

    def ctx_clean(self, dirty: torch.Tensor, ctx: torch.Tensor, cleaning_type: str, alpha: float,
                  normalize_inputs: bool = False) -> torch.Tensor:
        """Patchioner.ctx_cleaner (P/src/model.py:1425-1436): dirty [B, S, D], ctx [B, D] -> cleaned [B, S, D];
        ``normalize_inputs`` L2-normalises both first (model.py:907-913)."""
        dirty, ctx = self._dev(dirty), self._dev(ctx)
        B, S, D = dirty.shape
        out = torch.empty_like(dirty)
        check(self.lib.pio_ctx_clean(self.h, ptr(dirty), ptr(ctx), B * S, D, S,
                                     {"orthogonal_projection": 0, "contrastive_mask": 1}[cleaning_type], float(alpha),
                                     1 if normalize_inputs else 0, ptr(out), _stream()))
        return out

    def project_many(self, q: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """project() for any number of queries [N, D] (in-place L2 normalisation of q, like project): chunks of
        max_prefixes rows, one pass over the bank per 16 of them."""
        assert q.is_cuda and q.dtype == torch.float32 and q.is_contiguous()
        out = torch.empty_like(q)
        for s in range(0, q.shape[0], self.max_prefixes):
            out[s:s + self.max_prefixes] = self.project(q[s:s + self.max_prefixes], normalize=normalize)
        return out

    def gaussian_map(self, variance: float) -> torch.Tensor:
        m = torch.empty(self.n2, device=self.device, dtype=torch.float32)
        check(self.lib.pio_gaussian_map(self.h, float(variance), ptr(m), _stream()))
        return m

    # ------------------------------------------------------------------ a9/a10 projection
    def project(self, q: torch.Tensor, temperature: float = 0.01, normalize: bool = False,
                n_best: Optional[int] = None):
        """q [N,D] CUDA fp32 contiguous is L2-normalised IN PLACE (reference quirk)."""
        assert q.is_cuda and q.dtype == torch.float32 and q.is_contiguous()
        if q.shape[1] != self.bank_dim:        # the kernel strides q / out by the bank's width
            raise PioError(_lib.PIO_ERR_SHAPE, "query width %d does not match the %d-wide memory bank" % (q.shape[1], self.bank_dim))
        N = q.shape[0]
        out = torch.empty(N, q.shape[1], device=self.device, dtype=torch.float32)
        nb = int(n_best) if n_best else 0
        best = torch.empty(N, nb, device=self.device, dtype=torch.float32) if nb else None
        check(self.lib.pio_mem_project(self.h, ptr(q), N, float(temperature), 1 if normalize else 0, ptr(out), nb,
                                       ptr(best), _stream()))
        return (out, best) if nb else out

    def topk_rows(self, q: torch.Tensor, k: int = 1):
        """project(..., return_argmax_text=True, return_n_best_sims=k) without the texts: q [N, D] is L2-normalised IN PLACE,
        returns (sims [N, k] descending, rows [N, k] int64) of the k most similar bank rows per query."""
        assert q.is_cuda and q.dtype == torch.float32 and q.is_contiguous()
        if q.shape[1] != self.bank_dim:
            raise PioError(_lib.PIO_ERR_SHAPE, "query width %d does not match the %d-wide memory bank" % (q.shape[1], self.bank_dim))
        N = q.shape[0]
        sims = torch.empty(N, k, device=self.device, dtype=torch.float32)
        rows = torch.empty(N, k, device=self.device, dtype=torch.int64)
        for s in range(0, N, self.max_prefixes):
            e = min(N, s + self.max_prefixes)
            check(self.lib.pio_mem_topk(self.h, ptr(q[s:e]), e - s, int(k), ptr(sims[s:e]), ptr(rows[s:e]), _stream()))
        return sims, rows

    ACTS = {None: 0, "none": 0, "relu": 1, "tanh": 2, "sigmoid": 3}

    def text_project(self, x: torch.Tensor, w1, b1, w2=None, b2=None, act="tanh") -> torch.Tensor:
        """ProjectionLayer.project_clip_txt (P/src/talk2dino/talk2dino.py:73-83): x [N, in] CLIP text features ->
        hidden_layer(act(linear_layer(x))) [N, out] (linear_layer alone without w2), exact fp32 on the device."""
        x, w1, b1 = self._dev(x), self._dev(w1), self._dev(b1)
        N, out_dim = x.shape[0], w1.shape[0]
        if w1.shape[1] != x.shape[1] or b1.shape != (out_dim,):
            raise PioError(_lib.PIO_ERR_SHAPE, "linear_layer %s does not take %d-wide features" % (tuple(w1.shape), x.shape[1]))
        out = torch.empty(N, out_dim, device=self.device, dtype=torch.float32)
        hid = None
        if w2 is not None:
            w2, b2 = self._dev(w2), self._dev(b2)
            if w2.shape != (out_dim, out_dim) or b2.shape != (out_dim,):
                raise PioError(_lib.PIO_ERR_SHAPE, "hidden layer %s is not [%d, %d]" % (tuple(w2.shape), out_dim, out_dim))
            hid = torch.empty_like(out)
        check(self.lib.pio_text_project(self.h, ptr(x), N, x.shape[1], ptr(w1), ptr(b1), out_dim, ptr(w2) if w2 is not None else None,
                                        ptr(b2) if w2 is not None else None, self.ACTS[act], ptr(hid) if hid is not None else None,
                                        ptr(out), _stream()))
        return out

    def revert_transformation(self, x: torch.Tensor) -> torch.Tensor:
        x = self._dev(x)
        out = torch.empty(x.shape[0], self.prefix_size, device=self.device, dtype=torch.float32)
        check(self.lib.pio_revert_transformation(self.h, ptr(x), x.shape[0], ptr(out), _stream()))
        return out

    # ------------------------------------------------------------------ f1 ViECap head
    def viecap_set_entities(self, emb: torch.Tensor) -> None:
        t = emb.detach().to(dtype=torch.float32, device="cpu").contiguous()
        check(self.lib.pio_viecap_set_entities(self.h, ptr(t), t.shape[0], t.shape[1]))
        self.viecap_entities = int(t.shape[0])

    def viecap_mapping(self, feats: torch.Tensor) -> torch.Tensor:
        """feats [N, C] CUDA fp32 contiguous, L2-normalised IN PLACE -> soft prompt [N, Lc, E]."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous()
        N = feats.shape[0]
        Lc = self.lib.pio_viecap_prompt_length(self.h)
        out = torch.empty(N, Lc, self.cfg.dec_embd, device=self.device, dtype=torch.float32)
        for s in range(0, N, self.max_prefixes):
            e = min(N, s + self.max_prefixes)
            check(self.lib.pio_viecap_mapping(self.h, ptr(feats[s:e]), e - s, ptr(out[s:e]), _stream()))
        return out

    def viecap_entity_logits(self, feats: torch.Tensor, temperature: float) -> torch.Tensor:
        N = feats.shape[0]
        out = torch.empty(N, self.viecap_entities, device=self.device, dtype=torch.float32)
        check(self.lib.pio_viecap_entity_logits(self.h, ptr(feats), N, float(temperature), ptr(out), _stream()))
        return out

    def lm_score(self, rows) -> torch.Tensor:
        """Teacher-forced negative log-likelihood sums of token rows (list of lists of ids) under the language model
        (pio_lm_score): [N] float32 on the device; the reference's loss is this / (len - 1)."""
        N = len(rows)
        out = torch.empty(N, device=self.device, dtype=torch.float32)
        cap = min(self.max_prefixes, 64)
        for s in range(0, N, cap):
            chunk = rows[s:s + cap]
            L = max(1, max(len(r) for r in chunk))
            tok = torch.zeros(len(chunk), L, dtype=torch.int32)
            for i, r in enumerate(chunk):
                tok[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
            lens = torch.tensor([len(r) for r in chunk], dtype=torch.int32)
            tok_d, lens_d = self._dev(tok, torch.int32), self._dev(lens, torch.int32)
            check(self.lib.pio_lm_score(self.h, ptr(tok_d), ptr(lens_d), len(chunk), L, ptr(out[s:s + cap]), _stream()))
        return out

    def viecap_decode(self, cont: Optional[torch.Tensor], tokens: Optional[torch.Tensor], soft_first: bool = True, steps: int = 64) -> torch.Tensor:
        """cont [N, Lc, E] device (None: only_hard_prompt, the prompt is the tokens' word embeddings alone), tokens [N, Lt] int32
        (host or device) or None -> greedy ids [N, steps] int32."""
        tok = self._dev(tokens, torch.int32) if tokens is not None else None
        if cont is None and tok is None:
            raise ValueError("viecap_decode: neither a soft prompt nor prompt tokens")
        N = cont.shape[0] if cont is not None else tok.shape[0]
        Lt = int(tok.shape[1]) if tok is not None else 0
        ids = torch.empty(N, steps, device=self.device, dtype=torch.int32)
        for s in range(0, N, self.max_prefixes):
            e = min(N, s + self.max_prefixes)
            check(self.lib.pio_viecap_decode(self.h, ptr(cont[s:e].contiguous()) if cont is not None else None,
                                             ptr(tok[s:e].contiguous()) if tok is not None else None,
                                             e - s, Lt, 1 if soft_first else 0, int(steps), ptr(ids[s:e]), _stream()))
        return ids

    def viecap_build_prompt(self, cont: Optional[torch.Tensor], tokens: Optional[torch.Tensor], soft_first: bool = True) -> torch.Tensor:
        """The prompt embeddings [N, Lc + Lt, E] as viecap_decode assembles them (word_embed + torch.cat, entrypoint.py:126-135)."""
        tok = self._dev(tokens, torch.int32) if tokens is not None else None
        if cont is None and tok is None:
            raise ValueError("viecap_build_prompt: neither a soft prompt nor prompt tokens")
        N = cont.shape[0] if cont is not None else tok.shape[0]
        Lt = int(tok.shape[1]) if tok is not None else 0
        Lc = int(cont.shape[1]) if cont is not None else 0
        out = torch.empty(N, Lc + Lt, self.cfg.dec_embd, device=self.device, dtype=torch.float32)
        check(self.lib.pio_viecap_build_prompt(self.h, ptr(cont.contiguous()) if cont is not None else None,
                                               ptr(tok) if tok is not None else None, N, Lt, 1 if soft_first else 0, ptr(out), _stream()))
        return out

    # ---- beam search building blocks (search.py:193-285): the W beams of ONE image are the rows of these calls
    def lm_prefill(self, embeds: torch.Tensor) -> torch.Tensor:
        """embeds [W, P, E] (the prompt, once per beam) -> log(softmax(next-token logits)) [W, vocab]."""
        embeds = self._dev(embeds).contiguous()
        W, P = int(embeds.shape[0]), int(embeds.shape[1])
        logp = torch.empty(W, self.cfg.dec_vocab, device=self.device, dtype=torch.float32)
        check(self.lib.pio_lm_prefill(self.h, ptr(embeds), W, P, ptr(logp), _stream()))
        return logp

    def lm_advance(self, tokens: torch.Tensor, src_rows: Optional[torch.Tensor], pos: int) -> torch.Tensor:
        """Row n continues the sequence of row src_rows[n] (None: as they are) with tokens[n] at position ``pos`` -> log-probabilities."""
        tok = self._dev(tokens, torch.int32).contiguous()
        src = self._dev(src_rows, torch.int32).contiguous() if src_rows is not None else None
        W = int(tok.shape[0])
        logp = torch.empty(W, self.cfg.dec_vocab, device=self.device, dtype=torch.float32)
        check(self.lib.pio_lm_advance(self.h, ptr(tok), ptr(src) if src is not None else None, W, int(pos), ptr(logp), _stream()))
        return logp

    def beam_select(self, logp: torch.Tensor, scores=None, lens=None, stopped=None):
        """One selection of beam_search (see pio_beam_select) -> (values [W] fp32, flat indices [W] int64), on the host."""
        W = int(logp.shape[0])
        val = torch.empty(W, device=self.device, dtype=torch.float32)
        idx = torch.empty(W, device=self.device, dtype=torch.int64)
        if scores is None:
            check(self.lib.pio_beam_select(self.h, ptr(logp), None, None, None, W, ptr(val), ptr(idx), _stream()))
        else:
            sc, ln = self._dev(scores, torch.float32).contiguous(), self._dev(lens, torch.float32).contiguous()
            st = self._dev(stopped, torch.int32).contiguous()
            check(self.lib.pio_beam_select(self.h, ptr(logp), ptr(sc), ptr(ln), ptr(st), W, ptr(val), ptr(idx), _stream()))
        return val.cpu(), idx.cpu()

    # ------------------------------------------------------------------ a11/a12 decoder
    def decode_greedy(self, prefix: torch.Tensor, steps: int = 30, want_logprob: bool = False):
        prefix = self._dev(prefix)
        N = prefix.shape[0]
        ids = torch.empty(N, steps, device=self.device, dtype=torch.int32)
        lp = torch.empty(N, steps, device=self.device, dtype=torch.float32) if want_logprob else None
        chunk = min(self.max_prefixes, 64) if want_logprob else self.max_prefixes     # the exact (log-prob) head: <= 64 rows
        for s in range(0, N, chunk):
            e = min(N, s + chunk)
            check(self.lib.pio_decode_greedy(self.h, ptr(prefix[s:e]), e - s, steps, ptr(ids[s:e]),
                                             ptr(lp[s:e]) if want_logprob else None, _stream()))
        return ids, lp



class DecoderClone:
    """Handle from pio_clone_decoder: borrows the parent's decoder weights, owns its workspaces.  Only decode_greedy."""

    def __init__(self, parent: Engine):
        self.lib, self.parent, self.device = parent.lib, parent, parent.device
        self.max_prefixes, self.max_steps, self.prefix_size = parent.max_prefixes, parent.max_steps, parent.prefix_size
        h = ctypes.c_void_p()
        check(self.lib.pio_clone_decoder(parent.h, ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.pio_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    _dev = Engine._dev
    decode_greedy = Engine.decode_greedy
