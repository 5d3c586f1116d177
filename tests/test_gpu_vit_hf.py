"""The HIP ViT against the INDEPENDENT HF port of DINOv2-with-registers at full depth (the backbone's
arithmetic is third-party and absent from the reference tree, so this is the strongest pin available for row a2).

tests/golden/vit_hf_depth12.npz holds transformers.Dinov2WithRegistersModel outputs (fp32, CPU, eager attention) for our
seeded 12-block state dict mapped into its layout (tools/oracle/gen_golden.py: gen_vit_hf12): 224^2 and 518^2, plain
weights and the outlier-channel variant (weights.add_outlier_channels: residual-stream channels in the hundreds, a few
MLP hidden units 30x larger), the five global tokens and every 7th / 29th patch token of 2 images.

Tolerances (max |err| / max |ref| over the compared tokens): fp16 operands 4e-3, bf16 3e-2 -- the same bars as the
oracle comparison in test_gpu_parity.py; with the outlier weights the bar is stated against the fixture's own maximum
(the outlier channels dominate it), plus a cosine bar per token that the small channels decide.
"""
import numpy as np
import pytest
import torch

from patchioner_amd import weights as W

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _tokens(sd, size, dtype, batch=2):
    from patchioner_amd.engine import Engine
    e = Engine(embed_dim=768, depth=12, num_heads=12, num_registers=4, crop_dim=size, max_batch=batch, vit_dtype=dtype)
    try:
        e.load_state_dict(sd)
        e.finalize()
        tokens, _ = e.vit_forward(W.synth_images(84, batch, size), want_qkv=False)
        torch.cuda.synchronize()
        return tokens.cpu()
    finally:
        e.close()


@pytest.mark.parametrize("size,stride", [(224, 7), (518, 29)])
@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 3e-2)])
def test_hip_vit_depth12_vs_hf_port(golden, size, stride, dtype, tol):
    g = golden("vit_hf_depth12")
    got = _tokens(W.synth_dinov2(83, depth=12), size, dtype)
    ref = torch.cat([torch.from_numpy(g["plain%d_global" % size]), torch.from_numpy(g["plain%d_patch_sample" % size])], 1)
    cmp = torch.cat([got[:, :5], got[:, 5::stride]], 1)
    assert cmp.shape == ref.shape and torch.isfinite(got).all()
    err = float((cmp - ref).abs().max() / ref.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(cmp, ref, dim=-1).min())
    print("HF depth-12 %d^2 [%s]: rel-max-err %.2e, min cosine %.6f" % (size, dtype, err, cos))
    assert err <= tol and cos >= 1 - 4 * tol * tol


@pytest.mark.parametrize("size,stride", [(224, 7), (518, 29)])
@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 3e-2)])
def test_hip_vit_outlier_channels_vs_hf_port(golden, size, stride, dtype, tol):
    """Residual-stream channels near 470 next to channels of order 0.4, hidden units 30x larger than the rest: the fp32
    residual stream and fp32 LayerNorm statistics keep the fp16 / bf16 operand path finite and inside the same bar."""
    g = golden("vit_hf_depth12")
    sd = W.add_outlier_channels(W.synth_dinov2(83, depth=12))
    got = _tokens(sd, size, dtype)
    ref = torch.cat([torch.from_numpy(g["outlier%d_global" % size]), torch.from_numpy(g["outlier%d_patch_sample" % size])], 1)
    cmp = torch.cat([got[:, :5], got[:, 5::stride]], 1)
    assert torch.isfinite(got).all()
    err = float((cmp - ref).abs().max() / ref.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(cmp, ref, dim=-1).min())
    # the final LayerNorm rescales every token by its (outlier-dominated) deviation: also compare with the outlier
    # channels masked out, relative to the largest remaining reference value
    keep = torch.ones(768, dtype=torch.bool)
    keep[[7, 300, 611]] = False
    err_small = float((cmp[..., keep] - ref[..., keep]).abs().max() / ref[..., keep].abs().max())
    print("HF outliers %d^2 [%s]: rel-max-err %.2e (other channels %.2e), min cosine %.6f" % (size, dtype, err, err_small, cos))
    assert err <= tol and err_small <= 2 * tol and cos >= 1 - 4 * tol * tol


@pytest.mark.parametrize("tag,name,D,heads,depth", [("vitl", "dinov2_vitl14_reg", 1024, 16, 24), ("vits", "dinov2_vits14_reg", 384, 6, 12)])
def test_hip_vit_other_sizes_vs_hf_port(golden, tag, name, D, heads, depth):
    """ViT-L/14-reg at depth 24 (BASELINE config 5's backbone) and ViT-S/14-reg at depth 12, directly against the HF port's
    outputs (tests/golden/vit_hf_variants.npz): row a2's pin no longer stops at ViT-B."""
    from patchioner_amd.engine import Engine
    g = golden("vit_hf_variants")
    e = Engine(embed_dim=D, depth=depth, num_heads=heads, num_registers=4, crop_dim=224, max_batch=2, vit_dtype="fp16",
               readout_heads=16 if D != 384 else 6)
    try:
        e.load_state_dict(W.synth_dinov2(85, name, depth=depth))
        e.finalize()
        got, _ = e.vit_forward(W.synth_images(86, 2, 224), want_qkv=False)
        got = got.cpu()
    finally:
        e.close()
    ref = torch.cat([torch.from_numpy(g["%s_global" % tag]), torch.from_numpy(g["%s_patch_sample" % tag])], 1)
    cmp = torch.cat([got[:, :5], got[:, 5::7]], 1)
    assert cmp.shape == ref.shape and torch.isfinite(got).all()
    err = float((cmp - ref).abs().max() / ref.abs().max())
    print("HF %s depth %d: rel-max-err %.2e" % (tag, depth, err))
    assert err <= 4e-3


@pytest.mark.parametrize("tag,name,patch,stride", [("b16", "vit_base_patch16_clip_224.openai", 16, 5), ("b32", "vit_base_patch32_clip_224.openai", 32, 1)])
@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 3e-2)])
def test_hip_clip_vit_depth12_vs_hf_clip(golden, tag, name, patch, stride, dtype, tol):
    """The OpenAI-CLIP ViT variant (P/src/model.py:358-392, 786-796): patch 16 / 32, norm_pre, QuickGELU, no LayerScale, final
    norm + bias-free 768 -> 512 head on every token -- the HIP path against transformers.CLIPVisionModelWithProjection."""
    from patchioner_amd.engine import Engine
    g = golden("clip_vit_hf")
    n = 224 // patch
    e = Engine(embed_dim=768, depth=12, num_heads=12, num_registers=0, crop_dim=224, patch_size=patch, pretrain_grid=n,
               max_batch=2, vit_dtype=dtype, vit_arch="clip", vit_out_dim=512, vit_ln_eps=1e-5, prefix_size=512)
    try:
        e.load_state_dict(W.synth_clip_vit(87, name, depth=12))
        e.finalize()
        got, _ = e.vit_forward(W.synth_images(88, 2, 224), want_qkv=False)
        got = got.cpu()
    finally:
        e.close()
    assert got.shape == (2, 1 + n * n, 512) and torch.isfinite(got).all()
    ref = torch.cat([torch.from_numpy(g["%s_cls" % tag])[:, None], torch.from_numpy(g["%s_patch_sample" % tag])], 1)
    cmp = torch.cat([got[:, :1], got[:, 1::stride]], 1)
    err = float((cmp - ref).abs().max() / ref.abs().max())
    print("HF CLIP %s [%s]: rel-max-err %.2e" % (tag, dtype, err))
    assert err <= tol


@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 3e-2)])
def test_hip_clip_vit_at_592_vs_hf_clip(golden, dtype, tol):
    """configs/decap_B16_resize.k.yaml (resize_dim = crop_dim = 592; P/src/model.py:371 hands img_size=592 to timm, which resamples the
    14 x 14 position table at load): the HIP path gets the 14 x 14 table, resamples it on the host (pio_finalize_weights) and runs
    T = 1 + 37 x 37 tokens; held to an HF CLIP vision model built natively for 592 x 592 (tools/oracle/gen_golden.py: gen_clip_hf_592)."""
    from patchioner_amd.engine import Engine
    g = golden("clip_vit_hf_592")
    e = Engine(embed_dim=768, depth=12, num_heads=12, num_registers=0, crop_dim=592, patch_size=16, pretrain_grid=14,
               max_batch=1, vit_dtype=dtype, vit_arch="clip", vit_out_dim=512, vit_ln_eps=1e-5, prefix_size=512)
    try:
        e.load_state_dict(W.synth_clip_vit(87, "vit_base_patch16_clip_224.openai", depth=12))
        e.finalize()
        got, _ = e.vit_forward(W.synth_images(89, 1, 592), want_qkv=False)
        got = got.cpu()
    finally:
        e.close()
    assert got.shape == (1, 1 + 37 * 37, 512) and torch.isfinite(got).all()
    ref = torch.cat([torch.from_numpy(g["b16_cls"])[:, None], torch.from_numpy(g["b16_patch_sample"])], 1)
    cmp = torch.cat([got[:, :1], got[:, 1::9]], 1)
    err = float((cmp - ref).abs().max() / ref.abs().max())
    print("HF CLIP b16 at 592 [%s]: rel-max-err %.2e" % (dtype, err))
    assert err <= tol
