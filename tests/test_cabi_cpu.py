"""CPU-side checks of the C-ABI library: it builds, loads, exports every declared symbol, fails loudly
without a GPU, and its host-only load-time code (position-grid interpolation) matches torch."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from patchioner_amd import build
    build.build()
    from patchioner_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from patchioner_amd import _lib
    header = open(os.path.join(ROOT, "include", "patchioner_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:const\s+char\*|int64_t|int)\s+(pio_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pio_version()


def test_clip_position_table_resample_14_to_37_matches_the_fixture(lib):
    """configs/decap_B16_resize.k.yaml (592 x 592 through a patch-16 CLIP ViT): the host resampling of the 14 x 14 table against the
    table the 592 fixture's HF model carried (tools/oracle/gen_golden.py: gen_clip_hf_592 = the torch call timm makes)."""
    from patchioner_amd import weights as W
    g = np.load(os.path.join(ROOT, "tests", "golden", "clip_vit_hf_592.npz"))
    pos = W.synth_clip_vit(87, "vit_base_patch16_clip_224.openai", depth=1)["pos_embed"][0].contiguous()
    out = torch.empty(1 + 37 * 37, 768)
    assert lib.pio_host_interpolate_pos_embed(pos.data_ptr(), 14, 768, 37, out.data_ptr()) == 0
    np.testing.assert_allclose(out[::17, ::3].numpy(), g["pos_sample"], rtol=1e-5, atol=2e-7)
    assert torch.equal(out[0], pos[0])


@pytest.mark.parametrize("n,D,g", [(16, 768, 37), (8, 384, 37), (37, 64, 37), (20, 64, 37), (37, 96, 14), (24, 64, 7)])
def test_pos_embed_interpolation_matches_torch(lib, n, D, g):
    pos = torch.randn(1 + g * g, D, generator=torch.Generator().manual_seed(n))
    out = torch.empty(1 + n * n, D)
    rc = lib.pio_host_interpolate_pos_embed(pos.data_ptr(), g, D, n, out.data_ptr())
    assert rc == 0
    ref = torch.nn.functional.interpolate(pos[1:].reshape(1, g, g, D).permute(0, 3, 1, 2), size=(n, n),
                                          mode="bicubic", antialias=True)
    ref = torch.cat([pos[:1], ref.permute(0, 2, 3, 1).reshape(n * n, D)], 0)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("n,D", [(16, 768), (8, 384), (37, 64), (20, 64), (45, 32)])
def test_pos_embed_interpolation_without_registers_matches_torch(lib, n, D):
    """Hub models WITHOUT registers: bicubic, no antialias, scale_factor (n + 0.1) / 37 (the published
    DinoVisionTransformer.interpolate_pos_encoding with interpolate_offset = 0.1), down- and up-sampling."""
    import ctypes
    g = 37
    pos = torch.randn(1 + g * g, D, generator=torch.Generator().manual_seed(100 + n))
    out = torch.empty(1 + n * n, D)
    assert lib.pio_host_interpolate_pos_embed_plain(pos.data_ptr(), g, D, n, ctypes.c_double(0.1), out.data_ptr()) == 0
    if n == g:
        ref = pos[1:].reshape(g * g, D)
    else:
        ref = torch.nn.functional.interpolate(pos[1:].reshape(1, g, g, D).permute(0, 3, 1, 2), mode="bicubic",
                                              antialias=False, scale_factor=(float(n + 0.1) / g, float(n + 0.1) / g))
        assert ref.shape[-2:] == (n, n)
        ref = ref.permute(0, 2, 3, 1).reshape(n * n, D)
    np.testing.assert_allclose(out.numpy(), torch.cat([pos[:1], ref], 0).numpy(), rtol=1e-5, atol=2e-5)   # N(0, 1) inputs, 16 products per output: FMA contraction differs from ATen


def test_invalid_arguments_fail_loudly(lib):
    from patchioner_amd import _lib
    assert lib.pio_host_interpolate_pos_embed(None, 37, 8, 4, None) == -1
    assert b"bad argument" in lib.pio_last_error()
    with pytest.raises(_lib.PioError):
        _lib.check(lib.pio_create(None, None))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    from patchioner_amd import Patchioner
    from patchioner_amd._lib import PioError
    from patchioner_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU path"):
        Patchioner.from_config({"prefix_size": 768, "support_memory_size": 0, "dino_model": "dinov2_vitb14_reg",
                                "synthetic_seed": 0, "resize_dim": 224, "crop_dim": 224}, device="cpu")
    with pytest.raises(PioError):
        Engine(embed_dim=768, depth=2, num_heads=12, num_registers=4, crop_dim=224)


def test_out_of_scope_options_raise():
    from patchioner_amd import Patchioner
    base = {"prefix_size": 768, "support_memory_size": 0, "synthetic_seed": 0}
    with pytest.raises(NotImplementedError):
        Patchioner.from_config(dict(base, dino_model="dinov2_vitb14_reg", viecap={"meacap": True}), device="cuda")
    with pytest.raises(NotImplementedError):
        Patchioner.from_config(dict(base, dino_model="dinov2_vitb14_reg", clipcap={"x": 1}), device="cuda")
    with pytest.raises(ValueError):
        Patchioner.from_config(dict(base, dino_model="RN50x4"), device="cuda")
    with pytest.raises(ValueError):
        Patchioner.from_config(dict(base, dino_model="dinov2_vitl14_reg_dinotxt"), device="cuda")
    with pytest.raises(AssertionError, match="doesn't match model"):   # timm's tower is built for resize_dim (P/src/model.py:371) and
        Patchioner.from_config(dict(base, dino_model="vit_base_patch16_clip_224.openai", prefix_size=512, resize_dim=336,   # asserts on
                                    crop_dim=224), device="cuda")                                                        # crop_dim inputs
    with pytest.raises(Exception, match="projection_type"):
        Patchioner.from_config({"prefix_size": 768, "support_memory_size": 10, "decap_weights": "x.pt",
                                "dino_model": "dinov2_vitb14_reg", "projection_type": "nonsense"}, device="cuda")


def test_preprocess_and_bbox_adjust():
    from PIL import Image
    from patchioner_amd.preprocess import adjust_bbox_for_transform, make_transforms, process_bboxes
    rng = np.random.RandomState(0)
    img = Image.fromarray(rng.randint(0, 255, size=(300, 400, 3), dtype=np.uint8))
    t, t_nc = make_transforms(224, 224)
    a, b = t(img), t_nc(img)
    assert a.shape == (3, 224, 224) and b.shape == (3, 224, 224) and a.dtype == torch.float32
    box = adjust_bbox_for_transform(img, [100, 50, 120, 80], 224, 224)
    assert 0 <= box[0] < 224 and 0 <= box[1] < 224 and box[0] + box[2] <= 224.0001
    crops = process_bboxes([img], torch.tensor([[[10.0, 10.0, 50.0, 60.0], [0.0, 0.0, 100.0, 100.0]]]), t_nc)
    assert crops.shape == (2, 3, 224, 224)


def test_inline_asm_weight_loads_are_not_touched_before_their_wait(tmp_path):
    """k_lmhead_wide / k_lmhead_f16 hide their weight loads from hipcc in inline asm (decoder.hip); hipcc then knows nothing of
    their latency, so a compiler copy of a destination register between the load and the s_waitcnt that retires it
    would read garbage.  Audit the generated ISA of every instantiation (tools/microbench/asm_load_audit.py)."""
    import re
    import subprocess
    import sys
    asm = tmp_path / "decoder.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "patchioner_amd", "csrc"), "-S", "--cuda-device-only",
                    os.path.join(ROOT, "patchioner_amd", "csrc", "decoder.hip"), "-o", str(asm)], check=True)
    text = asm.read_text()
    names = sorted(set(re.findall(r"^(_ZN3pio\d+k_lmhead_(?:wide|f16)\w+):", text, flags=re.M)))
    assert len(names) == 10, names       # k_lmhead_wide<1,2,4>, k_lmhead_f16<1,2,4,8,16>, k_lmhead_f16_fused<false / true>
    tiled = sorted(set(re.findall(r"^(_ZN3pio\d+k_dec_gemm_b\w+):", text, flags=re.M)))
    assert len(tiled) >= 12, tiled       # k_dec_gemm_b: qkv / fc (3 shapes each), proj (2 x 2 epilogues), fc2 (3)
    names += tiled
    for name in names:
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        one = tmp_path / (name + ".s")
        one.write_text(body)
        assert "scratch_" not in body, name + ": spills next to asm loads"
        if "k_dec_gemm_b" in name:   # the audit reads the listing as straight-line code: true only without branches
            region = body[body.index("global_load_dwordx4"):body.rindex("v_mfma")]
            assert "s_cbranch" not in region, name + ": branch inside the hand-counted region"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "microbench", "asm_load_audit.py"), str(one)],
                             check=True, capture_output=True, text=True).stdout
        assert out.startswith("0 violations"), name + ": " + out[:400]
