"""Image transforms (P/src/model.py:347-357): the CPU oracle against Pillow's own outputs (committed
golden vectors and, where Pillow is installed, live), and the host mirror's transforms against the oracle."""
import numpy as np
import pytest

import golden_cases as gc
from oracle import preprocess_oracle as P


def test_oracle_resize_matches_pillow_golden(golden):
    g = golden("preprocess")
    for i, (w, h, nw, nh) in enumerate(gc.PREP_RESIZE_CASES):
        got = P.pil_bicubic_resize(gc.prep_image(i, w, h), nw, nh)
        assert np.array_equal(got, g["resize_%d" % i]), (i, w, h, nw, nh)      # bit-exact (uint8)


def test_oracle_resize_matches_live_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(5)
    for _ in range(12):
        w, h = int(rng.randint(3, 400)), int(rng.randint(3, 400))
        nw, nh = int(rng.randint(2, 300)), int(rng.randint(2, 300))
        arr = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        ref = np.asarray(Image.fromarray(arr).resize((nw, nh), Image.BICUBIC))
        assert np.array_equal(P.pil_bicubic_resize(arr, nw, nh), ref), (w, h, nw, nh)


def test_host_transforms_equal_the_oracle():
    """patchioner_amd.preprocess (PIL on the host, what model.image_transforms returns) == the oracle, including
    torchvision's pad-then-crop when resize_dim < crop_dim and Python's round-half-even crop origin."""
    Image = pytest.importorskip("PIL.Image")
    from patchioner_amd import preprocess as pp
    for i, (w, h, r, c) in enumerate(gc.PREP_TRANSFORM_CASES):
        arr = gc.prep_image(50 + i, w, h)
        crop_t, square_t = pp.make_transforms(r, c)
        assert np.array_equal(crop_t(Image.fromarray(arr)).numpy(), P.image_transforms(arr, r, c)), (w, h, r, c)
        if r <= 256:
            assert np.array_equal(square_t(Image.fromarray(arr)).numpy(), P.image_transforms_no_crop(arr, r)), (w, h, r)


def test_center_crop_origin_rules():
    from patchioner_amd.preprocess import center_crop_origin
    assert center_crop_origin(298, 224, 224) == (37, 0)
    assert center_crop_origin(299, 224, 224) == (38, 0)       # 37.5 -> 38 (half to even)
    assert center_crop_origin(297, 224, 224) == (36, 0)       # 36.5 -> 36
    assert center_crop_origin(200, 150, 224) == (-12, -37)
    assert center_crop_origin(201, 150, 224) == (-11, -37)    # pad 11 left, 12 right


def test_c_abi_pillow_tables_are_bit_exact():
    """The C++ port of Pillow's coefficient builder used by pio_preprocess (api.cpp: pil_axis_table) against the oracle's
    (itself pinned to Pillow), on the host, no GPU: identical int32 tables for up- and down-scaling, partial windows."""
    from patchioner_amd import build as pbuild
    from patchioner_amd._lib import load
    pbuild.build()
    lib = load()
    rng = np.random.RandomState(9)
    cases = [(640, 298), (480, 224), (100, 280), (17, 224), (1023, 224), (224, 224), (37, 6054), (5, 3), (3, 5)]
    cases += [(int(rng.randint(2, 2000)), int(rng.randint(2, 800))) for _ in range(20)]
    for in_size, out_size in cases:
        kk_ref, b_ref = P.precompute_coeffs(in_size, out_size)
        ks = lib.pio_host_pil_ksize(in_size, out_size)
        assert ks == kk_ref.shape[1], (in_size, out_size)
        first = int(rng.randint(0, out_size))
        count = int(rng.randint(0, out_size - first + 1))
        for f, c in ((0, out_size), (first, count)):
            kk = np.zeros((c, ks), dtype=np.int32)
            b = np.zeros((c, 2), dtype=np.int32)
            assert lib.pio_host_pil_table(in_size, out_size, f, c, kk.ctypes.data, b.ctypes.data) == 0
            assert np.array_equal(kk, kk_ref[f:f + c]) and np.array_equal(b, b_ref[f:f + c]), (in_size, out_size, f, c)
    assert lib.pio_host_pil_table(10, 5, 3, 3, None, None) < 0
