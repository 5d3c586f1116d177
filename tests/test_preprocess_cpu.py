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
