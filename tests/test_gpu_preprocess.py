"""pio_preprocess on the GPU against the oracle / Pillow: bit-exact floats."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from patchioner_amd import weights as W

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def engine():
    from patchioner_amd.engine import Engine
    e = Engine(embed_dim=768, depth=1, num_heads=12, num_registers=4, crop_dim=224, max_batch=2, vit_dtype="fp16")
    yield e
    e.close()


def test_device_transforms_are_bit_exact(engine):
    from oracle import preprocess_oracle as P
    cases = gc.PREP_TRANSFORM_CASES
    for r, c in sorted({(r, c) for _, _, r, c in cases}):
        idx = [i for i, k in enumerate(cases) if k[2:] == (r, c)]
        arrs = [gc.prep_image(50 + i, cases[i][0], cases[i][1]) for i in idx]
        got = engine.preprocess(arrs, r, c).cpu().numpy()
        for j, a in enumerate(arrs):
            want = P.image_transforms(a, r, c)
            assert np.array_equal(got[j], want), ("crop", cases[idx[j]], float(np.abs(got[j] - want).max()))
        if r <= 256:
            got2 = engine.preprocess(arrs, r, c, no_crop=True).cpu().numpy()
            for j, a in enumerate(arrs):
                assert np.array_equal(got2[j], P.image_transforms_no_crop(a, r)), ("square", cases[idx[j]])


def test_device_resize_matches_pillow_golden(engine, golden):
    """mode 1 with resize_dim = the golden square cases; non-square golden cases go through mode 0's crop window."""
    from oracle import preprocess_oracle as P
    g = golden("preprocess")
    for i, (w, h, nw, nh) in enumerate(gc.PREP_RESIZE_CASES):
        if nw != nh:
            continue
        got = engine.preprocess([gc.prep_image(i, w, h)], nw, nw, no_crop=True).cpu().numpy()[0]
        assert np.array_equal(got, P.to_tensor_normalized(g["resize_%d" % i])), (w, h, nw, nh)


def test_device_transform_feeds_the_model_like_the_host_transform():
    """Patchioner.preprocess_images == stacking model.image_transforms(PIL) on the host, on a ragged batch, and a batch
    of 16 camera-sized images in one call."""
    Image = pytest.importorskip("PIL.Image")
    from patchioner_amd import Patchioner
    cfg = {"decap_weights": W.synth_decap(3), "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 0,
           "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": 224, "crop_dim": 224,
           "dino_weights": W.synth_dinov2(91, "dinov2_vitb14_reg", depth=1), "memory_bank": None, "max_batch": 16}
    m = Patchioner.from_config(cfg, device="cuda")
    sizes = [(640, 480), (480, 640), (500, 333), (224, 224), (1024, 768), (333, 500), (640, 427), (300, 300),
             (640, 480), (612, 612), (427, 640), (500, 375), (640, 360), (200, 600), (800, 600), (231, 217)]
    pil = [Image.fromarray(gc.prep_image(200 + i, w, h)) for i, (w, h) in enumerate(sizes)]
    want = torch.stack([m.image_transforms(im) for im in pil])
    got = m.preprocess_images(pil)
    assert got.is_cuda and got.shape == (16, 3, 224, 224)
    assert torch.equal(got.cpu(), want)
    want2 = torch.stack([m.image_transforms_no_crop(im) for im in pil[:5]])
    assert torch.equal(m.preprocess_images(pil[:5], no_crop=True).cpu(), want2)
