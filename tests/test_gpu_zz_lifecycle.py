"""Create / use / destroy in ONE interpreter, repeatedly (DESIGN.md "Open observations" of round 1: a host segfault in the
third iteration of such a loop): engines, decoder clones, CU-masked streams, the grouped-decode pipeline with its graphs and
staging rings, all torn down and rebuilt four times, with the same captions every time.  Runs last (file name) so that a
crash here cannot hide another test's result."""
import gc as pygc

import numpy as np
import pytest
import torch

import golden_cases as gc
from patchioner_amd import weights as W

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _model(max_batch=8):
    from patchioner_amd import Patchioner
    cfg = {"decap_weights": W.synth_decap(3), "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 2048,
           "dino_model": "dinov2_vitb14_reg", "normalize": True, "resize_dim": 224, "crop_dim": 224,
           "dino_weights": W.synth_dinov2(91, depth=2), "memory_bank": W.synth_bank(61, 2048), "max_batch": max_batch}
    return Patchioner.from_config(cfg, device="cuda")


def test_create_use_destroy_loop_in_one_process():
    from patchioner_amd.pipeline import TraceCaptionPipeline
    rng = np.random.RandomState(3)
    batches = []
    for i, b in enumerate([4, 8, 3, 8]):
        imgs = W.synth_images(100 + i, b, 224).cuda()
        batches.append((imgs, [gc.block_trace(int(rng.randint(0, 13)), int(rng.randint(0, 13))) for _ in range(b)]))
    first = None
    for it in range(4):
        m = _model()
        want = [m(imgs, get_cls_capt=False, traces=tr)["trace_capts"] for imgs, tr in batches]
        pipe = TraceCaptionPipeline(m, group_batches=2, vit_batches=2, decode_clones=2)
        assert list(pipe.run(batches)) == want
        pipe.close()
        pipe = TraceCaptionPipeline(m, group_batches=4, stage_cus=192, decode_cus=64, decode_clones=1)
        assert list(pipe.run(batches)) == want
        pipe.close()
        assert list(pipe.run(batches[:2])) == want[:2]           # still usable after close()
        pipe.close()
        m.engine.close()
        with pytest.raises(Exception):
            m.engine.vit_forward(batches[0][0])                   # a closed engine fails loudly, it does not crash
        del pipe, m
        pygc.collect()
        torch.cuda.synchronize()
        if first is None:
            first = want
        assert want == first, "iteration %d" % it
