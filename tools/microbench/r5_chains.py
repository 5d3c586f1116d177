"""k greedy-decode chains of N prefixes side by side, nothing else on the chip (diagnostic; run under rocprofv3 --kernel-trace --stats
to see whether the kernels stretch or the gaps between them grow).  usage: r5_chains.py N k [rounds]
PIO_CHAIN_MASK="skip:n,skip:n,...": chain i decodes on a stream confined to CUs [skip, skip + n) (pio_stream_create)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from patchioner_amd import Patchioner, weights as W
from patchioner_amd.pipeline import TraceCaptionPipeline


def main():
    N, k = int(sys.argv[1]), int(sys.argv[2])
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    torch.cuda.set_device(0)
    cfg = {"decap_weights": W.synth_decap(3), "dino_weights": W.synth_dinov2(1), "memory_bank": W.synth_bank(6, 4096).cuda(),
           "prefix_size": 768, "linear_talk2dino": False, "support_memory_size": 4096, "dino_model": "dinov2_vitb14_reg",
           "normalize": True, "resize_dim": 224, "crop_dim": 224, "max_batch": 16, "max_prefixes": 256}
    m = Patchioner.from_config(cfg, device="cuda:0")
    pipe = TraceCaptionPipeline(m, group_batches=8, vit_batches=1, decode_clones=k - 1)
    engines, streams = pipe.decode_engines, pipe.decode_streams
    masks = os.environ.get("PIO_CHAIN_MASK", "")
    raws = []
    if masks:
        import ctypes
        from patchioner_amd._lib import load, check
        streams = []
        for spec in masks.split(",")[:k]:
            skip, n = (int(v) for v in spec.split(":"))
            raw = ctypes.c_void_p()
            check(load().pio_stream_create(0, skip, n, ctypes.byref(raw)))
            raws.append(raw)
            streams.append(torch.cuda.ExternalStream(raw.value, device=torch.device("cuda:0")))
        assert len(streams) == k
    g = torch.Generator(device="cuda").manual_seed(5)
    pres = [torch.nn.functional.normalize(torch.randn(N, 768, device="cuda", generator=g), dim=-1) for _ in range(k)]

    def round_():
        for e, st, p in zip(engines, streams, pres):
            with torch.cuda.stream(st):
                e.decode_greedy(p, steps=30)
    round_(); round_(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(R):
        round_()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / R
    print("N=%d k=%d %s: %.2f ms per round = %.0f captions/s" % (N, k, masks or "unmasked", dt * 1e3, N * k / dt), flush=True)
    pipe.close()


if __name__ == "__main__":
    main()
