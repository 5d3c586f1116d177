"""hipBLASLt (through torch.matmul, fp16, no epilogue) on the ViT GEMM shapes: which kernels it picks and how long they take.
Diagnostic only (run under rocprofv3 --kernel-trace --stats to see the kernel names = tile configuration)."""
import sys
import torch

def main():
    imgs = [int(a) for a in sys.argv[1:]] or [80]
    dev = torch.device("cuda:0")
    for b in imgs:
        M = b * 264          # the engine pads an image's 261 tokens to 264 rows
        for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
            a = torch.randn(M, K, device=dev, dtype=torch.float16)
            w = torch.randn(N, K, device=dev, dtype=torch.float16)
            for _ in range(5):
                (a @ w.t())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                (a @ w.t())
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 50
            print(f"{name:5s} {M}x{N}x{K}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
