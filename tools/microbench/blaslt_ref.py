"""Library reference point for the four ViT GEMM shapes (hipBLASLt through torch.matmul, fp16, no epilogue)."""
import sys, torch, time
for M in ([int(a) for a in sys.argv[1:]] or [16 * 264]):
  for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
      a = torch.randn(M, K, device="cuda", dtype=torch.float16) * 0.1
      w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.1
      for _ in range(5): torch.matmul(a, w.t())
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(50): torch.matmul(a, w.t())
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) * 1e3 / 50
      print("%-5s %dx%dx%d  %7.1f us  %7.1f TFLOP/s" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
