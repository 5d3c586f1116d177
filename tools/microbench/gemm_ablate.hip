// Stand-alone timing of k_vit_gemm with parts compiled out (diagnostic build, never shipped):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I patchioner_amd/csrc [-DPIO_ABL_x] tools/microbench/gemm_ablate.hip
#include "../../patchioner_amd/csrc/vit_gemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace pio;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int B = 16, Tp = 264, T = 261, D = 768, M = B * Tp;
  void *A, *A4, *W, *W4, *out16, *q, *k, *v; float *bias, *x, *ls;
  CK(hipMalloc(&A, (size_t)M * D * 2)); CK(hipMalloc(&A4, (size_t)M * 4 * D * 2));
  CK(hipMalloc(&W, (size_t)4 * D * D * 2)); CK(hipMalloc(&W4, (size_t)4 * D * D * 2));
  CK(hipMalloc(&out16, (size_t)M * 4 * D * 2)); CK(hipMalloc(&q, (size_t)B * 12 * 320 * 64 * 2));
  CK(hipMalloc(&k, (size_t)B * 12 * 320 * 64 * 2)); CK(hipMalloc(&v, (size_t)B * 12 * 320 * 64 * 2));
  CK(hipMalloc(&bias, 4 * D * 4)); CK(hipMalloc(&x, (size_t)M * D * 4)); CK(hipMalloc(&ls, D * 4));
  std::vector<uint16_t> h((size_t)M * 4 * D);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3000 + (uint16_t)((i * 2654435761u) >> 22);   // random-ish fp16 in [0.125, 0.25)
  CK(hipMemcpy(A4, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(A, h.data(), (size_t)M * D * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)4 * D * D * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W4, h.data(), (size_t)4 * D * D * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, 4 * D * 4)); CK(hipMemset(x, 0, (size_t)M * D * 4)); CK(hipMemset(ls, 0, D * 4));
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.T = T; g.Tp = Tp; g.Tk = 320; g.G = 5; g.n2 = 256; g.D = D; g.H = 12; g.x = x; g.q = q; g.k = k; g.vT = v; g.bias = bias; g.out16 = out16;
  struct Case { const char* name; GemmEpilogue e; const void* A; int lda, N, K; const void* W; } cases[] = {
    {"qkv  4224x2304x768 ", EPI_QKV, A, D, 3 * D, D, W}, {"proj 4224x768x768  ", EPI_RESIDUAL, A, D, D, D, W},
    {"fc1  4224x3072x768 ", EPI_GELU, A, D, 4 * D, D, W}, {"fc2  4224x768x3072 ", EPI_RESIDUAL, A4, 4 * D, D, 4 * D, W4}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& c : cases) {
    GemmArgs a = g; a.A = c.A; a.lda = c.lda; a.W = c.W; a.M = M; a.N = c.N; a.K = c.K;
    for (int i = 0; i < 3; ++i) CK(launch_vit_gemm(OP_F16, c.e, a, 0));
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int i = 0; i < it; ++i) CK(launch_vit_gemm(OP_F16, c.e, a, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / it, tf = 2.0 * M * c.N * (double)c.K / (us * 1e-6) / 1e12;
    printf("%s %8.1f us  %7.1f TFLOP/s\n", c.name, us, tf);
  }
  return 0;
}
