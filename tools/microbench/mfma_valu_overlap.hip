// Do MFMA and plain VALU / transcendental instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// One workgroup of 8 waves on one CU = 2 waves per SIMD (wave w and w + 4 share SIMD w % 4).  Modes:
//   0: waves 0-3 run 32x32x16 f16 MFMAs (4 independent chains), waves 4-7 idle      -> T_mfma
//   1: waves 0-3 idle, waves 4-7 run v_fma_f32 chains (8 independent)                -> T_valu
//   2: both at once                                                                   -> max(T) if the pipes overlap, sum if not
//   3 / 4: the same with v_exp_f32 instead of v_fma_f32
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap.hip -o tools/microbench/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int valu_kind>
__global__ __launch_bounds__(512) void k(float* out, int iters, int do_mfma) {
  const int wid = threadIdx.x >> 6;
  float sink = 0.f;
  if (wid < 4) {
    if (do_mfma) {
      f32x16 acc[4];
      for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      f16x8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(1.0f + j * 1e-2f); }
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
      }
      for (int c = 0; c < 4; ++c) sink += acc[c][0] + acc[c][7];
    }
  } else if (valu_kind) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-4f + j;
    const float m = 0.999f, ad = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (valu_kind == 1) v[j] = __builtin_fmaf(v[j], m, ad);
          else v[j] = __builtin_amdgcn_exp2f(v[j] * 1e-3f);
        }
    }
    for (int j = 0; j < 8; ++j) sink += v[j];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <int valu_kind>
static float run(float* out, int iters, int do_mfma) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<valu_kind>, dim3(1), dim3(512), 0, 0, out, iters, do_mfma);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<valu_kind>, dim3(1), dim3(512), 0, 0, out, iters, do_mfma);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  const int iters = 200000;     // per iteration: 4 MFMAs (128 cycles of matrix pipe) / 32 VALU instructions
  const float tm = run<0>(out, iters, 1);
  const float tf = run<1>(out, iters, 0), tmf = run<1>(out, iters, 1);
  const float te = run<2>(out, iters, 0), tme = run<2>(out, iters, 1);
  printf("MFMA alone %.2f ms (%.1f ns per MFMA)\n", tm, tm * 1e6 / (iters * 4.0));
  printf("v_fma_f32 alone %.2f ms (%.2f ns per instruction)   together %.2f ms   (sum %.2f, max %.2f)\n", tf, tf * 1e6 / (iters * 32.0), tmf, tm + tf, tm > tf ? tm : tf);
  printf("v_exp_f32 alone %.2f ms (%.2f ns per instruction)   together %.2f ms   (sum %.2f, max %.2f)\n", te, te * 1e6 / (iters * 32.0), tme, tm + te, tm > te ? tm : te);
  return 0;
}
