// How fast do v_mfma_f32_16x16x4_f32 issue?  One workgroup on one CU, 1 or 2 waves per SIMD, each wave running CH independent
// accumulator chains (CH = 1: every MFMA depends on the previous one, as in the projection's GEMM1; 12: as in its GEMM2).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f32_issue.hip -o tools/microbench/bin/mfma_f32_issue && ./...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void k(float* out, long long* cycles, int iters) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 48 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int CH> void run(int threads, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<CH>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<CH>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime counts at a fixed 100 MHz: report as is and per MFMA of one wave
  printf("chains %2d  waves/SIMD %d : %lld counter ticks for %d MFMAs per wave = %.3f ticks per MFMA per wave\n", CH, threads / 256, c,
         iters * 48, (double)c / (iters * 48));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64);
  for (int threads : {256, 512, 1024}) {
    run<1>(threads, out, cyc);
    run<2>(threads, out, cyc);
    run<4>(threads, out, cyc);
    run<12>(threads, out, cyc);
  }
  // wall-clock calibration of the counter: a long run timed with events
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<12>), dim3(1), dim3(256), 0, 0, out, cyc, 200000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("calibration: %lld ticks in %.3f ms -> %.1f MHz counter; 12 chains, 1 wave/SIMD: %.2f ns per MFMA\n", c, ms, c / ms * 1e-3,
         ms * 1e6 / (200000.0 * 48));
  return 0;
}
