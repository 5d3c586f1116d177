// Round 3 re-measurement of "does another wave's VALU run under MFMAs on gfx950?".  Round 2's probe (mfma_valu_overlap.hip)
// was compiled into v_pk_fma_f32 by hipcc's SLP vectoriser, so its "no overlap" verdict is about PACKED fp32 only.  Here the
// VALU streams are inline asm (the opcode is what the mode says) and three placements are timed:
//   cross-wave: waves 0-3 MFMA, waves 4-7 VALU (one pair per SIMD)          -> max(T) if they overlap, sum if not
//   same-wave : one wave per SIMD issues NV valu instructions after every MFMA (NV = 0, 2, 4, 6, 8)
// hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap2.hip -o tools/microbench/bin/mfma_valu_overlap2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { V_FMA = 1, V_PKFMA = 2, V_EXP = 3, V_CVTPK = 4, V_ADD = 5, V_MOV = 6, V_PKFMA16 = 7, V_EXP16 = 8, V_PKMAX16 = 9 };   // 7-9: round 4 (packed / transcendental fp16)

template <int KIND>
__device__ __forceinline__ void valu8(float (&v)[8], f32x2 (&p)[4]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (KIND == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(0.999f), "v"(1e-3f));
    else if (KIND == V_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(1e-3f));
    else if (KIND == V_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(v[j]) : "v"(v[(j + 1) & 7]));
    else if (KIND == V_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
    else if (KIND == V_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 1) & 7]));
    else if (KIND == V_PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[j]) : "v"(v[(j + 1) & 7]), "v"(v[(j + 2) & 7]));
    else if (KIND == V_EXP16) asm volatile("v_exp_f16 %0, %0" : "+v"(v[j]));
    else if (KIND == V_PKMAX16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[j]) : "v"(v[(j + 1) & 7]));
    else if (KIND == V_PKFMA) { if (j < 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(p[(j + 1) & 3]), "v"(p[(j + 2) & 3])); }
  }
}

// cross-wave
template <int KIND>
__global__ __launch_bounds__(512) void k_cross(float* out, int iters, int do_mfma, int do_valu) {
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
  } else if (do_valu) {
    float v[8]; f32x2 p[4];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-4f + j;
    for (int j = 0; j < 4; ++j) p[j] = (f32x2){v[j], v[j + 4]};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) valu8<KIND>(v, p);
    }
    for (int j = 0; j < 8; ++j) sink += v[j];
    for (int j = 0; j < 4; ++j) sink += p[j][0] + p[j][1];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

// same-wave: NV valu instructions after every MFMA; 256 threads = one wave per SIMD
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k_same(float* out, int iters) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(1.0f + j * 1e-2f); }
  float v[8]; f32x2 p[4];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-4f + j;
  for (int j = 0; j < 4; ++j) p[j] = (f32x2){v[j], v[j + 4]};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (KIND == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(0.999f), "v"(1e-3f));
        else if (KIND == V_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 3]) : "v"(p[(j + 1) & 3]), "v"(p[(j + 2) & 3]));
        else if (KIND == V_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
        else if (KIND == V_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]));
        else if (KIND == V_PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(v[(j + 1) & 7]), "v"(v[(j + 2) & 7]));
        else if (KIND == V_EXP16) asm volatile("v_exp_f16 %0, %0" : "+v"(v[j & 7]));
      }
    }
  }
  float sink = 0.f;
  for (int c = 0; c < 4; ++c) sink += acc[c][0] + acc[c][7];
  for (int j = 0; j < 8; ++j) sink += v[j];
  for (int j = 0; j < 4; ++j) sink += p[j][0] + p[j][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <typename F>
static float timed(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int KIND>
static void cross(const char* name, float* out, int iters, int per_iter) {
  const float tm = timed([&] { hipLaunchKernelGGL(k_cross<KIND>, dim3(1), dim3(512), 0, 0, out, iters, 1, 0); });
  const float tv = timed([&] { hipLaunchKernelGGL(k_cross<KIND>, dim3(1), dim3(512), 0, 0, out, iters, 0, 1); });
  const float tb = timed([&] { hipLaunchKernelGGL(k_cross<KIND>, dim3(1), dim3(512), 0, 0, out, iters, 1, 1); });
  printf("cross-wave %-14s MFMA alone %.2f ms, VALU alone %.2f ms (%.2f ns/instr), together %.2f ms  (sum %.2f, max %.2f)\n", name, tm, tv,
         tv * 1e6 / ((double)iters * per_iter), tb, tm + tv, tm > tv ? tm : tv);
}

template <int KIND, int NV>
static void same(const char* name, float* out, int iters) {
  const float t = timed([&] { hipLaunchKernelGGL((k_same<KIND, NV>), dim3(1), dim3(256), 0, 0, out, iters); });
  printf("same-wave  %-14s %d per MFMA: %.2f ns per MFMA\n", name, NV, t * 1e6 / ((double)iters * 4));
}

int main() {
  float* out; hipMalloc(&out, 4096);
  const int iters = 100000;
  cross<V_FMA>("v_fma_f32", out, iters, 32);
  cross<V_ADD>("v_add_f32", out, iters, 32);
  cross<V_MOV>("v_mov_b32", out, iters, 32);
  cross<V_PKFMA>("v_pk_fma_f32", out, iters, 16);
  cross<V_EXP>("v_exp_f32", out, iters, 32);
  cross<V_PKFMA16>("v_pk_fma_f16", out, iters, 32);
  cross<V_PKMAX16>("v_pk_max_f16", out, iters, 32);
  cross<V_EXP16>("v_exp_f16", out, iters, 32);
  same<V_PKFMA16, 2>("v_pk_fma_f16", out, iters); same<V_PKFMA16, 4>("v_pk_fma_f16", out, iters); same<V_PKFMA16, 6>("v_pk_fma_f16", out, iters);
  cross<V_CVTPK>("v_cvt_pk_f16", out, iters, 32);
  same<V_FMA, 0>("v_fma_f32", out, iters);
  same<V_FMA, 2>("v_fma_f32", out, iters);
  same<V_FMA, 4>("v_fma_f32", out, iters);
  same<V_FMA, 6>("v_fma_f32", out, iters);
  same<V_FMA, 8>("v_fma_f32", out, iters);
  same<V_PKFMA, 2>("v_pk_fma_f32", out, iters);
  same<V_PKFMA, 4>("v_pk_fma_f32", out, iters);
  same<V_EXP, 2>("v_exp_f32", out, iters);
  same<V_EXP, 4>("v_exp_f32", out, iters);
  same<V_CVTPK, 4>("v_cvt_pk_f16", out, iters);
  return 0;
}
