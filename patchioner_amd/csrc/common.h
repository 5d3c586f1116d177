// Shared device/host helpers for libpatchioner_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pio {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> struct Vec8;
template <> struct Vec8<f16> { typedef _Float16 type __attribute__((ext_vector_type(8))); };
template <> struct Vec8<bf16> { typedef __bf16 type __attribute__((ext_vector_type(8))); };

// D = A(32x16) * B(16x32) + C, fp32 accumulate.  Lane l (r = l&31, h = l>>5) holds A[r][8h+j],
// B[8h+j][r] (j = 0..7); C/D register i holds row (i&3)+8*(i>>2)+4*h, column r.
__device__ __forceinline__ f32x16 mfma32(Vec8<f16>::type a, Vec8<f16>::type b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(Vec8<bf16>::type a, Vec8<bf16>::type b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row32(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous run of logical
// tile ids so that neighbouring tiles (same A panel) hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace pio
