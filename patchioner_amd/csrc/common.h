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

// x combined with the lanes 16 / 32 away, without the LDS crossbar (ds_bpermute, ~130 cycles on a serial chain): gfx950's
// v_permlane16_swap / v_permlane32_swap exchange rows / halves between two registers; fed (x, x) they return (own-or-partner,
// partner-or-own), and max / + are commutative, so the result is bit-identical to x op __shfl_xor(x, 16 | 32).
__device__ __forceinline__ float xor16_max(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_add(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_add(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Reductions over the 64 lanes, result in every lane.  The butterfly v op= v[lane ^ o], o = 32, 16, 8, 4, 2, 1, WITHOUT the LDS crossbar:
// __shfl_xor is a ds_bpermute_b32 (~130 cycles each on a serial chain; six of them are 0.4 us, and the decoder's kernels are 5 us long).
// lane ^ 32 / ^ 16 come from v_permlane32_swap / v_permlane16_swap (above), ^ 8 and ^ 4 from DPP row rotations by 8 and 4 -- a rotation
// is not an XOR, but once the values agree across ^ 8 (and ^ 4) the lane it reads holds the XOR partner's value -- and ^ 2, ^ 1 from DPP
// quad permutations.  Same operands, same order of operations: bit-identical to the __shfl_xor form for + and max.
template <int CTRL> __device__ __forceinline__ float pio_dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = xor32_add(v);
  v = xor16_add(v);
  v += pio_dpp_f32<0x128>(v);     // row_ror:8
  v += pio_dpp_f32<0x124>(v);     // row_ror:4
  v += pio_dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
  v += pio_dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = xor32_max(v);
  v = xor16_max(v);
  v = fmaxf(v, pio_dpp_f32<0x128>(v));
  v = fmaxf(v, pio_dpp_f32<0x124>(v));
  v = fmaxf(v, pio_dpp_f32<0x4E>(v));
  v = fmaxf(v, pio_dpp_f32<0xB1>(v));
  return v;
}

// GELU(v) = 0.5 v (1 + erf(v / sqrt 2)), the exact-erf form DINOv2's MLP uses, to 7.5e-7 absolute (fp32 evaluation,
// checked against scipy on [-12, 12]; the result is rounded to fp16/bf16 anyway).  erfc(t) = 2^(-t q(t)) with a degree-5
// polynomial q fitted on [0, 4.3] (least squares weighted for the absolute error of erf; erfc(4.3) = 1.2e-9).  With
// u = min(|v|, 4.3 sqrt 2), E = 2^-(1 + u q'(u)) = erfc(u / sqrt 2) / 2 (q' = q rescaled to the argument v) and h = v E:
//     GELU(v) = max(v, 0) - |h|          (v > 0: v - v E;  v < 0: v E)
// One transcendental (v_exp_f32) and 11 plain VALU instructions per element; the Abramowitz-Stegun 7.1.26 form it
// replaces needed v_rcp + v_exp + 16 (the epilogue of the fc1 GEMM is VALU-bound: 64 Ki elements per 256 x 256 tile).
// NaN in, NaN out; +-inf gives NaN (activations are finite: the operand type saturates long before).
__device__ __forceinline__ float gelu_erf(float v) {
  const float u = fminf(fabsf(v), 6.0811183f);
  float q = -1.971039006e-05f;
  q = fmaf(q, u, 6.613329563e-04f);
  q = fmaf(q, u, -7.757447031e-03f);
  q = fmaf(q, u, 5.296219534e-02f);
  q = fmaf(q, u, 4.590671448e-01f);
  q = fmaf(q, u, 1.151118979e+00f);
  const float h = v * __builtin_amdgcn_exp2f(-fmaf(u, q, 1.0f));
  return fmaxf(v, 0.f) - fabsf(h);
}

// gelu_erf on two values at once: the polynomial and the final product as packed fp32 FMAs / multiplies (v_pk_fma_f32: two
// IEEE FMAs per instruction, so each component is bit-identical to gelu_erf).  The fc1 epilogue evaluates 64 Ki of these per
// tile and is VALU-bound.
typedef float pio_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pio_f32x2 gelu_erf2(pio_f32x2 v) {
  pio_f32x2 u;
  u[0] = fminf(fabsf(v[0]), 6.0811183f);
  u[1] = fminf(fabsf(v[1]), 6.0811183f);
  const pio_f32x2 c5 = {-1.971039006e-05f, -1.971039006e-05f}, c4 = {6.613329563e-04f, 6.613329563e-04f};
  const pio_f32x2 c3 = {-7.757447031e-03f, -7.757447031e-03f}, c2 = {5.296219534e-02f, 5.296219534e-02f};
  const pio_f32x2 c1 = {4.590671448e-01f, 4.590671448e-01f}, c0 = {1.151118979e+00f, 1.151118979e+00f}, one = {1.0f, 1.0f};
  pio_f32x2 q = __builtin_elementwise_fma(c5, u, c4);
  q = __builtin_elementwise_fma(q, u, c3);
  q = __builtin_elementwise_fma(q, u, c2);
  q = __builtin_elementwise_fma(q, u, c1);
  q = __builtin_elementwise_fma(q, u, c0);
  const pio_f32x2 e = __builtin_elementwise_fma(u, q, one);
  pio_f32x2 x;
  x[0] = __builtin_amdgcn_exp2f(-e[0]);
  x[1] = __builtin_amdgcn_exp2f(-e[1]);
  const pio_f32x2 h = v * x;
  pio_f32x2 r;
  r[0] = fmaxf(v[0], 0.f) - fabsf(h[0]);
  r[1] = fmaxf(v[1], 0.f) - fabsf(h[1]);
  return r;
}

// QuickGELU of the OpenAI-CLIP towers, x * sigmoid(1.702 x) (P/src/model.py:363-365), as x / (1 + 2^(-1.702 log2(e) x)):
// one v_exp_f32, one v_rcp_f32 (1 ulp), three plain VALU.  One definition for every GEMM kernel: an output element does not
// depend on which of them produced it.  (x -> -inf: 2^(+inf) = inf, rcp = 0, x * 0 = -0 ... only at |x| beyond fp16 anyway.)
__device__ __forceinline__ float quick_gelu(float v) {
  const float e = __builtin_amdgcn_exp2f(v * -2.4554669595930157f);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// hipFuncSetAttribute (the opt-in above 48 KiB of dynamic LDS) is per DEVICE: one flag per device for the once-only call, so a
// second engine on another device of the same process gets its own opt-in
struct DeviceOnce {
  bool done[64] = {};
  bool& flag() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return done[dev & 63];
  }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace pio
