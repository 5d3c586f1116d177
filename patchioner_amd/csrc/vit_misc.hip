// Bandwidth-bound helpers of the ViT encoder (gfx950): LayerNorm, im2col for the patch embedding,
// cls / register / pad token initialisation.  One wavefront per row, 16-byte accesses.
#include "common.h"
#include "kernels.h"

namespace pio {

// LayerNorm over rows of the fp32 residual stream (DINOv2 norm1 / norm2 / final norm, eps 1e-6).
// out16 != null: every padded row m -> out16[m] in operand precision (feeds the next GEMM).
// out32 != null: final norm; valid rows only, compacted from Tp to T rows per image (the
//                x_norm_clstoken | x_norm_regtokens | x_norm_patchtokens layout, P/src/dino_extraction.py:14-22).
template <typename T>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bvec, float eps, int M, int D,
                                                   T* out16, float* out32, int Tt, int Tp) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  int orow = m;
  if (out32 != nullptr) {
    const int b = m / Tp, t = m - b * Tp;
    if (t >= Tt) return;
    orow = b * Tt + t;
  }
  const int nv = D >> 2;  // float4 per row (D % 4 == 0)
  const float4* xr = (const float4*)(x + (size_t)m * D);
  float4 v[4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      v[i] = xr[c];
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float a = v[i].x - mean, b2 = v[i].y - mean, c2 = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b2 * b2) + (c2 * c2 + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float4 ww = ((const float4*)w)[c], bb = ((const float4*)bvec)[c];
      const float y0 = (v[i].x - mean) * rstd * ww.x + bb.x, y1 = (v[i].y - mean) * rstd * ww.y + bb.y;
      const float y2 = (v[i].z - mean) * rstd * ww.z + bb.z, y3 = (v[i].w - mean) * rstd * ww.w + bb.w;
      if (out32 != nullptr) {
        ((float4*)(out32 + (size_t)orow * D))[c] = make_float4(y0, y1, y2, y3);
      } else {
        typedef T half4_t __attribute__((ext_vector_type(4)));
        half4_t o;
        o[0] = (T)y0; o[1] = (T)y1; o[2] = (T)y2; o[3] = (T)y3;
        ((half4_t*)(out16 + (size_t)orow * D))[c] = o;
      }
    }
  }
}

hipError_t launch_layernorm(OperandType t, const float* x, const float* w, const float* b, float eps, int M, int D,
                            void* out16, float* out32, int T, int Tp, hipStream_t s) {
  if (D % 4 != 0 || D > 1024) return hipErrorInvalidValue;
  dim3 grid(ceil_div(M, 4));
  if (t == OP_F16) hipLaunchKernelGGL((k_layernorm<f16>), grid, dim3(256), 0, s, x, w, b, eps, M, D, (f16*)out16, out32, T, Tp);
  else hipLaunchKernelGGL((k_layernorm<bf16>), grid, dim3(256), 0, s, x, w, b, eps, M, D, (bf16*)out16, out32, T, Tp);
  return hipGetLastError();
}

// Patch rows for the patch-embedding GEMM: row (b, gy, gx), column k = c*p*p + py*p + px, which is the
// flattening order of the Conv2d(3, D, 14, 14) weight.  One thread per (row, c, py) run of p pixels.
template <typename T>
__global__ __launch_bounds__(256) void k_im2col(const float* __restrict__ imgs, int B, int S, int p, int n, int Kpad,
                                                T* out) {
  const int total = B * n * n * 3 * p;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int py = idx % p;
  int r = idx / p;
  const int c = r % 3;
  r /= 3;                       // patch row index b*n*n + gy*n + gx
  const int gx = r % n, gy = (r / n) % n, b = r / (n * n);
  const float* src = imgs + (((size_t)b * 3 + c) * S + (gy * p + py)) * S + gx * p;
  T* dst = out + (size_t)r * Kpad + c * p * p + py * p;
  for (int px = 0; px < p; ++px) dst[px] = (T)src[px];
}

hipError_t launch_im2col(OperandType t, const float* imgs, int B, int S, int p, int n, int Kpad, void* out,
                         hipStream_t s) {
  const int total = B * n * n * 3 * p;
  dim3 grid(ceil_div(total, 256));
  if (t == OP_F16) hipLaunchKernelGGL((k_im2col<f16>), grid, dim3(256), 0, s, imgs, B, S, p, n, Kpad, (f16*)out);
  else hipLaunchKernelGGL((k_im2col<bf16>), grid, dim3(256), 0, s, imgs, B, S, p, n, Kpad, (bf16*)out);
  return hipGetLastError();
}

// Global tokens of DINOv2 prepare_tokens_with_masks: row 0 = cls_token + pos_embed[0]; rows 1..R =
// register tokens (inserted after the position add, so no position term); pad rows T..Tp-1 = 0.
__global__ __launch_bounds__(256) void k_token_init(float* x, const float* cls, const float* pos0, const float* reg,
                                                    int B, int R, int Tt, int Tp, int D) {
  const int rows_per_img = 1 + R + (Tp - Tt);
  const int row = blockIdx.x;
  const int b = row / rows_per_img, j = row - b * rows_per_img;
  int t;
  if (j <= R) t = j; else t = Tt + (j - R - 1);
  float* dst = x + (size_t)(b * Tp + t) * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    float v;
    if (j == 0) v = cls[d] + pos0[d];
    else if (j <= R) v = reg[(j - 1) * D + d];
    else v = 0.f;
    dst[d] = v;
  }
}

hipError_t launch_token_init(float* x, const float* cls, const float* pos0, const float* reg, int B, int R, int T,
                             int Tp, int D, hipStream_t s) {
  dim3 grid(B * (1 + R + (Tp - T)));
  hipLaunchKernelGGL(k_token_init, grid, dim3(256), 0, s, x, cls, pos0, reg, B, R, T, Tp, D);
  return hipGetLastError();
}


// ---- double-DINO boxes: gather the block input of every (image, box) sequence ------------------------------
__global__ __launch_bounds__(256) void k_box_sequences(const float* __restrict__ tokens, const int32_t* __restrict__ slices,
                                                       int s_base, int NB, int T, int Tp, int G, int n, int D, int use_global,
                                                       float* __restrict__ x, int32_t* __restrict__ lens) {
  const int s = blockIdx.y, t = blockIdx.x;      // s: sequence inside this chunk; s_base + s: (image, box) index
  const int b = (s_base + s) / NB;
  const int32_t* sl = slices + 4 * (size_t)(s_base + s);
  // slices are python-normalised by the caller; clamped here so that a bad table can never index outside the grid
  const int ys = min(max(sl[0], 0), n), ye = min(max(sl[1], 0), n), xs = min(max(sl[2], 0), n), xe = min(max(sl[3], 0), n);
  const int hh = ye > ys ? ye - ys : 0, ww = xe > xs ? xe - xs : 0;
  const int Gs = use_global ? G : 0;
  const int len = Gs + hh * ww;
  if (t == 0 && threadIdx.x == 0) lens[s] = len;
  const float* src = nullptr;
  if (t < Gs) src = tokens + ((size_t)b * T + t) * D;
  else if (t < len) {
    const int r = t - Gs, y = ys + r / ww, xx = xs + r % ww;
    src = tokens + ((size_t)b * T + G + y * n + xx) * D;
  }
  float4* dst = (float4*)(x + ((size_t)s * Tp + t) * D);
  for (int c = threadIdx.x; c < D / 4; c += 256) dst[c] = src ? ((const float4*)src)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void k_box_seq_reduce(const float* __restrict__ x, const int32_t* __restrict__ lens, int Tp,
                                                        int D, int Gs, int mode, float* __restrict__ out) {
  const int s = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const float* xs = x + (size_t)s * Tp * D + d;
  if (mode == 0) { out[(size_t)s * D + d] = xs[0]; return; }
  const int len = lens[s];
  float acc = 0.f;
  for (int t = Gs; t < len; ++t) acc += xs[(size_t)t * D];
  out[(size_t)s * D + d] = acc / (float)(len - Gs);          // 0 / 0 = NaN for an empty region (torch.mean of no rows)
}

hipError_t launch_box_sequences(const float* tokens, const int32_t* slices, int s_base, int Ns, int NB, int T, int Tp, int G,
                                int n, int D, int use_global, float* x, int32_t* lens, hipStream_t s) {
  if (Ns < 1 || NB < 1 || D % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_box_sequences, dim3(Tp, Ns), dim3(256), 0, s, tokens, slices, s_base, NB, T, Tp, G, n, D, use_global, x, lens);
  return hipGetLastError();
}

hipError_t launch_box_seq_reduce(const float* x, const int32_t* lens, int Ns, int Tp, int D, int Gs, int mode, float* out,
                                 hipStream_t s) {
  if (Ns < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_box_seq_reduce, dim3(ceil_div(D, 256), Ns), dim3(256), 0, s, x, lens, Tp, D, Gs, mode, out);
  return hipGetLastError();
}

}  // namespace pio
