// Device-side image transforms: Resize(bicubic) -> CenterCrop -> ToTensor -> Normalize, bit-exact to the
// PIL / torchvision pipeline the reference runs on the host (P/src/model.py:347-357).
//
// Pillow's 8-bit resampler is two separable passes of 22-bit fixed-point integer arithmetic with a uint8 image
// between them (Resample.c: ImagingResampleHorizontal_8bpc, then ImagingResampleVertical_8bpc): per output sample
//     ss = 2^21 + sum_t pixel[min + t] * k[t];   out = clip8(ss >> 22)
// The coefficient tables (normalised in double, rounded to int) are built on the host by pio_preprocess (api.cpp),
// one table per image and axis, and only for the output columns / rows that survive the centre crop.
//   k_prep_horizontal: one thread per (intermediate row, output column): 3 channels, <= kh taps of the source row
//   k_prep_vertical  : one thread per output pixel: 3 channels, <= kv taps down the intermediate image, then the
//                      768-entry fp32 table ((v / 255) - mean_c) / std_c (built on the host in IEEE fp32, so
//                      the floats are the ones torch's ToTensor + Normalize produce); zero padding -> table[c][0].
// Byte work, a few MB per batch: nothing to tile; loads are as coalesced as 3-byte pixels allow.
#include "common.h"
#include "kernels.h"

namespace pio {

static constexpr int PREC_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= PREC_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void k_prep_horizontal(const uint8_t* __restrict__ pixels, const PrepImage* __restrict__ imgs,
                                                         const int32_t* __restrict__ tables, uint8_t* __restrict__ tmp) {
  const PrepImage im = imgs[blockIdx.y];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (im.nx <= 0 || e >= im.nr * im.nx) return;
  const int r = e / im.nx, x = e - r * im.nx;
  const int32_t* k = tables + im.coef_h + (int64_t)x * im.kh;
  const int xmin = tables[im.bnd_h + 2 * x], n = tables[im.bnd_h + 2 * x + 1];
  const uint8_t* src = pixels + im.src + ((int64_t)(im.r0 + r) * im.W + xmin) * 3;
  int s0 = 1 << (PREC_BITS - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int kt = k[t];
    s0 += (int)src[3 * t + 0] * kt;
    s1 += (int)src[3 * t + 1] * kt;
    s2 += (int)src[3 * t + 2] * kt;
  }
  uint8_t* d = tmp + im.tmp + ((int64_t)r * im.nx + x) * 3;
  d[0] = clip8(s0); d[1] = clip8(s1); d[2] = clip8(s2);
}

__global__ __launch_bounds__(256) void k_prep_vertical(const PrepImage* __restrict__ imgs, const int32_t* __restrict__ tables,
                                                       const uint8_t* __restrict__ tmp, const float* __restrict__ lut, int S,
                                                       float* __restrict__ out) {
  const PrepImage im = imgs[blockIdx.y];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= S * S) return;
  const int y = e / S, x = e - y * S;
  int v0 = 0, v1 = 0, v2 = 0;                       // zero padding outside the resized image
  const int yy = y - im.y0, xx = x - im.x0;
  if (yy >= 0 && yy < im.ny && xx >= 0 && xx < im.nx) {
    const int32_t* k = tables + im.coef_v + (int64_t)yy * im.kv;
    const int ymin = tables[im.bnd_v + 2 * yy], n = tables[im.bnd_v + 2 * yy + 1];   // ymin is relative to r0
    const uint8_t* src = tmp + im.tmp + ((int64_t)ymin * im.nx + xx) * 3;
    const int64_t stride = (int64_t)im.nx * 3;
    int s0 = 1 << (PREC_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
      const int kt = k[t];
      s0 += (int)src[t * stride + 0] * kt;
      s1 += (int)src[t * stride + 1] * kt;
      s2 += (int)src[t * stride + 2] * kt;
    }
    v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
  }
  float* o = out + (size_t)blockIdx.y * 3 * S * S + e;
  o[0] = lut[v0];
  o[(size_t)S * S] = lut[256 + v1];
  o[(size_t)2 * S * S] = lut[512 + v2];
}

hipError_t launch_preprocess(const uint8_t* pixels, const PrepImage* imgs, const int32_t* tables, uint8_t* tmp,
                             const float* lut, int B, int S, int max_tmp_elems, float* out, hipStream_t s) {
  if (B <= 0 || S <= 0) return hipErrorInvalidValue;
  if (max_tmp_elems > 0)
    hipLaunchKernelGGL(k_prep_horizontal, dim3(ceil_div(max_tmp_elems, 256), B), dim3(256), 0, s, pixels, imgs, tables, tmp);
  hipLaunchKernelGGL(k_prep_vertical, dim3(ceil_div(S * S, 256), B), dim3(256), 0, s, imgs, tables, tmp, lut, S, out);
  return hipGetLastError();
}

}  // namespace pio
