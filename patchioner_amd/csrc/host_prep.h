// Host-side (load-time) weight preparation: operand-precision conversion and the DINOv2 position-grid
// interpolation.  Runs once in pio_finalize_weights, never on the forward path.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pio {

inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // keep NaN a NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                              // round to nearest even
  return (uint16_t)(u >> 16);
}

inline uint16_t f32_to_f16_bits(float f) {
  const _Float16 h = (_Float16)f;  // IEEE round-to-nearest-even, saturates to inf
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

inline void convert_row(bool to_f16, const float* src, int64_t n, uint16_t* dst) {
  if (to_f16) for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_f16_bits(src[i]);
  else for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_bf16_bits(src[i]);
}

// ---- torch.nn.functional.interpolate(mode="bicubic", antialias=True, size=(n,n)), align_corners=False ----
// Separable, Keys cubic with a = -0.5 (the anti-aliased kernel), support scaled by the down-sampling
// ratio, weights normalised per output sample; horizontal pass first, then vertical.
inline float cubic_aa(float x) {
  const float a = -0.5f;
  x = std::fabs(x);
  if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
  if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
  return 0.0f;
}

struct AaTaps {
  std::vector<int> xmin, xsize;
  std::vector<float> w;  // [out][max_taps]
  int max_taps;
};

inline AaTaps aa_taps(int in, int out) {
  AaTaps t;
  const float scale = (float)in / (float)out;
  const float support = scale >= 1.0f ? 2.0f * scale : 2.0f;
  const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  t.max_taps = (int)std::ceil(support) * 2 + 1;
  t.xmin.resize(out);
  t.xsize.resize(out);
  t.w.assign((size_t)out * t.max_taps, 0.f);
  for (int i = 0; i < out; ++i) {
    const float center = scale * ((float)i + 0.5f);
    const int xmin = std::max((int)(center - support + 0.5f), 0);
    const int xsize = std::min((int)(center + support + 0.5f), in) - xmin;
    float total = 0.f;
    float* w = &t.w[(size_t)i * t.max_taps];
    for (int j = 0; j < xsize; ++j) {
      w[j] = cubic_aa(((float)(j + xmin) - center + 0.5f) * invscale);
      total += w[j];
    }
    for (int j = 0; j < xsize; ++j) w[j] /= total;
    t.xmin[i] = xmin;
    t.xsize[i] = xsize;
  }
  return t;
}

// pos: [1 + g*g][D] (row 0 = class position) -> out: [1 + n*n][D].  DINOv2 interpolate_pos_encoding with
// interpolate_antialias=True, interpolate_offset=0.0 (the *_reg hub models); identity when n == g.
inline void interpolate_pos_embed(const float* pos, int g, int D, int n, float* out) {
  std::memcpy(out, pos, sizeof(float) * D);
  if (n == g) {
    std::memcpy(out + D, pos + D, sizeof(float) * (size_t)g * g * D);
    return;
  }
  const AaTaps tx = aa_taps(g, n), ty = aa_taps(g, n);
  std::vector<float> tmp((size_t)g * n * D);  // [g rows][n cols][D]: horizontal pass
  for (int y = 0; y < g; ++y)
    for (int ox = 0; ox < n; ++ox) {
      float* dst = &tmp[((size_t)y * n + ox) * D];
      std::fill(dst, dst + D, 0.f);
      const float* w = &tx.w[(size_t)ox * tx.max_taps];
      for (int j = 0; j < tx.xsize[ox]; ++j) {
        const float* src = pos + D + ((size_t)y * g + tx.xmin[ox] + j) * D;
        const float wj = w[j];
        for (int d = 0; d < D; ++d) dst[d] += wj * src[d];
      }
    }
  for (int oy = 0; oy < n; ++oy)
    for (int ox = 0; ox < n; ++ox) {
      float* dst = out + D + ((size_t)oy * n + ox) * D;
      std::fill(dst, dst + D, 0.f);
      const float* w = &ty.w[(size_t)oy * ty.max_taps];
      for (int j = 0; j < ty.xsize[oy]; ++j) {
        const float* src = &tmp[((size_t)(ty.xmin[oy] + j) * n + ox) * D];
        const float wj = w[j];
        for (int d = 0; d < D; ++d) dst[d] += wj * src[d];
      }
    }
}

// ---- torch.nn.functional.interpolate(mode="bicubic", antialias=False, scale_factor=(n + offset) / g) ----
// What the hub builds the models WITHOUT registers with (interpolate_antialias=False, interpolate_offset=0.1): ATen's
// upsample_bicubic2d, align_corners=False: source coordinate (o + 0.5) / scale_factor - 0.5 (the scale factor that was
// GIVEN, not n / g), Keys cubic with a = -0.75, four taps with clamped indices; output size floor(g * scale) = n.
inline void cubic_taps_m075(float t, float w[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

inline void interpolate_pos_embed_plain(const float* pos, int g, int D, int n, double offset, float* out) {
  std::memcpy(out, pos, sizeof(float) * D);
  if (n == g) {
    std::memcpy(out + D, pos + D, sizeof(float) * (size_t)g * g * D);
    return;
  }
  const float inv = (float)(1.0 / (((double)n + offset) / (double)g));     // ATen: static_cast<float>(1.0 / scale_factor)
  std::vector<int> idx((size_t)n * 4);
  std::vector<float> wt((size_t)n * 4);
  for (int o = 0; o < n; ++o) {
    const float real = inv * ((float)o + 0.5f) - 0.5f;
    const float fl = std::floor(real);
    cubic_taps_m075(real - fl, &wt[(size_t)o * 4]);
    for (int k = 0; k < 4; ++k) idx[(size_t)o * 4 + k] = std::min(std::max((int)fl - 1 + k, 0), g - 1);
  }
  for (int oy = 0; oy < n; ++oy)
    for (int ox = 0; ox < n; ++ox) {
      float* dst = out + D + ((size_t)oy * n + ox) * D;
      for (int d = 0; d < D; ++d) {
        float acc = 0.f;
        for (int j = 0; j < 4; ++j) {
          const float* row = pos + D + (size_t)idx[(size_t)oy * 4 + j] * g * D + d;
          float r = 0.f;
          for (int i = 0; i < 4; ++i) r += wt[(size_t)ox * 4 + i] * row[(size_t)idx[(size_t)ox * 4 + i] * D];
          acc += wt[(size_t)oy * 4 + j] * r;
        }
        dst[d] = acc;
      }
    }
}

}  // namespace pio
