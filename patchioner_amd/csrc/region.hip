// Attention read-out and region weighting kernels (fp32, HBM/L2-bound wavefront reductions, gfx950).
//
//   k_cls_logits      process_self_attention, P/src/dino_extraction.py:24-34: only the CLS row of
//                     (q*scale) k^T is ever used, so only that row is computed (the reference
//                     materialises [B,16,T,T]).
//   k_softmax_rows    the two softmaxes of P/src/model.py:868-871.
//   k_region_reduce   every "weights x patch tokens" contraction of the path: traces (model.py:1054),
//                     boxes (bbox_utils.py:53,79,96,109), whole-image gaussian (model.py:90-92),
//                     attention-weighted means (model.py:869,872).
//   k_trace_grids     map_traces_to_grid, P/src/bbox_utils.py:158-168.
//   k_bbox_weights    the weight construction of extract_bboxes_feats, P/src/bbox_utils.py:37-104.
#include "common.h"
#include "kernels.h"

namespace pio {

// One wave per (image, patch).  Lane l covers D/64 consecutive channels; a read-out head is D/16
// channels = 4 lanes (quirk: always 16 heads and scale 0.125, P/src/model.py:336-337).
__global__ __launch_bounds__(256) void k_cls_logits(const float* __restrict__ qkv, int B, int T, int G, int D, int Hr,
                                                    float scale, float* mean_logits, float* head_logits) {
  const int lane = threadIdx.x & 63;
  const int n2 = T - G;
  const int wp = blockIdx.x * 4 + (threadIdx.x >> 6);   // b*n2 + p
  if (wp >= B * n2) return;
  const int b = wp / n2, p = wp - b * n2;
  const int per = D >> 6;
  const float* q = qkv + (size_t)b * T * 3 * D;                      // CLS row, q part
  const float* k = qkv + ((size_t)b * T + G + p) * 3 * D + D;        // patch row, k part
  if (Hr * 64 == D) {
    // ViT-S read-out: 6 heads x 64 channels (P/src/model.py:336).  Lane l takes channel l of every head, so head i
    // is one wave reduction.
    float tot = 0.f;
    for (int i = 0; i < Hr; ++i) {
      const float hs = wave_sum((q[64 * i + lane] * scale) * k[64 * i + lane]);
      if (head_logits != nullptr && lane == 0) head_logits[((size_t)b * Hr + i) * n2 + p] = hs;
      tot += hs;
    }
    if (lane == 0) mean_logits[(size_t)b * n2 + p] = tot / (float)Hr;
    return;
  }
  float part = 0.f;
  for (int i = 0; i < per; ++i) {
    const int d = lane * per + i;
    part += (q[d] * scale) * k[d];
  }
  float hs = part + __shfl_xor(part, 1);
  hs += __shfl_xor(hs, 2);
  if (head_logits != nullptr && (lane & 3) == 0) head_logits[((size_t)b * Hr + (lane >> 2)) * n2 + p] = hs;
  const float tot = wave_sum(part);
  if (lane == 0) mean_logits[(size_t)b * n2 + p] = tot / (float)Hr;
}

// Row softmax, one workgroup per row (rows of n2 <= a few thousand).
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ in, float* out, int n) {
  __shared__ float red[4];
  const float* r = in + (size_t)blockIdx.x * n;
  float* o = out + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float mx = -INFINITY;
  for (int i = tid; i < n; i += 256) mx = fmaxf(mx, r[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < n; i += 256) sum += expf(r[i] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wid] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.0f / sum;
  for (int i = tid; i < n; i += 256) o[i] = expf(r[i] - mx) * inv;
}

// out[r][:] = scale * sum_p w[r][p] * tokens[img(r)][G+p][:].  One workgroup per output row; each
// thread owns 4 channels (16-B loads, coalesced along D); zero weights are skipped (regions are
// small), NaN weights propagate like the reference's NaN means.
__global__ __launch_bounds__(256) void k_region_reduce(const float* __restrict__ tokens, int T, int G, int D, int n2,
                                                       const float* __restrict__ w, const int32_t* img_index,
                                                       float scale, float* out) {
  // grid (R, ceil(D/256)): lane = one float4 column of a 256-channel slab; the 4 waves split the patches
  // (p = wave, wave+4, ...) with the weight row staged in LDS, then meet in LDS.
  __shared__ float s_w[1536];
  __shared__ __attribute__((aligned(16))) float s_acc[4][256];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int img = img_index ? img_index[r] : r;
  for (int p = tid; p < n2; p += 256) s_w[p] = w[(size_t)r * n2 + p];
  __syncthreads();
  const int c = blockIdx.y * 64 + lane;            // float4 column
  const bool live = c * 4 < D;
  const float4* base = (const float4*)(tokens + ((size_t)img * T + G) * D) + (live ? c : 0);
  const int stride = D >> 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = wid; p < n2; p += 4) {
    const float wv = s_w[p];
    if (wv == 0.f) continue;                        // wave-uniform: regions are small
    const float4 t = base[(size_t)p * stride];
    acc.x += wv * t.x; acc.y += wv * t.y; acc.z += wv * t.z; acc.w += wv * t.w;
  }
  *(float4*)&s_acc[wid][4 * lane] = acc;
  __syncthreads();
  if (wid == 0 && live) {
    const float4 a0 = *(const float4*)&s_acc[0][4 * lane], a1 = *(const float4*)&s_acc[1][4 * lane];
    const float4 a2 = *(const float4*)&s_acc[2][4 * lane], a3 = *(const float4*)&s_acc[3][4 * lane];
    ((float4*)(out + (size_t)r * D))[c] = make_float4(((a0.x + a1.x) + (a2.x + a3.x)) * scale, ((a0.y + a1.y) + (a2.y + a3.y)) * scale,
                                                     ((a0.z + a1.z) + (a2.z + a3.z)) * scale, ((a0.w + a1.w) + (a2.w + a3.w)) * scale);
  }
}

__global__ __launch_bounds__(64) void k_trace_grids(const double* __restrict__ xy, const int32_t* __restrict__ offs,
                                                    int n, float* grids) {
  const int b = blockIdx.x;
  const double patch = 1.0 / (double)n;   // the reference divides by this double, it does not multiply by n
  float* g = grids + (size_t)b * n * n;
  for (int i = offs[b] + threadIdx.x; i < offs[b + 1]; i += 64) {
    const double x = xy[2 * i], y = xy[2 * i + 1];
    if (x >= 0.0 && x <= 1.0 && y >= 0.0 && y <= 1.0) {
      int gx = (int)(x / patch), gy = (int)(y / patch);
      gx = gx < n - 1 ? gx : n - 1;
      gy = gy < n - 1 ? gy : n - 1;
      atomicAdd(&g[gy * n + gx], 1.0f);   // small integer counts: exact and order-independent in fp32
    }
  }
}

// python slice [start:stop] on an axis of length n
__device__ __forceinline__ void py_slice(int start, int stop, int n, int& s, int& e) {
  if (start < 0) { start += n; if (start < 0) start = 0; } else if (start > n) start = n;
  if (stop < 0) { stop += n; if (stop < 0) stop = 0; } else if (stop > n) stop = n;
  s = start; e = stop > start ? stop : start;
}
// torch.linspace(-1, 1, steps)[i] in fp32 (symmetric two-sided evaluation, steps==1 -> -1)
__device__ __forceinline__ float linspace_pm1(int i, int steps) {
  if (steps == 1) return -1.0f;
  const float step = 2.0f / (float)(steps - 1);
  return i < steps / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(steps - 1 - i);
}

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup per image; boxes are walked in order because the attention-map mode renormalises the
// caller's map in place per box (later overlapping boxes see the rescaled values, bbox_utils.py:47-48).
__global__ __launch_bounds__(256) void k_bbox_weights(const int32_t* __restrict__ boxes, int NB, int n, int mode,
                                                      float variance, const int32_t* __restrict__ center, float* attn,
                                                      float* weights, int single_map, float* single) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, n2 = n * n;
  float* tot = single_map ? single + (size_t)b * n2 : nullptr;
  if (tot) for (int c = tid; c < n2; c += 256) tot[c] = 0.f;
  float* am = attn ? attn + (size_t)b * n2 : nullptr;
  for (int j = 0; j < NB; ++j) {
    const int32_t* bx = boxes + ((size_t)b * NB + j) * 4;
    const int x1 = bx[0], y1 = bx[1], w = bx[2], h = bx[3];
    float* wr = weights + ((size_t)b * NB + j) * n2;
    if (single_map && (x1 + y1 + w + h) < 0) {           // dummy box: skipped entirely (bbox_utils.py:40-42)
      for (int c = tid; c < n2; c += 256) wr[c] = 0.f;
      continue;
    }
    int ys, ye, xs, xe;
    py_slice(y1, y1 + h + 1, n, ys, ye);                 // inclusive end: [y1 : y2 + 1]
    py_slice(x1, x1 + w + 1, n, xs, xe);
    const int hs = ye - ys, ws = xe - xs;
    const bool empty = hs * ws == 0;
    float local[6];                                      // n2 <= 1536 cells per image
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = tid + 256 * i;
      float v = 0.f;
      if (c < n2) {
        const int cy = c / n, cx = c - cy * n;
        const bool in = cy >= ys && cy < ye && cx >= xs && cx < xe;
        if (in) {
          if (mode == 0) v = 1.0f;
          else if (mode == 1) {
            const float yy = linspace_pm1(cy - ys, hs), xx = linspace_pm1(cx - xs, ws);
            v = expf(-(xx * xx + yy * yy) / variance);
          } else if (mode == 2) {
            const int32_t* ch = center + ((size_t)b * NB + j) * 2;
            v = (cy - ys == ch[0] && cx - xs == ch[1]) ? 1.0f : 0.f;
          } else v = am[c];
        }
      }
      local[i] = v;
      part += v;
    }
    const float sum = block_sum256(part, red);
    // uniform: ones/(hs*ws); gaussian & attention: w / w.sum(); one-hot: as is
    const float inv = (mode == 2) ? 1.0f : 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = tid + 256 * i;
      if (c < n2) {
        const int cy = c / n, cx = c - cy * n;
        const bool in = cy >= ys && cy < ye && cx >= xs && cx < xe;
        float v = in ? local[i] * inv : 0.f;
        if (empty && mode == 0) v = NAN;                 // mean of an empty slice (bbox_utils.py:94)
        wr[c] = v;
        if (in) {
          if (mode == 3) am[c] = v;                      // in-place renormalisation of the caller's map
          if (tot) tot[c] += v;
        }
      }
    }
    __syncthreads();
  }
  if (tot) {
    float part = 0.f;
    for (int c = tid; c < n2; c += 256) part += tot[c];
    const float sum = block_sum256(part, red);
    for (int c = tid; c < n2; c += 256) tot[c] = tot[c] / sum;
  }
}

__global__ __launch_bounds__(256) void k_gaussian_map(int n, float variance, float* map) {
  __shared__ float red[4];
  const int tid = threadIdx.x, n2 = n * n;
  float local[6];
  float part = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = tid + 256 * i;
    float v = 0.f;
    if (c < n2) {
      const int cy = c / n, cx = c - cy * n;
      if (variance >= 100.f) v = 1.0f;
      else {
        const float yy = linspace_pm1(cy, n), xx = linspace_pm1(cx, n);
        v = expf(-(xx * xx + yy * yy) / variance);
      }
    }
    local[i] = v;
    part += v;
  }
  const float sum = block_sum256(part, red);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = tid + 256 * i;
    if (c < n2) map[c] = variance >= 100.f ? 1.0f / (float)n2 : local[i] / sum;
  }
}

hipError_t launch_cls_logits(const float* qkv_last, int B, int T, int G, int D, int Hr, float scale,
                             float* mean_logits, float* head_logits, hipStream_t s) {
  const int n2 = T - G;
  if (D % 64 != 0 || (Hr != 16 && Hr * 64 != D)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_cls_logits, dim3(ceil_div(B * n2, 4)), dim3(256), 0, s, qkv_last, B, T, G, D, Hr, scale,
                     mean_logits, head_logits);
  return hipGetLastError();
}

hipError_t launch_softmax_rows(const float* in, float* out, int rows, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_softmax_rows, dim3(rows), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}

hipError_t launch_trace_grids(const double* xy, const int32_t* offsets, int B, int n, float* grids, hipStream_t s) {
  hipError_t e = hipMemsetAsync(grids, 0, (size_t)B * n * n * sizeof(float), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_trace_grids, dim3(B), dim3(64), 0, s, xy, offsets, n, grids);
  return hipGetLastError();
}

hipError_t launch_bbox_weights(const int32_t* boxes, int B, int NB, int n, int mode, float variance,
                               const int32_t* center_choice, float* attn, float* weights, int single_map,
                               float* single, hipStream_t s) {
  if (n * n > 1536 || mode < 0 || mode > 3) return hipErrorInvalidValue;
  if (mode == 2 && center_choice == nullptr) return hipErrorInvalidValue;
  if (mode == 3 && attn == nullptr) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_bbox_weights, dim3(B), dim3(256), 0, s, boxes, NB, n, mode, variance, center_choice, attn,
                     weights, single_map, single);
  return hipGetLastError();
}

hipError_t launch_region_reduce(const float* tokens, int T, int G, int D, int n2, const float* weights,
                                const int32_t* img_index, int R, float scale, float* out, hipStream_t s) {
  if (D % 4 != 0 || D > 1024 || R <= 0 || n2 > 1536) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_region_reduce, dim3(R, ceil_div(D, 256)), dim3(256), 0, s, tokens, T, G, D, n2, weights, img_index, scale, out);
  return hipGetLastError();
}

hipError_t launch_gaussian_map(int n, float variance, float* map, hipStream_t s) {
  if (n * n > 1536 || !(variance > 0.f)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_gaussian_map, dim3(1), dim3(256), 0, s, n, variance, map);
  return hipGetLastError();
}


// ctx_cleaner (P/src/model.py:1425-1436): row r of `dirty` [R][D] belongs to context row r / rows_per_ctx of `ctx`.
//   mode 0 orthogonal_projection: out = dirty - alpha * (dirty . ctx / |ctx|^2) * ctx
//   mode 1 contrastive_mask     : out = dirty * (1 - ctx / (|ctx| + 1e-6))
//   normalize_inputs: dirty and ctx are L2-normalised first (the reference's clean_after_projection=False branch,
//   model.py:907-913).  One wave per row, the row in registers (D <= 1024); in place when out == dirty.
__global__ __launch_bounds__(256) void k_ctx_clean(const float* __restrict__ dirty, const float* __restrict__ ctx, int R, int D,
                                                   int rows_per_ctx, int mode, float alpha, int normalize_inputs, float* out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* dp = dirty + (size_t)r * D;
  const float* cp = ctx + (size_t)(r / rows_per_ctx) * D;
  float d[16], c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = lane + 64 * i;
    d[i] = k < D ? dp[k] : 0.f;
    c[i] = k < D ? cp[k] : 0.f;
  }
  if (normalize_inputs) {
    float nd = 0.f, nc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { nd += d[i] * d[i]; nc += c[i] * c[i]; }
    nd = sqrtf(wave_sum(nd));
    nc = sqrtf(wave_sum(nc));
#pragma unroll
    for (int i = 0; i < 16; ++i) { d[i] = d[i] / nd; c[i] = c[i] / nc; }
  }
  float dot = 0.f, nn = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dot += d[i] * c[i]; nn += c[i] * c[i]; }
  dot = wave_sum(dot);
  nn = wave_sum(nn);
  float* o = out + (size_t)r * D;
  const float nrm = sqrtf(nn);
  const float proj = dot / (nrm * nrm);              // torch.norm(ctx) ** 2, as the reference writes it
  const float den = nrm + 1e-6f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = lane + 64 * i;
    if (k < D) o[k] = mode == 0 ? d[i] - alpha * proj * c[i] : d[i] * (1.0f - c[i] / den);
  }
}

hipError_t launch_ctx_clean(const float* dirty, const float* ctx, int R, int D, int rows_per_ctx, int mode, float alpha,
                            int normalize_inputs, float* out, hipStream_t s) {
  if (R < 1 || D < 1 || D > 1024 || rows_per_ctx < 1 || mode < 0 || mode > 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_ctx_clean, dim3(ceil_div(R, 4)), dim3(256), 0, s, dirty, ctx, R, D, rows_per_ctx, mode, alpha,
                     normalize_inputs, out);
  return hipGetLastError();
}

}  // namespace pio
