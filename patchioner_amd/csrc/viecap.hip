// ViECap head, device side (gfx950): the mapping network and the prompt assembly.
//
// Replaces, on the path Patchioner.caption_tokens -> VieCap.forward (P/src/model.py:1394-1398,
// P/src/viecap/entrypoint.py:98-153):
//   MappingNetwork.forward (P/src/viecap/ClipCap.py:122-153): Linear(C -> 10 x 768), concatenation with the learnt
//     prefix, 8 pre-LN transformer layers (8 heads x 96, bias-free q / kv projections, ReLU MLP of ratio 2, LN eps 1e-5),
//     last 10 positions;
//   image_text_simiarlity (P/src/viecap/retrieval_categories.py:61-95): softmax(f t^T / T) over the entity vocabulary;
//   word_embed + torch.cat of soft and hard prompt (entrypoint.py:126-133).
// The 64-step greedy search itself (P/src/viecap/search.py:108-191) runs on the decoder kernels (decoder.hip:
// launch_decode_prompted).
//
// Everything is exact fp32 (v_mfma_f32_16x16x4_f32 = an fp32 FMA chain): the head feeds a greedy arg-max.  The GEMMs
// here are small (rows = 20 per caption, K, N <= 7680; 1.5 GFLOP per caption against 15.8 for the language model): one
// plain LDS-tiled kernel with fused bias / ReLU / residual / scale, no hand scheduling.
#include "common.h"
#include "kernels.h"

namespace pio {

__device__ __forceinline__ f32x4 vc_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// C[m][n] = alpha * sum_k A[m][k] W[n][k] (+ bias[n]) (ReLU) (+ C[m][n] if residual);  A [M][lda], W [N][ldw], C [M][ldc].
// 64 x 64 tile per 256-thread workgroup (4 waves as 2 x 2, each 32 x 32 = 2 x 2 MFMA tiles), K in chunks of 32 through LDS
// (rows padded by 4 floats: conflict-free for the per-lane reads below).  Ragged M and N: loads clamp, stores mask.
// MFMA 16x16x4 operand map: lane (li = lane & 15, kq = lane >> 4): A[row li][k = 4 t + kq] ... any fixed bijection
// between (instruction t, lane group kq) and k works as long as A and B agree; C: column li, rows 4 kq + i.
template <int RELU, int RESID>
__global__ __launch_bounds__(256) void k_sgemm_tn(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                  const float* __restrict__ bias, float alpha, float* C, int ldc, int M, int N, int K) {
  constexpr int TK = 32, LD = TK + 4;
  __shared__ __attribute__((aligned(16))) float sA[64 * LD];
  __shared__ __attribute__((aligned(16))) float sW[64 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1, li = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // staging: 64 rows x 8 float4 per operand = 512 float4, two per thread
  const int srow = tid >> 3, sc4 = tid & 7;
  for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = srow + 32 * h;
      const int am = min(m0 + r, M - 1), wn = min(n0 + r, N - 1);
      *(float4*)(sA + r * LD + 4 * sc4) = *(const float4*)(A + (size_t)am * lda + k0 + 4 * sc4);
      *(float4*)(sW + r * LD + 4 * sc4) = *(const float4*)(W + (size_t)wn * ldw + k0 + 4 * sc4);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TK / 4; ++t) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = sA[(wr * 32 + i * 16 + li) * LD + 4 * t + kq];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = sW[(wc * 32 + j * 16 + li) * LD + 4 * t + kq];
      // rows of C come from the A operand's lane index: D = A(16 x 4) B(4 x 16), A[i = li][k], B[k][j = li]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = vc_mfma16(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  // C/D map of 16x16x4: column = lane & 15 (the B operand's lane), rows 4 kq + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 32 + j * 16 + li;
      if (n >= N) continue;
      const float bn = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 32 + i * 16 + 4 * kq + r;
        if (m >= M) continue;
        float v = alpha * acc[i][j][r] + bn;
        if (RELU) v = fmaxf(v, 0.f);
        float* dst = C + (size_t)m * ldc + n;
        if (RESID) v += *dst;
        *dst = v;
      }
    }
}

hipError_t launch_sgemm_tn(const float* A, int lda, const float* W, int ldw, const float* bias, float alpha, float* C, int ldc,
                           int M, int N, int K, int relu, int resid, hipStream_t s) {
  if (M < 1 || N < 1 || K < 32 || K % 32 != 0 || lda % 4 != 0 || ldw % 4 != 0) return hipErrorInvalidValue;
  const dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
  if (relu && !resid) hipLaunchKernelGGL((k_sgemm_tn<1, 0>), grid, dim3(256), 0, s, A, lda, W, ldw, bias, alpha, C, ldc, M, N, K);
  else if (!relu && resid) hipLaunchKernelGGL((k_sgemm_tn<0, 1>), grid, dim3(256), 0, s, A, lda, W, ldw, bias, alpha, C, ldc, M, N, K);
  else if (!relu && !resid) hipLaunchKernelGGL((k_sgemm_tn<0, 0>), grid, dim3(256), 0, s, A, lda, W, ldw, bias, alpha, C, ldc, M, N, K);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// y = act(y) in place: 2 = tanh, 3 = sigmoid (the activations ProjectionLayer.from_config offers besides ReLU, talk2dino.py:44-53)
__global__ __launch_bounds__(256) void k_activation_f32(float* y, size_t n, int act) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = y[i];
  y[i] = act == 2 ? tanhf(v) : 1.0f / (1.0f + expf(-v));
}

hipError_t launch_activation_f32(float* y, size_t n, int act, hipStream_t s) {
  if (act != 2 && act != 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_activation_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, n, act);
  return hipGetLastError();
}

// rows of x [M][D] -> LayerNorm (biased variance, eps inside the root: torch.nn.LayerNorm) -> y [M][D], one wave per row
__global__ __launch_bounds__(256) void k_layernorm_f32(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       float eps, int M, int D, float* y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* r = x + (size_t)row * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += r[d];
  const float mu = wave_sum(s) / (float)D;
  float v = 0.f;
  for (int d = lane; d < D; d += 64) { const float c = r[d] - mu; v += c * c; }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
  for (int d = lane; d < D; d += 64) y[(size_t)row * D + d] = (r[d] - mu) * rstd * w[d] + b[d];
}

hipError_t launch_layernorm_f32(const float* x, const float* w, const float* b, float eps, int M, int D, float* y, hipStream_t s) {
  hipLaunchKernelGGL(k_layernorm_f32, dim3(ceil_div(M, 4)), dim3(256), 0, s, x, w, b, eps, M, D, y);
  return hipGetLastError();
}

// x[n][0..Lp) = lin[n] viewed as [Lp][E] (the projected feature), x[n][Lp..Lp+Lc) = prefix_const
__global__ __launch_bounds__(256) void k_map_inputs(const float* __restrict__ lin, const float* __restrict__ prefix_const, int Lp, int Lc,
                                                    int E, float* x) {
  const int n = blockIdx.y, t = blockIdx.x;      // token t of sample n
  const float* src = t < Lp ? lin + ((size_t)n * Lp + t) * E : prefix_const + (size_t)(t - Lp) * E;
  float* dst = x + ((size_t)n * (Lp + Lc) + t) * E;
  for (int d = threadIdx.x; d < E; d += 256) dst[d] = src[d];
}

// MultiHeadAttention (ClipCap.py:52-70) of one (sample, head): S tokens (<= 32), head size hd (<= 128), no mask.
// q [M][E], kv [M][2E] viewed as [m][2][heads][hd]: keys = columns [0, E), values = columns [E, 2E).
__global__ __launch_bounds__(256) void k_map_attention(const float* __restrict__ q, const float* __restrict__ kv, int S, int E, int heads,
                                                       float* out) {
  __shared__ float s_p[32][33];
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads, hd = E / heads, tid = threadIdx.x;
  const float scale = 1.0f / sqrtf((float)hd);
  const float* qb = q + (size_t)n * S * E + h * hd;
  const float* kb = kv + (size_t)n * S * 2 * E + h * hd;
  const float* vb = kb + E;
  for (int p = tid; p < S * S; p += 256) {
    const int i = p / S, j = p - i * S;
    float s = 0.f;
    for (int d = 0; d < hd; ++d) s += qb[(size_t)i * E + d] * kb[(size_t)j * 2 * E + d];
    s_p[i][j] = s * scale;
  }
  __syncthreads();
  if (tid < S) {                                   // softmax over the keys of query tid
    float mx = -INFINITY;
    for (int j = 0; j < S; ++j) mx = fmaxf(mx, s_p[tid][j]);
    float sum = 0.f;
    for (int j = 0; j < S; ++j) { const float e = expf(s_p[tid][j] - mx); s_p[tid][j] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < S; ++j) s_p[tid][j] *= inv;
  }
  __syncthreads();
  for (int p = tid; p < S * hd; p += 256) {
    const int i = p / hd, d = p - i * hd;
    float o = 0.f;
    for (int j = 0; j < S; ++j) o += s_p[i][j] * vb[(size_t)j * 2 * E + d];
    out[((size_t)n * S + i) * E + h * hd + d] = o;
  }
}

// out[n][c][:] = x[n][Lp + c][:]   (MappingNetwork.forward: outputs[:, clip_project_length:, :])
__global__ __launch_bounds__(256) void k_map_take_tail(const float* __restrict__ x, int Lp, int Lc, int E, float* out) {
  const int n = blockIdx.y, c = blockIdx.x;
  const float* src = x + ((size_t)n * (Lp + Lc) + Lp + c) * E;
  float* dst = out + ((size_t)n * Lc + c) * E;
  for (int d = threadIdx.x; d < E; d += 256) dst[d] = src[d];
}

hipError_t launch_viecap_mapping(const ViecapMapArgs& a, hipStream_t s) {
  const int N = a.N, E = a.E, S = a.Lp + a.Lc, M = N * S, H = a.hidden;
  if (S > 32 || E % a.heads != 0 || E / a.heads > 128 || a.C % 32 != 0 || E % 32 != 0 || H % 32 != 0) return hipErrorInvalidValue;
  hipError_t e;
  // image_features /= norm (in place, entrypoint.py:108) is done by the caller (k_l2norm_rows of project.hip)
  if ((e = launch_sgemm_tn(a.feats, a.C, a.lin_w, a.C, a.lin_b, 1.f, a.lin, a.Lp * E, N, a.Lp * E, a.C, 0, 0, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_map_inputs, dim3(S, N), dim3(256), 0, s, a.lin, a.prefix_const, a.Lp, a.Lc, E, a.x);
  for (int l = 0; l < a.layers; ++l) {
    const ViecapMapLayerW& w = a.layer[l];
    if ((e = launch_layernorm_f32(a.x, w.n1w, w.n1b, a.eps, M, E, a.ln, s)) != hipSuccess) return e;
    if ((e = launch_sgemm_tn(a.ln, E, w.q_w, E, nullptr, 1.f, a.q, E, M, E, E, 0, 0, s)) != hipSuccess) return e;
    if ((e = launch_sgemm_tn(a.ln, E, w.kv_w, E, nullptr, 1.f, a.kv, 2 * E, M, 2 * E, E, 0, 0, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_map_attention, dim3(N * a.heads), dim3(256), 0, s, a.q, a.kv, S, E, a.heads, a.att);
    if ((e = launch_sgemm_tn(a.att, E, w.proj_w, E, w.proj_b, 1.f, a.x, E, M, E, E, 0, 1, s)) != hipSuccess) return e;
    if ((e = launch_layernorm_f32(a.x, w.n2w, w.n2b, a.eps, M, E, a.ln, s)) != hipSuccess) return e;
    if ((e = launch_sgemm_tn(a.ln, E, w.fc1_w, E, w.fc1_b, 1.f, a.hid, H, M, H, E, 1, 0, s)) != hipSuccess) return e;
    if ((e = launch_sgemm_tn(a.hid, H, w.fc2_w, H, w.fc2_b, 1.f, a.x, E, M, E, H, 0, 1, s)) != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_map_take_tail, dim3(a.Lc, N), dim3(256), 0, s, a.x, a.Lp, a.Lc, E, a.out);
  return hipGetLastError();
}

// prompt[n][p][:] for p < P = Lc + Lt: soft prompt rows (cont [N][Lc][E]) and word embeddings wte[tokens[n][t]] (tokens
// [N][Lt], already padded with the pad id like pad_sequence does), soft first or hard first (entrypoint.py:128-133)
__global__ __launch_bounds__(256) void k_build_prompt(const float* __restrict__ cont, const int32_t* __restrict__ tokens,
                                                      const float* __restrict__ wte, int Lc, int Lt, int E, int V, int soft_first,
                                                      float* prompt) {
  const int n = blockIdx.y, p = blockIdx.x, P = Lc + Lt;
  const bool soft = soft_first ? p < Lc : p >= Lt;
  const float* src;
  if (soft) {
    src = cont + ((size_t)n * Lc + (soft_first ? p : p - Lt)) * E;
  } else {
    int id = tokens[(size_t)n * Lt + (soft_first ? p - Lc : p)];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    src = wte + (size_t)id * E;
  }
  float* dst = prompt + ((size_t)n * P + p) * E;
  for (int d = threadIdx.x; d < E; d += 256) dst[d] = src[d];
}

hipError_t launch_build_prompt(const float* cont, const int32_t* tokens, const float* wte, int N, int Lc, int Lt, int E, int V,
                               int soft_first, float* prompt, hipStream_t s) {
  if (N < 1 || Lc + Lt < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_build_prompt, dim3(Lc + Lt, N), dim3(256), 0, s, cont, tokens, wte, Lc, Lt, E, V, soft_first, prompt);
  return hipGetLastError();
}

}  // namespace pio
