// DeCap / CapDec text decoder: greedy decode with a KV cache, exact fp32 (gfx950).
//
// Replaces decoding_batched (P/src/decap/decap.py:116-155): the reference re-encodes the whole growing
// sequence each step (465 token-forwards per caption, LM head on every position); here each step
// forwards ONE new position per prefix against cached keys/values -- the same function in exact
// arithmetic, verified id-for-id against the reference fixtures.  fp32 throughout because greedy ids
// must be bit-exact and top-2 logit margins of ~1e-4 occur (tests/golden/decoder.npz).
//
// At <= 64 prefixes every linear layer is a weight-streaming "skinny" GEMM (HBM / Infinity-Cache
// bound): k_dec_gemm streams W [out][in] once with 16-B loads straight to VGPRs and multiplies on
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain), 16 output columns per workgroup, K split over the 4
// waves and reduced through LDS.  The step's small kernels (LayerNorm, attention over <= 31 cached
// positions, arg-max + log-sum-exp + next embedding) are one wave or one workgroup per prefix.
// The whole 30-step loop is captured once into a hipGraph by api.cpp.
#include "common.h"
#include "kernels.h"

namespace pio {

enum DecEpi { DE_STORE = 0, DE_RESID = 1, DE_GELU = 2, DE_EMBED = 3, DE_LOGITS = 4 };

__device__ __forceinline__ f32x4 mfma16f(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_new(float x) {
  // transformers "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}

// out[n][j] (+)= sum_k X[n][k] * W[j][k]   (W: [Nout][K] row-major = [out][in])
template <int RG, int EPI>
__global__ __launch_bounds__(256) void k_dec_gemm(const float* __restrict__ W, const float* __restrict__ X, int N,
                                                  int Nout, int K, const float* __restrict__ bias, float* out,
                                                  const float* __restrict__ extra) {
  __shared__ __attribute__((aligned(16))) float red[4 * RG * 256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int col0 = blockIdx.x * 16;
  int col = col0 + li;
  col = col < Nout ? col : Nout - 1;
  const int kslice = K >> 2;
  const float* wp = W + (size_t)col * K + wid * kslice + 4 * kq;
  const float* xp[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    int n = g * 16 + li;
    n = n < N ? n : N - 1;
    xp[g] = X + (size_t)n * K + wid * kslice + 4 * kq;
  }
  f32x4 acc[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nchunk = kslice >> 4;
#pragma unroll 4
  for (int c = 0; c < nchunk; ++c) {
    const float4 w4 = *(const float4*)(wp + 16 * c);
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const float4 x4 = *(const float4*)(xp[g] + 16 * c);
      acc[g] = mfma16f(x4.x, w4.x, acc[g]);
      acc[g] = mfma16f(x4.y, w4.y, acc[g]);
      acc[g] = mfma16f(x4.z, w4.z, acc[g]);
      acc[g] = mfma16f(x4.w, w4.w, acc[g]);
    }
  }
#pragma unroll
  for (int g = 0; g < RG; ++g) *(f32x4*)(red + ((wid * RG + g) * 64 + lane) * 4) = acc[g];
  __syncthreads();
  for (int g = wid; g < RG; g += 4) {
    f32x4 s = *(const f32x4*)(red + ((0 * RG + g) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) s += *(const f32x4*)(red + ((w * RG + g) * 64 + lane) * 4);
    const int j = col0 + li;
    if (j >= Nout) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = g * 16 + 4 * kq + i;
      if (n >= N) continue;
      float v = s[i];
      float* o = out + (size_t)n * Nout + j;
      if constexpr (EPI == DE_STORE) *o = v + bias[j];
      else if constexpr (EPI == DE_RESID) *o += v + bias[j];
      else if constexpr (EPI == DE_GELU) *o = gelu_new(v + bias[j]);
      else if constexpr (EPI == DE_EMBED) *o = v + bias[j] + extra[j];
      else *o = v;
    }
  }
}

// GPT-2 LayerNorm (eps 1e-5), one workgroup per prefix row.
__global__ __launch_bounds__(256) void k_dec_layernorm(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float eps, int E, float* y) {
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* r = x + (size_t)n * E;
  float v[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + 256 * k;
    v[k] = d < E ? r[d] : 0.f;
    s += v[k];
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)E;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + 256 * k;
    if (d < E) { const float c = v[k] - mean; q += c * c; }
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[tid >> 6] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)E + eps);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + 256 * k;
    if (d < E) y[(size_t)n * E + d] = (v[k] - mean) * rstd * w[d] + b[d];
  }
}

// Causal attention for the new position `pos` of prefix n, head h: appends k,v to the cache and
// attends over positions 0..pos.  One wave per (n, head); head_dim = 192 = 3 per lane.
__global__ __launch_bounds__(64) void k_dec_attention(const float* __restrict__ qkv, float* kcache, float* vcache,
                                                      int E, int heads, int pos, int max_steps, float* att) {
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const int lane = threadIdx.x;
  const int hd = E / heads, per = hd >> 6;       // 192 / 64 = 3
  const float* q = qkv + (size_t)n * 3 * E + h * hd;
  const float* kn = q + E;
  const float* vn = q + 2 * E;
  float* kc = kcache + ((size_t)n * max_steps) * E + h * hd;
  float* vc = vcache + ((size_t)n * max_steps) * E + h * hd;
  float qv[4], o[4];
  for (int i = 0; i < per; ++i) {
    const int d = lane + 64 * i;
    qv[i] = q[d];
    kc[(size_t)pos * E + d] = kn[d];
    vc[(size_t)pos * E + d] = vn[d];
    o[i] = 0.f;
  }
  const float scale = 1.0f / sqrtf((float)hd);
  float mine = -INFINITY;                        // lane j keeps the score of cached position j (pos < 64)
  for (int j = 0; j <= pos; ++j) {
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
      const int d = lane + 64 * i;
      const float kv = (j == pos) ? kn[d] : kc[(size_t)j * E + d];
      s += qv[i] * kv;
    }
    s = wave_sum(s) * scale;
    if (lane == j) mine = s;
  }
  const float mx = wave_max(mine);
  const float e = lane <= pos ? expf(mine - mx) : 0.f;
  const float den = wave_sum(e);
  const float pmine = e / den;
  for (int j = 0; j <= pos; ++j) {
    const float p = __shfl(pmine, j);
    for (int i = 0; i < per; ++i) {
      const int d = lane + 64 * i;
      const float vv = (j == pos) ? vn[d] : vc[(size_t)j * E + d];
      o[i] += p * vv;
    }
  }
  for (int i = 0; i < per; ++i) att[(size_t)n * E + h * hd + lane + 64 * i] = o[i];
}

// arg-max (first index on ties, like torch.argmax) + log-softmax of the chosen logit + embedding of the
// next input (wte[id] + wpe[pos+1]).  One workgroup per prefix.
__global__ __launch_bounds__(256) void k_dec_select(const float* __restrict__ logits, int V, int E, int step, int steps,
                                                    const float* __restrict__ wte, const float* __restrict__ wpe,
                                                    int32_t* ids, float* logprob, float* x) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ float s_s[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* r = logits + (size_t)n * V;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < V; i += 256) {
    const float v = r[i];
    if (v > bv) { bv = v; bi = i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
  __syncthreads();
  bv = s_v[0]; bi = s_i[0];
  for (int w = 1; w < 4; ++w)
    if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
  float se = 0.f;
  for (int i = tid; i < V; i += 256) se += expf(r[i] - bv);
  se = wave_sum(se);
  if ((tid & 63) == 0) s_s[tid >> 6] = se;
  __syncthreads();
  se = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
  if (tid == 0) {
    ids[(size_t)n * steps + step] = bi;
    if (logprob != nullptr) logprob[(size_t)n * steps + step] = -logf(se);   // log softmax at the max logit
  }
  for (int d = tid; d < E; d += 256) x[(size_t)n * E + d] = wte[(size_t)bi * E + d] + wpe[(size_t)(step + 1) * E + d];
}

template <int EPI>
static hipError_t dec_gemm(const float* W, const float* X, int N, int Nout, int K, const float* bias, float* out,
                           const float* extra, hipStream_t s) {
  if (K % 64 != 0 || N < 1 || N > 128) return hipErrorInvalidValue;
  const dim3 grid(ceil_div(Nout, 16)), block(256);
  const int rg = ceil_div(N, 16);
  if (rg <= 1) hipLaunchKernelGGL((k_dec_gemm<1, EPI>), grid, block, 0, s, W, X, N, Nout, K, bias, out, extra);
  else if (rg <= 2) hipLaunchKernelGGL((k_dec_gemm<2, EPI>), grid, block, 0, s, W, X, N, Nout, K, bias, out, extra);
  else if (rg <= 4) hipLaunchKernelGGL((k_dec_gemm<4, EPI>), grid, block, 0, s, W, X, N, Nout, K, bias, out, extra);
  else hipLaunchKernelGGL((k_dec_gemm<8, EPI>), grid, block, 0, s, W, X, N, Nout, K, bias, out, extra);
  return hipGetLastError();
}

#define PIO_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

hipError_t launch_decode_greedy(const DecoderArgs& a, hipStream_t s) {
  const int N = a.N, E = a.E;
  if (a.steps > a.max_steps || a.steps > 64 || E > 1024 || (E / a.heads) % 64 != 0 || (E / a.heads) > 256)
    return hipErrorInvalidValue;
  // step 0 input: clip_project(prefix) + wpe[0]   (decap.py:124; wpe added inside GPT-2)
  PIO_TRY((dec_gemm<DE_EMBED>(a.clip_w, a.prefix, N, E, a.prefix_size, a.clip_b, a.x, a.wpe, s)));
  for (int step = 0; step < a.steps; ++step) {
    for (int l = 0; l < a.layers; ++l) {
      const DecLayerW& w = a.layer[l];
      float* kc = a.kcache + (size_t)l * N * a.max_steps * E;
      float* vc = a.vcache + (size_t)l * N * a.max_steps * E;
      hipLaunchKernelGGL(k_dec_layernorm, dim3(N), dim3(256), 0, s, a.x, w.ln1_w, w.ln1_b, a.eps, E, a.y);
      PIO_TRY((dec_gemm<DE_STORE>(w.attn_w, a.y, N, 3 * E, E, w.attn_b, a.qkv, nullptr, s)));
      hipLaunchKernelGGL(k_dec_attention, dim3(N * a.heads), dim3(64), 0, s, a.qkv, kc, vc, E, a.heads, step,
                         a.max_steps, a.att);
      PIO_TRY((dec_gemm<DE_RESID>(w.proj_w, a.att, N, E, E, w.proj_b, a.x, nullptr, s)));
      hipLaunchKernelGGL(k_dec_layernorm, dim3(N), dim3(256), 0, s, a.x, w.ln2_w, w.ln2_b, a.eps, E, a.y);
      PIO_TRY((dec_gemm<DE_GELU>(w.fc_w, a.y, N, 4 * E, E, w.fc_b, a.hid, nullptr, s)));
      PIO_TRY((dec_gemm<DE_RESID>(w.fc2_w, a.hid, N, E, 4 * E, w.fc2_b, a.x, nullptr, s)));
    }
    hipLaunchKernelGGL(k_dec_layernorm, dim3(N), dim3(256), 0, s, a.x, a.lnf_w, a.lnf_b, a.eps, E, a.y);
    PIO_TRY((dec_gemm<DE_LOGITS>(a.wte, a.y, N, a.vocab, E, nullptr, a.logits, nullptr, s)));
    hipLaunchKernelGGL(k_dec_select, dim3(N), dim3(256), 0, s, a.logits, a.vocab, E, step, a.steps, a.wte, a.wpe,
                       a.ids, a.logprob, a.x);
  }
  return hipGetLastError();
}

}  // namespace pio
