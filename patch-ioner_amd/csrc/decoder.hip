// DeCap / CapDec text decoder: greedy decode with a KV cache, exact fp32 (gfx950).
//
// Replaces decoding_batched (P/src/decap/decap.py:116-155): the reference re-encodes the whole growing
// sequence each step (465 token-forwards per caption, LM head on every position); here each step
// forwards ONE new position per prefix against cached keys/values -- the same function in exact
// arithmetic, verified id-for-id against the reference fixtures.  fp32 throughout because greedy ids
// must be bit-exact and top-2 logit margins of ~1e-4 occur (tests/golden/decoder.npz).
//
// At <= 128 prefixes every linear layer is a weight-streaming "skinny" GEMM (HBM / Infinity-Cache
// bound).  k_dec_gemm: one workgroup = 16 output columns x 768 k (4 waves x 192 k); every wave issues
// ALL of its 16-B weight loads up front (12 in flight per lane, straight to VGPRs, no LDS round trip) and
// multiplies on v_mfma_f32_16x16x4_f32 (an exact fp32 FMA chain).  K > 768 is split over workgroups
// (grid.y) which write partial sums; the consumer of a residual branch is k_dec_add_ln, which adds bias +
// partials into the residual stream and emits the next LayerNorm in the same pass (deterministic: no
// atomics).  The LM head never materialises logits: each workgroup reduces its 16 columns to
// (max, arg-max, sum-exp) per prefix and k_dec_select merges the 3142 partials, writes the id / log-prob
// and the next step's embedding + first LayerNorm.  30 kernels per step, captured once into a hipGraph.
#include "common.h"
#include "kernels.h"

namespace pio {

enum DecEpi { DE_STORE = 0, DE_PARTIAL = 1, DE_GELU = 2, DE_EMBED = 3, DE_ARGMAX = 4 };

__device__ __forceinline__ f32x4 mfma16f(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_new(float x) {
  // transformers "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}

// Partial / full product  out[n][j] = sum_{k in this workgroup's slice} X[n][k] * W[j][k].
//   W [Nout][K] row-major ([out][in]); X [N][K]; grid = (ceil(Nout/16), K / (64*CPW)); CPW = 16-k chunks per wave.
//   lane (li = lane&15, kq = lane>>4): B operand W[col0+li][k0 + 16c + 4kq + t], A operand X[n = 16g+li][same k]
//   (the k order inside a chunk is free as long as A and B agree); C: column li, row 4kq+i.
template <int RG, int CPW, int EPI>
__global__ __launch_bounds__(256) void k_dec_gemm(const float* __restrict__ W, const float* __restrict__ X, int N,
                                                  int Nout, int K, const float* __restrict__ bias, float* out,
                                                  const float* __restrict__ extra) {
  __shared__ __attribute__((aligned(16))) float red[4 * RG * 256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int col0 = blockIdx.x * 16;
  int col = col0 + li;
  col = col < Nout ? col : Nout - 1;
  const int k0 = (blockIdx.y * 4 + wid) * (16 * CPW) + 4 * kq;
  const float* wp = W + (size_t)col * K + k0;
  float4 w4[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) w4[c] = *(const float4*)(wp + 16 * c);     // the HBM stream: all in flight
  __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the loads next to their MFMAs (2 in flight)
  f32x4 acc[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    int n = g * 16 + li;
    n = n < N ? n : N - 1;
    const float* xp = X + (size_t)n * K + k0;
    float4 x4[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) x4[c] = *(const float4*)(xp + 16 * c);   // activations: L2-resident
    __builtin_amdgcn_sched_barrier(0);
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      a = mfma16f(x4[c].x, w4[c].x, a);
      a = mfma16f(x4[c].y, w4[c].y, a);
      a = mfma16f(x4[c].z, w4[c].z, a);
      a = mfma16f(x4[c].w, w4[c].w, a);
    }
    acc[g] = a;
  }
#pragma unroll
  for (int g = 0; g < RG; ++g) *(f32x4*)(red + ((wid * RG + g) * 64 + lane) * 4) = acc[g];
  __syncthreads();
  for (int g = wid; g < RG; g += 4) {
    f32x4 s = *(const f32x4*)(red + ((0 * RG + g) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) s += *(const f32x4*)(red + ((w * RG + g) * 64 + lane) * 4);
    const int j = col0 + li;
    if constexpr (EPI == DE_ARGMAX) {
      // per prefix: (max, arg-max, sum exp(logit - max)) over this workgroup's 16 columns
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = j < Nout ? s[i] : -INFINITY;
        int idx = j;
        float mx = v;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const float ov = __shfl_xor(mx, o);
          const int oi = __shfl_xor(idx, o);
          if (ov > mx || (ov == mx && oi < idx)) { mx = ov; idx = oi; }
        }
        float se = j < Nout ? expf(v - mx) : 0.f;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) se += __shfl_xor(se, o);
        const int n = g * 16 + 4 * kq + i;
        if (li == 0 && n < N) {
          float* p = out + ((size_t)blockIdx.x * N + n) * 4;
          p[0] = mx; p[1] = __int_as_float(idx); p[2] = se;
        }
      }
    } else {
      if (j >= Nout) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = g * 16 + 4 * kq + i;
        if (n >= N) continue;
        const float v = s[i];
        if constexpr (EPI == DE_STORE) out[(size_t)n * Nout + j] = v + bias[j];
        else if constexpr (EPI == DE_PARTIAL) out[((size_t)blockIdx.y * N + n) * Nout + j] = v;
        else if constexpr (EPI == DE_GELU) out[(size_t)n * Nout + j] = gelu_new(v + bias[j]);
        else if constexpr (EPI == DE_EMBED) out[(size_t)n * Nout + j] = v + bias[j] + extra[j];
      }
    }
  }
}

// x[n] += bias + sum_s part[s][n]   (bias == nullptr && nsplit == 0: x unchanged)
// y[n]  = LayerNorm(x[n]; w, b, eps)                    GPT-2 ln_1 / ln_2 / ln_f, eps 1e-5
// One wave per prefix row, E/64 <= 16 values per lane, no LDS, no barrier.
__global__ __launch_bounds__(64) void k_dec_add_ln(float* x, const float* __restrict__ part, int nsplit, int N,
                                                   const float* __restrict__ bias, const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps, int E, float* y) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const int nv = E >> 2;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    if (c < nv) {
      float4 a = ((const float4*)(x + (size_t)n * E))[c];
      if (bias != nullptr) {
        const float4 bb = ((const float4*)bias)[c];
        a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w;
      }
      for (int sp = 0; sp < nsplit; ++sp) {
        const float4 p = ((const float4*)(part + ((size_t)sp * N + n) * E))[c];
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
      }
      if (bias != nullptr || nsplit > 0) ((float4*)(x + (size_t)n * E))[c] = a;
      v[k] = a;
      s += (a.x + a.y) + (a.z + a.w);
    }
  }
  const float mean = wave_sum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    if (c < nv) {
      const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)E + eps);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    if (c < nv) {
      const float4 ww = ((const float4*)w)[c], bb = ((const float4*)b)[c];
      ((float4*)(y + (size_t)n * E))[c] =
          make_float4((v[k].x - mean) * rstd * ww.x + bb.x, (v[k].y - mean) * rstd * ww.y + bb.y,
                      (v[k].z - mean) * rstd * ww.z + bb.z, (v[k].w - mean) * rstd * ww.w + bb.w);
    }
  }
}

// Causal attention for the new position `pos` of prefix n, head h: appends k,v to the cache and attends
// over positions 0..pos.  One workgroup per (n, head): wave w scores positions w, w+4, ... (independent
// loads, all in flight), every thread then normalises the <= 64 scores from LDS, thread d accumulates
// output channel d over the cached values (coalesced rows).
__global__ __launch_bounds__(256) void k_dec_attention(const float* __restrict__ qkv, float* kcache, float* vcache,
                                                       int E, int heads, int pos, int max_steps, float* att) {
  __shared__ float s_sc[64];
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int hd = E / heads, per = hd >> 6;       // 192 / 64 = 3 (per <= 4)
  const float* q = qkv + (size_t)n * 3 * E + h * hd;
  const float* kn = q + E;
  const float* vn = q + 2 * E;
  float* kc = kcache + ((size_t)n * max_steps) * E + h * hd;
  float* vc = vcache + ((size_t)n * max_steps) * E + h * hd;
  float qv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) qv[i] = i < per ? q[lane + 64 * i] : 0.f;
  const float scale = 1.0f / sqrtf((float)hd);
  for (int j = wid; j <= pos; j += 4) {
    const float* kr = (j == pos) ? kn : kc + (size_t)j * E;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < per) s += qv[i] * kr[lane + 64 * i];
    s = wave_sum(s) * scale;
    if (lane == 0) s_sc[j] = s;
  }
  if (tid < hd) {                                  // append the new key / value
    kc[(size_t)pos * E + tid] = kn[tid];
    vc[(size_t)pos * E + tid] = vn[tid];
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int j = 0; j <= pos; ++j) mx = fmaxf(mx, s_sc[j]);
  float den = 0.f;
  for (int j = 0; j <= pos; ++j) den += expf(s_sc[j] - mx);
  const float inv = 1.0f / den;
  if (tid < hd) {
    float o = 0.f;
#pragma unroll 8
    for (int j = 0; j < pos; ++j) o += (expf(s_sc[j] - mx) * inv) * vc[(size_t)j * E + tid];
    o += (expf(s_sc[pos] - mx) * inv) * vn[tid];
    att[(size_t)n * E + h * hd + tid] = o;
  }
}

// Merge the LM head's per-workgroup (max, arg-max, sum-exp) partials: greedy id (first index on ties, like
// torch.argmax), log-softmax of the chosen logit; then the next step's input x = wte[id] + wpe[step+1] and
// its first LayerNorm y = ln_1^{(0)}(x).  One workgroup per prefix.
__global__ __launch_bounds__(256) void k_dec_select(const float* __restrict__ part, int nblk, int N, int E, int step,
                                                    int steps, const float* __restrict__ wte,
                                                    const float* __restrict__ wpe, const float* __restrict__ lnw,
                                                    const float* __restrict__ lnb, float eps, int32_t* ids,
                                                    float* logprob, float* x, float* y) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ float s_s[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int b = tid; b < nblk; b += 256) {
    const float* p = part + ((size_t)b * N + n) * 4;
    const float v = p[0];
    const int i = __float_as_int(p[1]);
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_v[wid] = bv; s_i[wid] = bi; }
  __syncthreads();
  bv = s_v[0]; bi = s_i[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
  float se = 0.f;
  for (int b = tid; b < nblk; b += 256) {
    const float* p = part + ((size_t)b * N + n) * 4;
    se += p[2] * expf(p[0] - bv);
  }
  se = wave_sum(se);
  if (lane == 0) s_s[wid] = se;
  __syncthreads();
  se = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
  if (tid == 0) {
    ids[(size_t)n * steps + step] = bi;
    if (logprob != nullptr) logprob[(size_t)n * steps + step] = -logf(se);
  }
  // next input row and its LayerNorm (E <= 1024: 4 values per thread)
  float v[4];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + 256 * k;
    v[k] = d < E ? wte[(size_t)bi * E + d] + wpe[(size_t)(step + 1) * E + d] : 0.f;
    if (d < E) x[(size_t)n * E + d] = v[k];
    sum += v[k];
  }
  __syncthreads();
  sum = wave_sum(sum);
  if (lane == 0) s_s[wid] = sum;
  __syncthreads();
  const float mean = ((s_s[0] + s_s[1]) + (s_s[2] + s_s[3])) / (float)E;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + 256 * k;
    if (d < E) { const float c = v[k] - mean; q += c * c; }
  }
  q = wave_sum(q);
  if (lane == 0) s_s[wid] = q;
  __syncthreads();
  const float rstd = rsqrtf(((s_s[0] + s_s[1]) + (s_s[2] + s_s[3])) / (float)E + eps);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = tid + 256 * k;
    if (d < E) y[(size_t)n * E + d] = (v[k] - mean) * rstd * lnw[d] + lnb[d];
  }
}

template <int CPW, int EPI>
static hipError_t dec_gemm_rg(const float* W, const float* X, int N, int Nout, int K, const float* bias, float* out,
                              const float* extra, hipStream_t s) {
  const dim3 grid(ceil_div(Nout, 16), K / (64 * CPW)), block(256);
  const int rg = ceil_div(N, 16);
  if (rg <= 1) hipLaunchKernelGGL((k_dec_gemm<1, CPW, EPI>), grid, block, 0, s, W, X, N, Nout, K, bias, out, extra);
  else if (rg <= 2) hipLaunchKernelGGL((k_dec_gemm<2, CPW, EPI>), grid, block, 0, s, W, X, N, Nout, K, bias, out, extra);
  else if (rg <= 4) hipLaunchKernelGGL((k_dec_gemm<4, CPW, EPI>), grid, block, 0, s, W, X, N, Nout, K, bias, out, extra);
  else hipLaunchKernelGGL((k_dec_gemm<8, CPW, EPI>), grid, block, 0, s, W, X, N, Nout, K, bias, out, extra);
  return hipGetLastError();
}

// K must be a multiple of 768 (12 chunks per wave) or, failing that, of 512 (8 chunks per wave).
template <int EPI>
static hipError_t dec_gemm(const float* W, const float* X, int N, int Nout, int K, const float* bias, float* out,
                           const float* extra, hipStream_t s) {
  if (N < 1 || N > 128) return hipErrorInvalidValue;
  if (K % 768 == 0) {
    if (EPI != DE_PARTIAL && K != 768) return hipErrorInvalidValue;   // only partial sums may split K
    return dec_gemm_rg<12, EPI>(W, X, N, Nout, K, bias, out, extra, s);
  }
  if (K % 512 == 0) {
    if (EPI != DE_PARTIAL && K != 512) return hipErrorInvalidValue;
    return dec_gemm_rg<8, EPI>(W, X, N, Nout, K, bias, out, extra, s);
  }
  return hipErrorInvalidValue;
}

#define PIO_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

int decoder_ksplit(int K) { return K % 768 == 0 ? K / 768 : K / 512; }

hipError_t launch_decode_greedy(const DecoderArgs& a, hipStream_t s) {
  const int N = a.N, E = a.E;
  if (a.steps > a.max_steps || a.steps > 64 || E > 1024 || E % 256 != 0 || (E / a.heads) % 64 != 0 ||
      (E / a.heads) > 256)
    return hipErrorInvalidValue;
  const int nblk = ceil_div(a.vocab, 16);
  const int ks_fc2 = decoder_ksplit(4 * E), ks_proj = decoder_ksplit(E);
  // step 0 input: clip_project(prefix) + wpe[0] (decap.py:124; GPT-2 adds wpe), then ln_1 of layer 0
  PIO_TRY((dec_gemm<DE_EMBED>(a.clip_w, a.prefix, N, E, a.prefix_size, a.clip_b, a.x, a.wpe, s)));
  hipLaunchKernelGGL(k_dec_add_ln, dim3(N), dim3(64), 0, s, a.x, (const float*)nullptr, 0, N, (const float*)nullptr,
                     a.layer[0].ln1_w, a.layer[0].ln1_b, a.eps, E, a.y);
  for (int step = 0; step < a.steps; ++step) {
    for (int l = 0; l < a.layers; ++l) {
      const DecLayerW& w = a.layer[l];
      float* kc = a.kcache + (size_t)l * N * a.max_steps * E;
      float* vc = a.vcache + (size_t)l * N * a.max_steps * E;
      PIO_TRY((dec_gemm<DE_STORE>(w.attn_w, a.y, N, 3 * E, E, w.attn_b, a.qkv, nullptr, s)));
      hipLaunchKernelGGL(k_dec_attention, dim3(N * a.heads), dim3(256), 0, s, a.qkv, kc, vc, E, a.heads, step,
                         a.max_steps, a.att);
      PIO_TRY((dec_gemm<DE_PARTIAL>(w.proj_w, a.att, N, E, E, nullptr, a.part, nullptr, s)));
      hipLaunchKernelGGL(k_dec_add_ln, dim3(N), dim3(64), 0, s, a.x, a.part, ks_proj, N, w.proj_b, w.ln2_w, w.ln2_b,
                         a.eps, E, a.y);
      PIO_TRY((dec_gemm<DE_GELU>(w.fc_w, a.y, N, 4 * E, E, w.fc_b, a.hid, nullptr, s)));
      PIO_TRY((dec_gemm<DE_PARTIAL>(w.fc2_w, a.hid, N, E, 4 * E, nullptr, a.part, nullptr, s)));
      const float* nw = (l + 1 < a.layers) ? a.layer[l + 1].ln1_w : a.lnf_w;
      const float* nb = (l + 1 < a.layers) ? a.layer[l + 1].ln1_b : a.lnf_b;
      hipLaunchKernelGGL(k_dec_add_ln, dim3(N), dim3(64), 0, s, a.x, a.part, ks_fc2, N, w.fc2_b, nw, nb, a.eps, E, a.y);
    }
    PIO_TRY((dec_gemm<DE_ARGMAX>(a.wte, a.y, N, a.vocab, E, nullptr, a.logits, nullptr, s)));
    hipLaunchKernelGGL(k_dec_select, dim3(N), dim3(256), 0, s, a.logits, nblk, N, E, step, a.steps, a.wte, a.wpe,
                       a.layer[0].ln1_w, a.layer[0].ln1_b, a.eps, a.ids, a.logprob, a.x, a.y);
  }
  return hipGetLastError();
}

}  // namespace pio
