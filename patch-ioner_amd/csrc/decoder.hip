// DeCap / CapDec text decoder: greedy decode with a KV cache, exact fp32 (gfx950).
//
// Replaces decoding_batched (P/src/decap/decap.py:116-155): the reference re-encodes the whole growing
// sequence each step (465 token-forwards per caption, LM head on every position); here each step
// forwards ONE new position per prefix against cached keys/values -- the same function in exact
// arithmetic, verified id-for-id against the reference fixtures.  fp32 throughout because greedy ids
// must be bit-exact and top-2 logit margins of ~1e-4 occur (tests/golden/decoder.npz).
//
// A step is a chain of small dependent kernels (each ~4-5 us at <= 16 prefixes), so the design minimises
// their number: 22 per step (4 layers x 5 + LM head + select), the whole 30-step loop captured once into a
// hipGraph by api.cpp.
//   * every linear layer is a weight-streaming "skinny" GEMM on v_mfma_f32_16x16x4_f32 (an exact fp32 FMA
//     chain): one workgroup = 16 output columns; each wave owns 192 k and issues ALL its 16-B weight loads
//     up front (12 in flight per lane, straight to VGPRs); waves are reduced through LDS.
//   * LayerNorm is folded into the GEMM that consumes it:  LN(x) W^T + b = r (x W'^T - mu c) + d  with
//     W' = W * ln_w (per input channel), c_j = sum_k W'_jk, d_j = sum_k ln_b_k W_jk + b_j precomputed at load
//     time; mu, r come from the x values the waves load anyway (one LDS reduction), so no LayerNorm kernel
//     and no normalised copy of x exist.
//   * the GEMMs that end a residual branch (attn.c_proj, mlp.c_proj) cover the full K in one workgroup
//     (4 or 16 waves) and add bias + result into the residual stream in place (deterministic, no atomics).
//   * the LM head never materialises logits: each workgroup reduces its 16 columns to (max, arg-max,
//     sum-exp) per prefix and k_dec_select merges the 3142 partials into the id / log-prob and writes the
//     next input embedding.
#include "common.h"
#include "kernels.h"

namespace pio {

// diagnostic ablations (tools/microbench/dec_bench.hip); all off in the shipped library
#ifndef PIO_DABL_NOX
#define PIO_DABL_NOX 0
#endif
#ifndef PIO_DABL_NOMFMA
#define PIO_DABL_NOMFMA 0
#endif
#ifndef PIO_LMHEAD_CG
#define PIO_LMHEAD_CG 1
#endif

enum DecEpi { DE_STORE = 0, DE_RESID = 1, DE_GELU = 2, DE_EMBED = 3, DE_ARGMAX = 4 };

__device__ __forceinline__ f32x4 mfma16f(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_new(float x) {
  // transformers "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}

// out[n][j] = epilogue( sum_k X[n][k] * W[j][k] )        W [Nout][K] ([out][in]), X [N][K]
//   grid = ceil(Nout/16) workgroups of NW waves; K = NW * CPW * 16 (NW = 4: K = 768 or 512; NW = 16: K = 3072).
//   lane (li = lane&15, kq = lane>>4): B operand W[col0+li][k0 + 16c + 4kq + t], A operand X[16g+li][same k]
//   (the k order inside a chunk is free as long as A and B agree); C: column li, row 4kq+i.
//   LN != 0: X is the raw residual stream; the LayerNorm is applied algebraically in the epilogue
//            (W is pre-scaled by ln_w; cvec / dvec as in the file header).
template <int RG, int CPW, int NW, int EPI, int LN, int CG>
__global__ __launch_bounds__(64 * NW) void k_dec_gemm(const float* __restrict__ W, const float* __restrict__ X, int N,
                                                      int Nout, int K, const float* __restrict__ bias, float* out,
                                                      const float* __restrict__ extra, const float* __restrict__ cvec,
                                                      float eps) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][RG*CG][256]
  __shared__ float s_sum[LN ? NW : 1][RG * 16], s_sq[LN ? NW : 1][RG * 16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int col0 = blockIdx.x * (16 * CG);
  const int k0 = wid * (16 * CPW) + 4 * kq;
  float4 w4[CG][CPW];
#pragma unroll
  for (int q = 0; q < CG; ++q) {
    int col = col0 + 16 * q + li;
    col = col < Nout ? col : Nout - 1;
    const float* wp = W + (size_t)col * K + k0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) w4[q][c] = *(const float4*)(wp + 16 * c);   // the HBM stream: all in flight
  }
  __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the loads next to their MFMAs (2 in flight)
  f32x4 acc[RG][CG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    int n = g * 16 + li;
    n = n < N ? n : N - 1;
    const float* xp = X + (size_t)n * K + k0;
    float4 x4[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) x4[c] = PIO_DABL_NOX ? make_float4(0.5f, 0.25f, 1.f, 2.f) : *(const float4*)(xp + 16 * c);   // activations: L2-resident
    __builtin_amdgcn_sched_barrier(0);
    f32x4 a[CG];
#pragma unroll
    for (int q = 0; q < CG; ++q) a[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float sx = 0.f, sq = 0.f;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
#pragma unroll
      for (int q = 0; q < CG; ++q) {
        if (PIO_DABL_NOMFMA) { a[q][0] += x4[c].x * w4[q][c].x + x4[c].y * w4[q][c].y + x4[c].z * w4[q][c].z + x4[c].w * w4[q][c].w; continue; }
        a[q] = mfma16f(x4[c].x, w4[q][c].x, a[q]);
        a[q] = mfma16f(x4[c].y, w4[q][c].y, a[q]);
        a[q] = mfma16f(x4[c].z, w4[q][c].z, a[q]);
        a[q] = mfma16f(x4[c].w, w4[q][c].w, a[q]);
      }
      if (LN) {
        sx += (x4[c].x + x4[c].y) + (x4[c].z + x4[c].w);
        sq += (x4[c].x * x4[c].x + x4[c].y * x4[c].y) + (x4[c].z * x4[c].z + x4[c].w * x4[c].w);
      }
    }
#pragma unroll
    for (int q = 0; q < CG; ++q) acc[g][q] = a[q];
    if (LN) {   // row statistics of x: this lane holds 4*CPW values of row 16g+li; sum the 4 kq groups
      sx += __shfl_xor(sx, 16); sx += __shfl_xor(sx, 32);
      sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
      if (kq == 0) { s_sum[wid][g * 16 + li] = sx; s_sq[wid][g * 16 + li] = sq; }
    }
  }
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int q = 0; q < CG; ++q) *(f32x4*)(red + ((wid * RG * CG + g * CG + q) * 64 + lane) * 4) = acc[g][q];
  __syncthreads();
  for (int gq = wid; gq < RG * CG; gq += NW) {
    const int g = gq / CG, q = gq - g * CG;
    f32x4 s = *(const f32x4*)(red + ((0 * RG * CG + gq) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < NW; ++w) s += *(const f32x4*)(red + ((w * RG * CG + gq) * 64 + lane) * 4);
    const int j = col0 + 16 * q + li;
    const int jc = j < Nout ? j : Nout - 1;
    if (LN) {   // s_i <- r_n (s_i - mu_n c_j) + d_j  for row n = 16g + 4kq + i
      const float cj = cvec[jc], dj = bias[jc];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = g * 16 + 4 * kq + i;
        float tx = 0.f, tq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { tx += s_sum[w][rr]; tq += s_sq[w][rr]; }
        const float mu = tx / (float)K;
        const float var = fmaxf(tq / (float)K - mu * mu, 0.f);
        s[i] = rsqrtf(var + eps) * (s[i] - mu * cj) + dj;
      }
    }
    if constexpr (EPI == DE_ARGMAX) {
      // per prefix: (max, arg-max, sum exp(logit - max)) over this workgroup's 16 columns
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = j < Nout ? s[i] : -INFINITY;
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
          float* p = out + ((size_t)(blockIdx.x * CG + q) * N + n) * 4;
          p[0] = mx; p[1] = __int_as_float(idx); p[2] = se;
        }
      }
    } else {
      if (j >= Nout) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = g * 16 + 4 * kq + i;
        if (n >= N) continue;
        const float v = LN ? s[i] : s[i] + bias[j];     // LN: the bias is already inside d_j
        float* o = out + (size_t)n * Nout + j;
        if constexpr (EPI == DE_STORE) *o = v;
        else if constexpr (EPI == DE_RESID) *o += v;    // each element has exactly one owner: in place is safe
        else if constexpr (EPI == DE_GELU) *o = gelu_new(v);
        else if constexpr (EPI == DE_EMBED) *o = v + extra[j];
      }
    }
  }
}

// Causal attention for the new position `pos` of prefix n, head h: appends k,v to the cache and attends
// over positions 0..pos (pos < 64).  One workgroup per (n, head).
//   scores : thread (j = tid>>3, seg = tid&7) takes hd/8 channels of key j -> all key loads of a 32-key
//            pass are in flight at once; 3 shuffles finish the dot product.
//   output : thread (c4 = tid % (hd/4), jg = tid / (hd/4)) accumulates 4 channels over keys j = jg (mod NJ);
//            the NJ partial sums meet in LDS.
__global__ __launch_bounds__(256) void k_dec_attention(const float* __restrict__ qkv, float* kcache, float* vcache,
                                                       int E, int heads, int pos, int max_steps, float* att) {
  __shared__ float s_sc[64];
  __shared__ __attribute__((aligned(16))) float s_o[5][256];
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const int tid = threadIdx.x;
  const int hd = E / heads;                      // 192; multiple of 32
  const float* q = qkv + (size_t)n * 3 * E + h * hd;
  const float* kn = q + E;
  const float* vn = q + 2 * E;
  float* kc = kcache + ((size_t)n * max_steps) * E + h * hd;
  float* vc = vcache + ((size_t)n * max_steps) * E + h * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  {
    const int seg = tid & 7, jl = tid >> 3, per4 = hd >> 5;   // hd/8 channels = per4 float4 (6 for 192)
    for (int j0 = 0; j0 <= pos; j0 += 32) {
      const int j = j0 + jl;
      float s = 0.f;
      if (j <= pos) {
        const float4* kr = (const float4*)((j == pos ? kn : kc + (size_t)j * E) + seg * (hd >> 3));
        const float4* qr = (const float4*)(q + seg * (hd >> 3));
        for (int i = 0; i < per4; ++i) {
          const float4 a = kr[i], b = qr[i];
          s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        }
      }
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      if (seg == 0 && j <= pos) s_sc[j] = s * scale;
    }
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
  const int nc4 = hd >> 2;                          // 48 float4 per head row
  const int NJ = (256 / nc4) < 5 ? (256 / nc4) : 5;   // 5 key groups (240 threads active)
  const int c4 = tid % nc4, jg = tid / nc4;
  if (jg < NJ) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int j = jg; j <= pos; j += NJ) {
      const float p = expf(s_sc[j] - mx) * inv;
      const float4 v = ((const float4*)(j == pos ? vn : vc + (size_t)j * E))[c4];
      o.x += p * v.x; o.y += p * v.y; o.z += p * v.z; o.w += p * v.w;
    }
    *(float4*)&s_o[jg][4 * c4] = o;
  }
  __syncthreads();
  if (tid < hd) {
    float o = 0.f;
    for (int g = 0; g < NJ; ++g) o += s_o[g][tid];
    att[(size_t)n * E + h * hd + tid] = o;
  }
}

// Merge the LM head's per-workgroup (max, arg-max, sum-exp) partials: greedy id (first index on ties, like
// torch.argmax), log-softmax of the chosen logit; then the next step's input x = wte[id] + wpe[step+1].
// One workgroup per prefix.
__global__ __launch_bounds__(256) void k_dec_select(const float* __restrict__ part, int nblk, int N, int E, int step,
                                                    int steps, const float* __restrict__ wte,
                                                    const float* __restrict__ wpe, int32_t* ids, float* logprob,
                                                    float* x) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ float s_s[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // each thread keeps its <= 16 partials (nblk <= 4096) in registers: one 16-B load each, all in flight
  float4 pr[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int b = tid + 256 * k;
    pr[k] = b < nblk ? *(const float4*)(part + ((size_t)b * N + n) * 4) : make_float4(-INFINITY, __int_as_float(0x7fffffff), 0.f, 0.f);
  }
  float bv = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float v = pr[k].x;
    const int i = __float_as_int(pr[k].y);
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
  for (int d = tid; d < E; d += 256) x[(size_t)n * E + d] = wte[(size_t)bi * E + d] + wpe[(size_t)(step + 1) * E + d];
  {
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) se += pr[k].z * expf(pr[k].x - bv);   // padded entries: 0 * exp(-inf) = 0
    se = wave_sum(se);
    if (lane == 0) s_s[wid] = se;
  }
  __syncthreads();
  if (tid == 0) {
    ids[(size_t)n * steps + step] = bi;
    if (logprob != nullptr) logprob[(size_t)n * steps + step] = -logf((s_s[0] + s_s[1]) + (s_s[2] + s_s[3]));
  }
}

template <int CPW, int NW, int EPI, int LN, int CG>
static hipError_t dec_gemm_rg(const float* W, const float* X, int N, int Nout, int K, const float* bias, float* out,
                              const float* extra, const float* cvec, float eps, hipStream_t s) {
  const dim3 grid(ceil_div(Nout, 16 * CG)), block(64 * NW);
  const int rg = ceil_div(N, 16);
#define PIO_DG(R) hipLaunchKernelGGL((k_dec_gemm<R, CPW, NW, EPI, LN, CG>), grid, block, NW * R * CG * 1024, s, W, X, N, Nout, K, bias, out, extra, cvec, eps)
  if (rg <= 1) PIO_DG(1);
  else if (rg <= 2) PIO_DG(2);
  else if (rg <= 4) PIO_DG(4);
  else if (NW == 4 && CG == 1) PIO_DG(8);
  else return hipErrorInvalidValue;      // 16-wave / 2-column-group workgroups are built for <= 64 prefixes
#undef PIO_DG
  return hipGetLastError();
}

// K = 768 (4 waves x 12 chunks), 512 (4 x 8) or 3072 (16 x 12)
template <int EPI, int LN>
static hipError_t dec_gemm(const float* W, const float* X, int N, int Nout, int K, const float* bias, float* out,
                           const float* extra, const float* cvec, float eps, hipStream_t s) {
  if (N < 1 || N > 128) return hipErrorInvalidValue;
  constexpr int CG = EPI == DE_ARGMAX ? PIO_LMHEAD_CG : 1;   // LM head: 32 columns per workgroup halve the re-reads of x
  if (K == 768) return dec_gemm_rg<12, 4, EPI, LN, CG>(W, X, N, Nout, K, bias, out, extra, cvec, eps, s);
  if (K == 512) return dec_gemm_rg<8, 4, EPI, LN, CG>(W, X, N, Nout, K, bias, out, extra, cvec, eps, s);
  if (K == 3072) return dec_gemm_rg<12, 16, EPI, LN, CG>(W, X, N, Nout, K, bias, out, extra, cvec, eps, s);
  return hipErrorInvalidValue;
}

hipError_t decoder_init() {
  // the 16-wave, 4-row-group instantiation needs 64 KiB of dynamic LDS
  return hipFuncSetAttribute((const void*)k_dec_gemm<4, 12, 16, DE_RESID, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             16 * 4 * 1024);
}

#define PIO_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

// LM head -> greedy partials: one (max, arg-max, sum-exp) per prefix and 16-column group.
// (A persistent variant that keeps x in registers and walks 5-7 column groups per workgroup measured
//  50 us against 41 us for this one at 16 prefixes: fewer bytes in flight per CU and a serial per-group
//  epilogue; dropped.)
hipError_t launch_lmhead(const float* W, const float* X, int N, int V, int E, const float* dvec, const float* cvec,
                         float eps, float* part, int* nblk, hipStream_t s) {
  *nblk = ceil_div(V, 16);
  return dec_gemm<DE_ARGMAX, 1>(W, X, N, V, E, dvec, part, nullptr, cvec, eps, s);
}

hipError_t launch_decode_greedy(const DecoderArgs& a, hipStream_t s) {
  const int N = a.N, E = a.E;
  if (a.steps > a.max_steps || a.steps > 64 || E != 768 || (E / a.heads) % 32 != 0 || (E / a.heads) > 256 || N > 64)
    return hipErrorInvalidValue;
  if (ceil_div(a.vocab, 16) > 4096) return hipErrorInvalidValue;
  // step 0 input: clip_project(prefix) + wpe[0]   (decap.py:124; GPT-2 adds wpe to inputs_embeds)
  PIO_TRY((dec_gemm<DE_EMBED, 0>(a.clip_w, a.prefix, N, E, a.prefix_size, a.clip_b, a.x, a.wpe, nullptr, 0.f, s)));
  for (int step = 0; step < a.steps; ++step) {
    for (int l = 0; l < a.layers; ++l) {
      const DecLayerW& w = a.layer[l];
      float* kc = a.kcache + (size_t)l * N * a.max_steps * E;
      float* vc = a.vcache + (size_t)l * N * a.max_steps * E;
      PIO_TRY((dec_gemm<DE_STORE, 1>(w.attn_w, a.x, N, 3 * E, E, w.attn_d, a.qkv, nullptr, w.attn_c, a.eps, s)));
      hipLaunchKernelGGL(k_dec_attention, dim3(N * a.heads), dim3(256), 0, s, a.qkv, kc, vc, E, a.heads, step,
                         a.max_steps, a.att);
      PIO_TRY((dec_gemm<DE_RESID, 0>(w.proj_w, a.att, N, E, E, w.proj_b, a.x, nullptr, nullptr, 0.f, s)));
      PIO_TRY((dec_gemm<DE_GELU, 1>(w.fc_w, a.x, N, 4 * E, E, w.fc_d, a.hid, nullptr, w.fc_c, a.eps, s)));
      PIO_TRY((dec_gemm<DE_RESID, 0>(w.fc2_w, a.hid, N, E, 4 * E, w.fc2_b, a.x, nullptr, nullptr, 0.f, s)));
    }
    int nblk = 0;
    PIO_TRY(launch_lmhead(a.head_w, a.x, N, a.vocab, E, a.head_d, a.head_c, a.eps, a.logits, &nblk, s));
    hipLaunchKernelGGL(k_dec_select, dim3(N), dim3(256), 0, s, a.logits, nblk, N, E, step, a.steps, a.wte, a.wpe,
                       a.ids, a.logprob, a.x);
  }
  return hipGetLastError();
}

}  // namespace pio
