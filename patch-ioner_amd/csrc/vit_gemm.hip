// ViT linear layers as one MFMA GEMM with fused epilogues (gfx950).
//
//   C[M][N] = A[M][K] * W[N][K]^T        A, W in fp16/bf16 (K contiguous for both: torch Linear layout),
//                                         fp32 accumulation on v_mfma_f32_32x32x16_{f16,bf16}.
//
// Replaces the stock ATen calls behind DINOv2's PatchEmbed conv, attn.qkv, attn.proj, mlp.fc1, mlp.fc2
// (reached from P/src/model.py:783) and fuses what followed them in the reference: bias, position
// embedding add, q/k/v head split (+ the fp32 capture the qkv forward hook takes,
// P/src/dino_extraction.py:7-9), LayerScale + residual add, exact-erf GELU.
//
// Tiling: BM x 128 x 64 per 256-thread workgroup, BM = 128 (4 waves as 2x2, each wave 64x64 = 2x2 MFMA
// tiles) for the wide GEMMs, BM = 64 (each wave 32x64) for the N = D ones so that they still give >= 1.5
// workgroups per CU at 16 images.  Operand tiles are register-staged into double-buffered LDS with a
// two-K-tile prefetch distance (two named register sets), one barrier per K-tile.  LDS rows are 128 B
// (64 halfs); the 16-B chunk index is XORed with (row>>1)&7 so that the ds_read_b128 fragment reads of a
// 16-lane group fall on 16 distinct 16-B slots of the 256-B bank row (conflict-free), while the staging
// ds_write_b128 of 8 consecutive lanes covers one whole row half.  The epilogue passes the accumulators
// through LDS so that every global access is a row-major 16 B (fp32) / 8 B (half) per lane.
// Workgroup ids are remapped so that each XCD's L2 sees a contiguous run of tiles sharing A panels.
#include "common.h"
#include "kernels.h"

namespace pio {

static constexpr int BN = 128, BK = 64;
// diagnostic ablations (tools/microbench/gemm_ablate.hip); all 0 in the shipped library
#ifndef PIO_ABL_NOGLOAD
#define PIO_ABL_NOGLOAD 0
#endif
#ifndef PIO_ABL_NOEPI
#define PIO_ABL_NOEPI 0
#endif
#ifndef PIO_ABL_NOMFMA
#define PIO_ABL_NOMFMA 0
#endif
#ifndef PIO_ABL_NOLDSW
#define PIO_ABL_NOLDSW 0
#endif
#ifndef PIO_GEMM_BM_NARROW      // tile height used when N == D (proj, fc2, patch embed)
#define PIO_GEMM_BM_NARROW 64
#endif

// erf to 1.5e-7 absolute (Abramowitz-Stegun 7.1.26): one v_rcp + one v_exp + 7 FMAs instead of libdevice's
// branchy erff; the result is rounded to fp16/bf16 anyway.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float r = 1.0f - poly * __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_fast(v * 0.70710678118654752440f)); }

template <typename T>
__device__ __forceinline__ void store_half4(T* dst, float a, float b, float c, float d) {
  typedef T half4_t __attribute__((ext_vector_type(4)));
  half4_t o;
  o[0] = (T)a; o[1] = (T)b; o[2] = (T)c; o[3] = (T)d;
  *(half4_t*)dst = o;
}

template <typename T, int EPI, int BM>
__global__ __launch_bounds__(256, 2) void k_vit_gemm(const GemmArgs g) {
  constexpr int MI = BM / 64;                      // 32-row MFMA tiles per wave along M (waves are 2 x 2)
  constexpr int NA = BM / 32;                      // 16-B A chunks per thread per K-tile (W: always 4)
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x STAGE; reused as fp32 [BM][128] by the epilogue
  typedef typename Vec8<T>::type frag_t;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, h = lane >> 5, r31 = lane & 31;

  const int ntn = g.N / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % ntn, tm = bid / ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging assignment: rows row0 + 32*i, 16-B chunk kc of the 128-B row ----
  const int kc = tid & 7;
  const int row0 = tid >> 3;
  const T* a_src[NA];
  const T* w_src[4];
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + 32 * i;
    w_src[i] = (const T*)g.W + (size_t)(n0 + row) * g.K + kc * 8;
    lds_off[i] = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int am = m0 + row0 + 32 * i;
    am = am < g.M ? am : g.M - 1;   // clamp: rows past M are computed on a copy of the last row, never stored
    a_src[i] = (const T*)g.A + (size_t)am * g.lda + kc * 8;
  }
  // ---- fragment read addresses ----
  const int sw7 = (lane >> 1) & 7;  // == ((row>>1)&7) for row = 32*x + (lane&31)
  int a_rd[MI], w_rd[2];
#pragma unroll
  for (int i = 0; i < MI; ++i) a_rd[i] = (wm * (BM / 2) + i * 32 + r31) * 128;
#pragma unroll
  for (int i = 0; i < 2; ++i) w_rd[i] = (wn * 64 + i * 32 + r31) * 128;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two named staging register sets (p, q): the loads of tile kt+2 are issued while tile kt is multiplied and
  // tile kt+1 (already in flight for a whole iteration) is written to the other LDS buffer, so global / L2
  // latency has two K-tiles of MFMA work to hide behind.  (Plain named values: staged arrays captured by
  // reference end up in scratch with hipcc.)
  uint4 pa0 = {}, pa1 = {}, pa2 = {}, pa3 = {}, pw0 = {}, pw1 = {}, pw2 = {}, pw3 = {};
  uint4 qa0 = {}, qa1 = {}, qa2 = {}, qa3 = {}, qw0 = {}, qw1 = {}, qw2 = {}, qw3 = {};
#define PIO_LOAD_TILE(S, kt)                                            \
  do {                                                                  \
    const int _ko = (kt) * BK;                                          \
    if (PIO_ABL_NOGLOAD) break;                                         \
    S##a0 = *(const uint4*)(a_src[0] + _ko);                            \
    S##a1 = *(const uint4*)(a_src[1] + _ko);                            \
    if constexpr (NA > 2) S##a2 = *(const uint4*)(a_src[NA > 2 ? 2 : 0] + _ko);   \
    if constexpr (NA > 2) S##a3 = *(const uint4*)(a_src[NA > 2 ? 3 : 0] + _ko);   \
    S##w0 = *(const uint4*)(w_src[0] + _ko);                            \
    S##w1 = *(const uint4*)(w_src[1] + _ko);                            \
    S##w2 = *(const uint4*)(w_src[2] + _ko);                            \
    S##w3 = *(const uint4*)(w_src[3] + _ko);                            \
  } while (0)
#define PIO_STORE_TILE(S, buf)                                          \
  do {                                                                  \
    char* _sa = smem + (buf) * STAGE;                                   \
    char* _sw = _sa + A_BYTES;                                          \
    if (PIO_ABL_NOLDSW) break;                                          \
    *(uint4*)(_sa + lds_off[0]) = S##a0;                                \
    *(uint4*)(_sa + lds_off[1]) = S##a1;                                \
    if constexpr (NA > 2) *(uint4*)(_sa + lds_off[2]) = S##a2;          \
    if constexpr (NA > 2) *(uint4*)(_sa + lds_off[3]) = S##a3;          \
    *(uint4*)(_sw + lds_off[0]) = S##w0;                                \
    *(uint4*)(_sw + lds_off[1]) = S##w1;                                \
    *(uint4*)(_sw + lds_off[2]) = S##w2;                                \
    *(uint4*)(_sw + lds_off[3]) = S##w3;                                \
  } while (0)
#define PIO_COMPUTE_TILE(buf)                                                        \
  do {                                                                               \
    const char* _sa = smem + (buf) * STAGE;                                          \
    const char* _sw = _sa + A_BYTES;                                                 \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                  \
      const int co = (((2 * s + h) ^ sw7) << 4);                                     \
      frag_t fa[MI];                                                                 \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) fa[i] = *(const frag_t*)(_sa + a_rd[i] + co); \
      const frag_t fw0 = *(const frag_t*)(_sw + w_rd[0] + co);                       \
      const frag_t fw1 = *(const frag_t*)(_sw + w_rd[1] + co);                       \
      if (PIO_ABL_NOMFMA) {                                                          \
        asm volatile("" ::"v"(fa[0]), "v"(fw0), "v"(fw1));                           \
        continue;                                                                    \
      }                                                                              \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) {                               \
        acc[i][0] = mfma32(fa[i], fw0, acc[i][0]);                                   \
        acc[i][1] = mfma32(fa[i], fw1, acc[i][1]);                                   \
      }                                                                              \
    }                                                                                \
  } while (0)

  const int nk = g.K / BK;                  // even and >= 2 (checked by the launcher)
  // steady state has no conditionals (tail peeled): conditional loads make hipcc's waitcnt insertion drain
  // the whole load queue at the loop head.
  PIO_LOAD_TILE(p, 0);
  PIO_LOAD_TILE(q, 1);
  PIO_STORE_TILE(p, 0);
  __syncthreads();
  for (int kt = 0; kt < nk - 2; kt += 2) {   // LDS buffer 0 holds tile kt, set q is loading tile kt+1
    PIO_LOAD_TILE(p, kt + 2);
    __builtin_amdgcn_sched_barrier(0);       // keep the loads ahead of the MFMA block (hipcc sinks them otherwise)
    PIO_COMPUTE_TILE(0);
    PIO_STORE_TILE(q, 1);
    __syncthreads();
    PIO_LOAD_TILE(q, kt + 3);
    __builtin_amdgcn_sched_barrier(0);
    PIO_COMPUTE_TILE(1);
    PIO_STORE_TILE(p, 0);
    __syncthreads();
  }
  PIO_COMPUTE_TILE(0);
  PIO_STORE_TILE(q, 1);
  __syncthreads();
  PIO_COMPUTE_TILE(1);
#undef PIO_LOAD_TILE
#undef PIO_STORE_TILE
#undef PIO_COMPUTE_TILE

  // ---- epilogue: accumulators -> LDS [BM][128] fp32 -> row-major 16-B-per-lane global accesses ----
  if (PIO_ABL_NOEPI) {
    if (acc[0][0][0] + acc[0][1][3] + acc[MI - 1][0][5] + acc[MI - 1][1][7] == 12345.678f) g.x[tid] = 1.f;   // keep acc live
    return;
  }
  __syncthreads();                       // every wave is done reading the operand tiles
  float* ct = (float*)smem;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ct[(wm * (BM / 2) + i * 32 + acc_row32(r, lane)) * BN + wn * 64 + j * 32 + r31] = acc[i][j][r];
  __syncthreads();

  if (EPI == EPI_QKV && n0 >= 2 * g.D) {
    // V block: stored transposed ([b][h][d][t]); each thread takes one column and 4 consecutive tokens
    // (Tp % 8 == 0 and m0 % 64 == 0, so the 4 tokens belong to one image and the 8-B store is aligned)
    const int col = tid & 127, rg = tid >> 7;
    const int n = n0 + col, hd = n - 2 * g.D, head = hd >> 6, d = hd & 63;
    const float bn = g.bias[n];
#pragma unroll 4
    for (int i = 0; i < BM / 8; ++i) {
      const int r4 = (rg + 2 * i) * 4, m = m0 + r4;
      if (m >= g.M) continue;
      const int b = m / g.Tp, t = m - b * g.Tp;
      const float v0 = ct[(r4 + 0) * BN + col] + bn, v1 = ct[(r4 + 1) * BN + col] + bn;
      const float v2 = ct[(r4 + 2) * BN + col] + bn, v3 = ct[(r4 + 3) * BN + col] + bn;
      store_half4<T>((T*)g.vT + ((size_t)(b * g.H + head) * 64 + d) * g.Tk + t, v0, v1, v2, v3);
      if (g.qkv_last != nullptr) {
        float* ql = g.qkv_last + ((size_t)b * g.T + t) * g.N + n;
        if (t + 0 < g.T) ql[0] = v0;
        if (t + 1 < g.T) ql[(size_t)g.N] = v1;
        if (t + 2 < g.T) ql[(size_t)2 * g.N] = v2;
        if (t + 3 < g.T) ql[(size_t)3 * g.N] = v3;
      }
    }
    return;
  }
  const int c4 = (tid & 31) * 4, rbase = tid >> 5;
  const int n = n0 + c4;
  const float4 b4 = *(const float4*)(g.bias + n);
  float4 l4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (EPI == EPI_RESIDUAL) l4 = *(const float4*)(g.ls + n);
#pragma unroll 4
  for (int i = 0; i < BM / 8; ++i) {
    const int row = rbase + 8 * i, m = m0 + row;
    if (m >= g.M) continue;
    float4 v = *(const float4*)(ct + row * BN + c4);
    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    if constexpr (EPI == EPI_PATCH_EMBED) {
      const int b = m / g.n2, p = m - b * g.n2;
      const float4 ps = *(const float4*)(g.pos + (size_t)(1 + p) * g.D + n);
      *(float4*)(g.x + (size_t)(b * g.Tp + g.G + p) * g.D + n) = make_float4(v.x + ps.x, v.y + ps.y, v.z + ps.z, v.w + ps.w);
    } else if constexpr (EPI == EPI_RESIDUAL) {
      float4* px = (float4*)(g.x + (size_t)m * g.N + n);
      float4 xo = *px;
      xo.x += l4.x * v.x; xo.y += l4.y * v.y; xo.z += l4.z * v.z; xo.w += l4.w * v.w;
      *px = xo;
    } else if constexpr (EPI == EPI_GELU) {
      store_half4<T>((T*)g.out16 + (size_t)m * g.N + n, gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
    } else {  // EPI_QKV, q or k block (block-uniform: D % 128 == 0)
      const int which = n0 >= g.D ? 1 : 0;
      const int hd = n - which * g.D, head = hd >> 6, d = hd & 63;
      const int b = m / g.Tp, t = m - b * g.Tp;
      T* dst = (which == 0 ? (T*)g.q : (T*)g.k) + ((size_t)(b * g.H + head) * g.Tk + t) * 64 + d;
      store_half4<T>(dst, v.x, v.y, v.z, v.w);
      if (g.qkv_last != nullptr && t < g.T) *(float4*)(g.qkv_last + ((size_t)b * g.T + t) * g.N + n) = v;
    }
  }
}

template <typename T, int EPI, int BM>
static hipError_t launch_one(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  const int smem_bytes = 2 * (BM + BN) * BK * 2;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_vit_gemm<T, EPI, BM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = ceil_div(a.M, BM) * (a.N / BN);
  hipLaunchKernelGGL((k_vit_gemm<T, EPI, BM>), dim3(grid), dim3(256), smem_bytes, s, a);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_typed(GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_PATCH_EMBED: return launch_one<T, EPI_PATCH_EMBED, PIO_GEMM_BM_NARROW>(a, s);
    case EPI_QKV: return launch_one<T, EPI_QKV, 128>(a, s);
    case EPI_RESIDUAL: return launch_one<T, EPI_RESIDUAL, PIO_GEMM_BM_NARROW>(a, s);
    case EPI_GELU: return launch_one<T, EPI_GELU, 128>(a, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_vit_gemm(OperandType t, GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N % BN != 0 || a.K % (2 * BK) != 0 || a.lda % 8 != 0) return hipErrorInvalidValue;
  return t == OP_F16 ? launch_typed<f16>(epi, a, s) : launch_typed<bf16>(epi, a, s);
}

}  // namespace pio
