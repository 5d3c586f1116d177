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
// Tiling: 128x128x64 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles),
// register-staged double-buffered LDS (2 x 32 KiB), one barrier per K-tile.  LDS rows are 128 B
// (64 halfs); the 16-B chunk index is XORed with (row>>1)&7 so that the ds_read_b128 fragment reads
// of a 16-lane group fall on 16 distinct 16-B slots of the 256-B bank row (conflict-free), while the
// staging ds_write_b128 of 8 consecutive lanes covers one whole row half.
// Workgroup ids are remapped so that each XCD's L2 sees a contiguous run of tiles sharing A panels.
#include "common.h"
#include "kernels.h"

namespace pio {

static constexpr int BM = 128, BN = 128, BK = 64;
static constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <typename T, int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int m, int n, float v) {
  if constexpr (EPI == EPI_PATCH_EMBED) {
    const int b = m / g.n2, p = m - b * g.n2;
    g.x[(size_t)(b * g.Tp + g.G + p) * g.D + n] = v + g.bias[n] + g.pos[(size_t)(1 + p) * g.D + n];
  } else if constexpr (EPI == EPI_RESIDUAL) {
    float* px = g.x + (size_t)m * g.N + n;
    *px += g.ls[n] * (v + g.bias[n]);
  } else if constexpr (EPI == EPI_GELU) {
    ((T*)g.out16)[(size_t)m * g.N + n] = (T)gelu_erf(v + g.bias[n]);
  }
}

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void k_vit_gemm(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Vec8<T>::type frag_t;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, h = lane >> 5, r31 = lane & 31;

  const int ntn = g.N / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % ntn, tm = bid / ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging assignment: 4 x 16-B chunks of A and of W per thread per K-tile ----
  const int kc = tid & 7;          // 16-B chunk inside the 128-B row
  const int row0 = tid >> 3;       // rows row0 + 32*i
  const T* a_src[4];
  const T* w_src[4];
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + 32 * i;
    int am = m0 + row;
    am = am < g.M ? am : g.M - 1;   // clamp: rows past M are computed on a copy of the last row, never stored
    a_src[i] = (const T*)g.A + (size_t)am * g.lda + kc * 8;
    w_src[i] = (const T*)g.W + (size_t)(n0 + row) * g.K + kc * 8;
    lds_off[i] = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
  }
  uint4 ra[4], rw[4];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *(const uint4*)(a_src[i] + kt * BK);
      rw[i] = *(const uint4*)(w_src[i] + kt * BK);
    }
  };
  auto store_tile = [&](int buf) {
    char* sa = smem + buf * 2 * TILE_BYTES;
    char* sw = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(sa + lds_off[i]) = ra[i];
      *(uint4*)(sw + lds_off[i]) = rw[i];
    }
  };

  // ---- fragment read addresses ----
  const int sw7 = (lane >> 1) & 7;  // == ((row>>1)&7) for row = 32*x + (lane&31)
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_rd[i] = (wm * 64 + i * 32 + r31) * 128;
    w_rd[i] = (wn * 64 + i * 32 + r31) * 128;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* sa = smem + buf * 2 * TILE_BYTES;
    const char* sw = sa + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int co = (((2 * s + h) ^ sw7) << 4);
      frag_t fa[2], fw[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *(const frag_t*)(sa + a_rd[i] + co);
        fw[i] = *(const frag_t*)(sw + w_rd[i] + co);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i], fw[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  if constexpr (EPI == EPI_QKV) {
    T* qb = (T*)g.q; T* kb = (T*)g.k; T* vb = (T*)g.vT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + acc_row32(r, lane);
        if (m >= g.M) continue;
        const int b = m / g.Tp, t = m - b * g.Tp;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 64 + j * 32 + r31;
          const float v = acc[i][j][r] + g.bias[n];
          const int which = n / g.D, hd = n - which * g.D;
          const int head = hd >> 6, d = hd & 63;
          const size_t bh = (size_t)b * g.H + head;
          if (which == 0) qb[(bh * g.Tk + t) * 64 + d] = (T)v;
          else if (which == 1) kb[(bh * g.Tk + t) * 64 + d] = (T)v;
          else vb[(bh * 64 + d) * g.Tk + t] = (T)v;
          if (g.qkv_last != nullptr && t < g.T) g.qkv_last[((size_t)b * g.T + t) * g.N + n] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + acc_row32(r, lane);
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 64 + j * 32 + r31;
          epilogue_store<T, EPI>(g, m, n, acc[i][j][r]);
        }
      }
  }
}

template <typename T, int EPI>
static hipError_t launch_one(const GemmArgs& a, hipStream_t s) {
  static bool attr_set = false;
  const int smem_bytes = 4 * TILE_BYTES;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_vit_gemm<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = ceil_div(a.M, BM) * (a.N / BN);
  hipLaunchKernelGGL((k_vit_gemm<T, EPI>), dim3(grid), dim3(256), smem_bytes, s, a);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_typed(GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_PATCH_EMBED: return launch_one<T, EPI_PATCH_EMBED>(a, s);
    case EPI_QKV: return launch_one<T, EPI_QKV>(a, s);
    case EPI_RESIDUAL: return launch_one<T, EPI_RESIDUAL>(a, s);
    case EPI_GELU: return launch_one<T, EPI_GELU>(a, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_vit_gemm(OperandType t, GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N % BN != 0 || a.K % BK != 0 || a.lda % 8 != 0) return hipErrorInvalidValue;
  return t == OP_F16 ? launch_typed<f16>(epi, a, s) : launch_typed<bf16>(epi, a, s);
}

}  // namespace pio
