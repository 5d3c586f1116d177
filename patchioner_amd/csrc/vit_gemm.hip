// ViT linear layers as one MFMA GEMM with fused epilogues (gfx950).
//
//   C[M][N] = A[M][K] * W[N][K]^T        A, W in fp16/bf16 (K contiguous for both: torch Linear layout),
//                                         fp32 accumulation on v_mfma_f32_32x32x16_{f16,bf16}.
//
// Replaces the stock ATen calls behind DINOv2's PatchEmbed conv, attn.qkv, attn.proj, mlp.fc1, mlp.fc2
// (reached from P/src/model.py:783) and fuses what followed them in the reference: bias, position
// embedding add, q/k/v head split (+ the fp32 capture the qkv forward hook takes,
// P/src/dino_extraction.py:7-9), LayerScale (folded into W and bias at load) + residual add, exact-erf GELU.
//
// This file: the dispatch (launch_vit_gemm) and k_vit_gemm, the 128-wide kernel that serves GEMMs with fewer than ~150
// tiles of 256 x 256 (a synchronous 16-image forward's proj / fc2, small box-sequence batches); larger ones go to
// k_vit_gemm256 (vit_gemm256.hip), which computes every output element with the same k order -- identical bits.
//
// k_vit_gemm tiling: BM x 128 x 64 per 256-thread workgroup, BM = 128 (4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles)
// for the wide GEMMs, BM = 64 (each wave 32x64) for the N = D ones so that they still give >= 1.5 workgroups per CU at 16
// images.  Operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write): the
// wide GEMMs use ONE 32-KiB buffer with up to four workgroups per CU covering each other's load latency, the N = D GEMMs
// two buffers (one barrier per K-tile).  LDS rows are 128 B (64 halfs); an LDS-DMA writes lane-linearly, so the XOR
// swizzle that makes the ds_read_b128 fragment reads of a 16-lane group fall on 16 distinct 16-B slots (conflict-free) is
// applied to each lane's SOURCE address instead.  The epilogue passes the accumulators through LDS so that every global
// access is a row-major 16 B (fp32) / 8 B (half) per lane; V is stored transposed straight from the registers
// (v_permlane32_swap).  Workgroup ids are remapped so that each XCD's L2 sees a contiguous run of tiles sharing A panels.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

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
#ifndef PIO_ABL_NOSTORE       // epilogue without its global stores (LDS staging and arithmetic kept)
#define PIO_ABL_NOSTORE 0
#endif
#ifndef PIO_GEMM_WIDE_OCC       // workgroups per CU the single-buffer (wide) form is compiled for
#define PIO_GEMM_WIDE_OCC 4
#endif
#ifndef PIO_GEMM_BM_NARROW      // tile height used when N == D (proj, fc2, patch embed)
#define PIO_GEMM_BM_NARROW 64
#endif

template <typename T>
__device__ __forceinline__ void store_half4(T* dst, float a, float b, float c, float d) {
  typedef T half4_t __attribute__((ext_vector_type(4)));
  half4_t o;
  o[0] = (T)a; o[1] = (T)b; o[2] = (T)c; o[3] = (T)d;
  *(half4_t*)dst = o;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <typename T, int EPI, int BM, int NBUF>
__global__ __launch_bounds__(256, BM >= 256 ? 2 : (NBUF == 1 ? PIO_GEMM_WIDE_OCC : 2)) void k_vit_gemm(const GemmArgs g) {
  constexpr int MI = BM / 64;                      // 32-row MFMA tiles per wave along M (waves are 2 x 2)
  constexpr int NPA = BM / 32;                     // 1-KiB A pieces per wave per K-tile (W: always 4)
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF x STAGE; reused as fp32 [64][128] by the epilogue
  typedef typename Vec8<T>::type frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, h = lane >> 5, r31 = lane & 31;

  const int ntn = g.N / BN;
  const int ntiles = ((g.M + BM - 1) / BM) * ntn;
  if ((int)blockIdx.x >= ntiles) {            // extra workgroups (launch_one): warm the Infinity Cache with the next GEMM's weights
    gemm_warm_next(g, blockIdx.x - ntiles, gridDim.x - ntiles, tid, 256);
    return;
  }
  const int bid = xcd_remap(blockIdx.x, ntiles);
  const int tn = bid % ntn, tm = bid / ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- LDS-DMA staging.  One global_load_lds_dwordx4 wave-instruction fills 1 KiB = 8 tile rows x 128 B in lane
  // order (destination = wave-uniform base + 16 * lane), so the XOR swizzle goes on the SOURCE address: lane l
  // fills slot (l & 7) of row (l >> 3) and therefore fetches the K chunk (l & 7) ^ ((row >> 1) & 7) of that row.
  // Wave w owns pieces w, w + 4, ... (rows 32 i + 8 w + (l >> 3)); (row >> 1) & 7 does not depend on i.
  const int prow = 8 * wid + (lane >> 3);
  const int kcs = ((lane & 7) ^ ((4 * wid + (lane >> 4)) & 7)) * 16;
  uint32_t a_off[NPA], w_off[4];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    int am = m0 + 32 * i + prow;
    am = am < g.M ? am : g.M - 1;   // clamp: rows past M are computed on a copy of the last row, never stored
    a_off[i] = (uint32_t)am * (uint32_t)(g.lda * 2) + kcs;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) w_off[i] = (uint32_t)(n0 + 32 * i + prow) * (uint32_t)(g.K * 2) + kcs;

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

  // EPI_RESIDUAL (kernels.h: resid_join_ktile): the old x of this wave's elements joins the running sums after the K-tile the canonical
  // order names: class e = (row mod 4) + 4 (column half of the 256-grid) -- the registers r = c (mod 4) of every lane (a block of 32 rows
  // starts at a multiple of 32), the column half being the same for the whole workgroup (BN = 128).
  // Two forms.  NBUF == 3 (64-row tiles): all of the wave's x up front, 32 registers.  NBUF < 3 (128-row tiles; every barrier there
  // follows an s_waitcnt vmcnt(0), PIO_LANDED): one class at a time, fetched a K-tile before it joins -- holding 64 registers of x
  // beside 64 of sums spilled into the main loop (round 5).  The one-class buffer needs every class on a K-tile of its own: nk >= 10
  // (launcher).
  const int nk = g.K / BK;                  // even and >= 2 (checked by the launcher)
  constexpr bool XLAZY = EPI == EPI_RESIDUAL && NBUF != 3;
  float xin[EPI == EPI_RESIDUAL && !XLAZY ? MI : 1][2][16];
  float xlz[XLAZY ? MI : 1][2][4];
  const int xj0 = 4 * ((n0 >> 7) & 1);      // classes xj0 .. xj0 + 3 live in this workgroup
#define PIO_X_INDEX(i, j, r) ((uint32_t)min(m0 + wm * (BM / 2) + (i) * 32 + acc_row32((r), lane), g.M - 1) * (uint32_t)g.N + \
                             (uint32_t)(n0 + wn * 64 + (j) * 32 + r31))      /* rows past M: a copy of the last row, never stored; M N < 2^30 (launcher) */
  if constexpr (EPI == EPI_RESIDUAL && !XLAZY) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) xin[i][j][r] = g.x[PIO_X_INDEX(i, j, r)];
  }
  // the class that joins after K-tile ktn, into the one-class buffer (lazy form): its register residue c is scalar, so ONE block of
  // loads serves every class (the rows 8 gq + 4 h + c of a 32-row block); the adds need the register index at compile time
#define PIO_FETCH_X(ktn)                                                                               \
  do {                                                                                                 \
    if constexpr (XLAZY) {                                                                             \
      const int _c = (ktn) - 1 - xj0;                                                                  \
      if (_c >= 0 && _c < 4) {                                                                         \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)   \
          _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                           \
            const int _m = min(m0 + wm * (BM / 2) + i * 32 + 8 * gq + 4 * h + _c, g.M - 1);            \
            xlz[i][j][gq] = g.x[(uint32_t)_m * (uint32_t)g.N + (uint32_t)(n0 + wn * 64 + j * 32 + r31)]; \
          }                                                                                            \
      }                                                                                                \
    }                                                                                                  \
  } while (0)
#define PIO_JOIN_X(kt)                                                                                 \
  do {                                                                                                 \
    if constexpr (XLAZY) {                                                                             \
      const int _c = (kt) - 1 - xj0;                                                                   \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                    \
        if (_c == c) {                                                                                 \
          _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
            _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) acc[i][j][4 * gq + c] += xlz[i][j][gq];   \
        }                                                                                              \
      PIO_FETCH_X((kt) + 1);                                                                           \
    } else if constexpr (EPI == EPI_RESIDUAL) {                                                        \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                    \
        if (resid_join_ktile(xj0 + c, nk) == (kt)) {                                                   \
          _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
            _Pragma("unroll") for (int gq = 0; gq < 4; ++gq)                                           \
              acc[i][j][4 * gq + c] += xin[i][j][4 * gq + c];                                          \
        }                                                                                              \
    }                                                                                                  \
  } while (0)
  PIO_FETCH_X(1);

#define PIO_ISSUE_TILE(kt, buf)                                                                        \
  do {                                                                                                 \
    if (PIO_ABL_NOGLOAD) break;                                                                        \
    const char* _ga = (const char*)g.A + (size_t)(kt) * (BK * 2);                                      \
    const char* _gw = (const char*)g.W + (size_t)(kt) * (BK * 2);                                      \
    char* _sa = smem + (buf) * STAGE + wid * 1024;                                                     \
    _Pragma("unroll") for (int i = 0; i < NPA; ++i)                                                    \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(_ga + a_off[i]), (lds_ptr_t)(_sa + i * 4096), 16, 0, 0);           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(_gw + w_off[i]), (lds_ptr_t)(_sa + A_BYTES + i * 4096), 16, 0, 0); \
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

  // An LDS-DMA tile has LANDED once the issuing wave's vmcnt has drained (and the barrier has passed, for the other waves' pieces).
  // Spelled out: the compiler's own wait insertion treats the DMA as an LDS write that a later ds_read may alias and usually puts a
  // vmcnt(0) in front of the barrier, but round 5 found it missing on the back edge of the two-buffer loop once the kernel also held
  // ordinary global loads (the old x of EPI_RESIDUAL): tiles were multiplied before they had arrived.
#define PIO_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
  if constexpr (NBUF == 1) {
    // one 32-KiB buffer, up to four workgroups per CU: a workgroup's own load and multiply phases alternate and
    // the CU overlaps them across its workgroups (the __syncthreads() after the issue waits vmcnt(0): landed).
    for (int kt = 0; kt < nk; ++kt) {
      PIO_ISSUE_TILE(kt, 0);
      PIO_LANDED();
      __syncthreads();
      PIO_COMPUTE_TILE(0);
      PIO_JOIN_X(kt);
      __syncthreads();
    }
  } else if constexpr (NBUF == 3) {
    // ring of three buffers, two tiles in flight: the wait before tile kt's barrier leaves the NPA + 4 LDS-DMA
    // instructions of tile kt+1 outstanding (counted vmcnt + raw s_barrier: __syncthreads() would drain them).
    // Buffer (kt+2) % 3 held tile kt-1, which every wave has finished reading once it reaches that barrier.
    PIO_ISSUE_TILE(0, 0);
    PIO_ISSUE_TILE(1, 1);
    int cur = 0, nxt = 2;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        if constexpr (NPA == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) PIO_ISSUE_TILE(kt + 2, nxt);
      PIO_COMPUTE_TILE(cur);
      PIO_JOIN_X(kt);
      cur = cur == 2 ? 0 : cur + 1;
      nxt = nxt == 2 ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // every wave is done reading the operand tiles
  } else {
    // two buffers: tile kt+1 is in flight (LDS-DMA, no registers) while tile kt is multiplied; one barrier per tile
    PIO_ISSUE_TILE(0, 0);
    for (int kt = 0; kt < nk - 2; kt += 2) {
      PIO_LANDED();
      __syncthreads();                       // tile kt landed; everyone is done reading buffer 1
      PIO_ISSUE_TILE(kt + 1, 1);
      PIO_COMPUTE_TILE(0);
      PIO_JOIN_X(kt);
      PIO_LANDED();
      __syncthreads();
      PIO_ISSUE_TILE(kt + 2, 0);
      PIO_COMPUTE_TILE(1);
      PIO_JOIN_X(kt + 1);
    }
    PIO_LANDED();
    __syncthreads();
    PIO_ISSUE_TILE(nk - 1, 1);
    PIO_COMPUTE_TILE(0);
    PIO_JOIN_X(nk - 2);
    PIO_LANDED();
    __syncthreads();
    PIO_COMPUTE_TILE(1);
    PIO_JOIN_X(nk - 1);
    __syncthreads();                         // every wave is done reading the operand tiles
  }
#undef PIO_ISSUE_TILE
#undef PIO_COMPUTE_TILE
#undef PIO_JOIN_X
#undef PIO_FETCH_X
#undef PIO_X_INDEX
#undef PIO_LANDED

  // ---- epilogue: accumulators -> LDS [64][128] fp32 (one 32-row MFMA tile row of each wave per pass) ->
  //      row-major 16-B-per-lane global accesses.  Image row c holds tile row (c >> 5) * (BM / 2) + 32 * pass + (c & 31).
  if (PIO_ABL_NOEPI) {
    if (acc[0][0][0] + acc[0][1][3] + acc[MI - 1][0][5] + acc[MI - 1][1][7] == 12345.678f) g.x[tid] = 1.f;   // keep acc live
    return;
  }
  float* ct = (float*)smem;
  const bool v_block = EPI == EPI_QKV && n0 >= 2 * g.D;      // block-uniform: D % 128 == 0
  const int c4 = (tid & 31) * 4, rbase = tid >> 5;
  const int n = n0 + c4;
  const float4 b4 = *(const float4*)(g.bias + n);
  if (v_block) {
    // V is stored TRANSPOSED ([b][h][d][t]) and the accumulator already is: a lane holds one column d and, per
    // register group a = r >> 2, the 4 consecutive tokens 8 a + 4 h + (r & 3).  v_permlane32_swap pairs the two
    // lanes of a column so that each ends up with 8 consecutive tokens = one 16-B store (lane h takes group
    // a + h of the pair (a, a + 1)); no LDS pass.  Tile rows come in aligned groups of 8 and Tp % 8 == 0, so a
    // group belongs to one image and lies wholly inside or outside M.
    typedef T half2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nn = n0 + wn * 64 + j * 32 + r31, hd = nn - 2 * g.D, head = hd >> 6, d = hd & 63;
        const float bn = g.bias[nn];
        if (g.qkv_last != nullptr) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (BM / 2) + i * 32 + acc_row32(r, lane);
            const int b = m / g.Tp, t = m - b * g.Tp;
            if (m < g.M && t < g.T) g.qkv_last[((size_t)b * g.T + t) * g.N + nn] = acc[i][j][r] + bn;
          }
        }
#pragma unroll
        for (int a = 0; a < 4; a += 2) {
          uint32_t x[2][2];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
              half2_t p2;
              p2[0] = (T)(acc[i][j][4 * (a + e) + 2 * w2] + bn);
              p2[1] = (T)(acc[i][j][4 * (a + e) + 2 * w2 + 1] + bn);
              x[e][w2] = __builtin_bit_cast(uint32_t, p2);
            }
          if (PIO_ABL_NOSTORE) {
            if (x[0][0] + x[0][1] + x[1][0] + x[1][1] == 0x12345678u) g.x[tid] = 1.f;
            continue;
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(x[0][0], x[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(x[0][1], x[1][1], false, false);
          const int m = m0 + wm * (BM / 2) + i * 32 + 8 * (a + h);
          if (m >= g.M) continue;
          const int b = m / g.Tp, t = m - b * g.Tp;
          *(uint4*)((T*)g.vT + ((size_t)(b * g.H + head) * 64 + d) * g.Tk + t) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
    return;
  }

#pragma unroll
  for (int pass = 0; pass < MI; ++pass) {
    if (pass > 0) __syncthreads();       // the previous pass has been read out
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ct[(wm * 32 + acc_row32(r, lane)) * BN + wn * 64 + j * 32 + r31] = acc[pass][j][r];
    __syncthreads();

#pragma unroll 4
    for (int i = 0; i < 8; ++i) {
      const int c = rbase + 8 * i, m = m0 + (c >> 5) * (BM / 2) + 32 * pass + (c & 31);
      if (m >= g.M) continue;
      float4 v = *(const float4*)(ct + c * BN + c4);
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      if (PIO_ABL_NOSTORE) {
        if (v.x + v.y + v.z + v.w == 12345.678f) g.x[tid] = 1.f;
        continue;
      }
      if constexpr (EPI == EPI_PATCH_EMBED) {
        const int b = m / g.n2, p = m - b * g.n2;
        const float4 ps = *(const float4*)(g.pos + (size_t)(1 + p) * g.D + n);
        *(float4*)(g.x + (size_t)(b * g.Tp + g.G + p) * g.D + n) = make_float4(v.x + ps.x, v.y + ps.y, v.z + ps.z, v.w + ps.w);
      } else if constexpr (EPI == EPI_RESIDUAL) {      // the old x joined the sum in the main loop (PIO_JOIN_X)
        *(float4*)(g.x + (size_t)m * g.N + n) = v;
      } else if constexpr (EPI == EPI_GELU) {
        if (g.act == 1) store_half4<T>((T*)g.out16 + (size_t)m * g.N + n, quick_gelu(v.x), quick_gelu(v.y), quick_gelu(v.z), quick_gelu(v.w));
        else store_half4<T>((T*)g.out16 + (size_t)m * g.N + n, gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
      } else {  // EPI_QKV, q or k block
        const int which = n0 >= g.D ? 1 : 0;
        const int hd = n - which * g.D, head = hd >> 6, d = hd & 63;
        const int b = m / g.Tp, t = m - b * g.Tp;
        T* dst = (which == 0 ? (T*)g.q : (T*)g.k) + ((size_t)(b * g.H + head) * g.Tk + t) * 64 + d;
        store_half4<T>(dst, v.x, v.y, v.z, v.w);
        if (g.qkv_last != nullptr && t < g.T) *(float4*)(g.qkv_last + ((size_t)b * g.T + t) * g.N + n) = v;
      }
    }
  }
}

// operand buffers per workgroup.  Wide GEMMs (qkv, fc1: >= 2.3 workgroups per CU at 16 images): one 32-KiB buffer
// and up to four co-resident workgroups that overlap each other's load and multiply phases.  N = D GEMMs (proj, fc2:
// 1.5 workgroups per CU, fc2 with 48 K-tiles): the workgroup pipelines its own loads.
#ifndef PIO_GEMM_NBUF_WIDE
#define PIO_GEMM_NBUF_WIDE 1
#endif
#ifndef PIO_GEMM_BM_BIG         // tile height of the wide GEMMs from PIO_GEMM_BIG_M rows on (0: never)
#define PIO_GEMM_BM_BIG 0
#endif
#ifndef PIO_GEMM_NBUF_BIG
#define PIO_GEMM_NBUF_BIG 1
#endif
#ifndef PIO_GEMM_BIG_M
#define PIO_GEMM_BIG_M 8192
#endif
#ifndef PIO_GEMM_NBUF_NARROW
#define PIO_GEMM_NBUF_NARROW 3
#endif

template <typename T, int EPI, int BM, int NBUF>
static hipError_t launch_one(const GemmArgs& a, hipStream_t s) {
  static DeviceOnce attr_once; bool& attr_set = attr_once.flag();
  const int stage = (BM + BN) * BK * 2;
  const int smem_bytes = NBUF * stage > 64 * BN * 4 ? NBUF * stage : 64 * BN * 4;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_vit_gemm<T, EPI, BM, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       smem_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  // "cold weights" (round 4): in a synchronous forward the bank pass of the projection has swept the Infinity Cache, so every weight
  // matrix comes from HBM once per forward; behind a 768-MB sweep fc2 at 16 images takes 48.4 instead of 37.0 us
  // (tools/microbench/gemm256_bench.hip cold).  32 extra workgroups read the next GEMM's weights while this one computes.
  const int grid = ceil_div(a.M, BM) * (a.N / BN) + (PIO_GEMM_WARM_NEXT && a.pf != nullptr && a.pf_bytes > 0 ? 32 : 0);
  hipLaunchKernelGGL((k_vit_gemm<T, EPI, BM, NBUF>), dim3(grid), dim3(256), smem_bytes, s, a);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_typed(GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_PATCH_EMBED: return launch_one<T, EPI_PATCH_EMBED, PIO_GEMM_BM_NARROW, PIO_GEMM_NBUF_NARROW>(a, s);
    case EPI_QKV:
      if constexpr (PIO_GEMM_BM_BIG != 0) { if (a.M >= PIO_GEMM_BIG_M) return launch_one<T, EPI_QKV, PIO_GEMM_BM_BIG == 0 ? 128 : PIO_GEMM_BM_BIG, PIO_GEMM_NBUF_BIG>(a, s); }
      return launch_one<T, EPI_QKV, 128, PIO_GEMM_NBUF_WIDE>(a, s);
    case EPI_RESIDUAL:
      // 32 images and more per launch (the pipeline's shared ViT launches): 128-row tiles fill the chip on their own
      // (>= 1.5 workgroups per CU) and halve the W re-reads: 2.90 vs 3.21 ms per 32-image forward.  Same k order per
      // element, so the result does not depend on the tile height.
      {
        static const int force_bm = [] { const char* e = getenv("PIO_GEMM_RES_BM"); return e ? atoi(e) : 0; }();   // diagnostic: 64 / 128
        if (a.K / BK >= 10 && (force_bm == 128 || (force_bm == 0 && ceil_div(a.M, 128) * (a.N / BN) >= 384))) return launch_one<T, EPI_RESIDUAL, 128, 2>(a, s);
      }
      return launch_one<T, EPI_RESIDUAL, PIO_GEMM_BM_NARROW, PIO_GEMM_NBUF_NARROW>(a, s);
    case EPI_GELU:
      if constexpr (PIO_GEMM_BM_BIG != 0) { if (a.M >= PIO_GEMM_BIG_M) return launch_one<T, EPI_GELU, PIO_GEMM_BM_BIG == 0 ? 128 : PIO_GEMM_BM_BIG, PIO_GEMM_NBUF_BIG>(a, s); }
      return launch_one<T, EPI_GELU, 128, PIO_GEMM_NBUF_WIDE>(a, s);
  }
  return hipErrorInvalidValue;
}

// The 256 x 256 kernel (vit_gemm256.hip) takes a GEMM over once it has enough 256-tiles to occupy the chip: measured
// on MI355X (tools/microbench/gemm256_bench.hip, profiles/r02_gemm256_microbench_*.log) it wins from ~150 tiles on (qkv
// and fc1 at 16 images: 153 / 204 tiles, 25.7 vs 27.8 and 30.4 vs 35.9 us; every GEMM at 64 images) and loses below
// (proj / fc2 at 16 images: 51 tiles).  Same arithmetic per output element, so the choice never changes a result.
// PIO_GEMM256_MIN_TILES overrides the threshold (0 = never use it).
static int gemm256_min_tiles() {
  static const int v = [] { const char* e = getenv("PIO_GEMM256_MIN_TILES"); return e ? atoi(e) : 144; }();
  return v;
}

// One exception, fc1 (GELU epilogue) when its last round of 256-tiles would be mostly empty: at 64 images 792 tiles are
// 3.09 rounds of 256 CUs and the 128-tile kernel's finer tail wins, 110 vs 116 us (at 80 images, 996 tiles = 3.89 rounds,
// the 256 kernel wins 125 vs 134; a single round, 204 tiles at 16 images, too: 30 vs 36).
static bool gemm256_wanted(GemmEpilogue epi, const GemmArgs& a) {
  const int tiles = ceil_div(a.M, 256) * (a.N / 256);
  if (tiles < gemm256_min_tiles()) return false;
  if (epi == EPI_GELU && tiles > 256) {
    const int rounds = ceil_div(tiles, 256);
    if (tiles * 100 < rounds * 256 * 85) return false;
  }
  return true;
}

// The persistent kernel with the rolling epilogue (vit_gemm_roll.hip) takes qkv (without the fp32 capture) and fc1 once every
// CU gets about three tiles: 747 / 996 tiles at 80 images 77.2 vs 81.7 and 112.3 vs 116.3 us, a tie at 64 images (594 / 792
// tiles), slower below (profiles/r03_gemm_microbench.log).  Same arithmetic per output element.  PIO_GEMM_ROLL_MIN_TILES
// overrides the threshold (0 = never).
static int gemm_roll_min_tiles() {
  static const int v = [] { const char* e = getenv("PIO_GEMM_ROLL_MIN_TILES"); return e ? atoi(e) : 704; }();
  return v;
}

// proj / fc2: the rolling kernel from the tile count at which 256 x 256 tiles fill the chip (the 256 kernel's own threshold; it has no
// residual epilogue any more).  PIO_GEMM_RRES_MIN_TILES overrides (0 = never: the 128-wide kernel everywhere).
static int gemm_rres_min_tiles() {
  static const int v = [] { const char* e = getenv("PIO_GEMM_RRES_MIN_TILES"); return e ? atoi(e) : 144; }();
  return v;
}

hipError_t launch_vit_gemm(OperandType t, GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  if (epi == EPI_RESIDUAL) {
    if (gemm_rres_min_tiles() > 0 && a.M > 0 && a.N % 256 == 0 && ceil_div(a.M, 256) * (a.N / 256) >= gemm_rres_min_tiles() &&
        vit_gemm_roll_fits(epi, a))
      return launch_vit_gemm_roll(t, epi, a, s);
  } else
  if (gemm_roll_min_tiles() > 0 && a.M > 0 && a.N % 256 == 0 && ceil_div(a.M, 256) * (a.N / 256) >= gemm_roll_min_tiles() &&
      vit_gemm_roll_fits(epi, a))
    return launch_vit_gemm_roll(t, epi, a, s);
  if (gemm256_min_tiles() > 0 && a.M > 0 && a.N % 256 == 0 && gemm256_wanted(epi, a) && vit_gemm256_fits(epi, a))
    return launch_vit_gemm256(t, epi, a, s);
  if (a.M <= 0 || a.N % BN != 0 || a.K % (2 * BK) != 0 || a.lda % 8 != 0) return hipErrorInvalidValue;
  // the staging offsets are 32-bit byte offsets from A and W
  if ((size_t)a.M * a.lda * 2 >= ((size_t)1 << 32) || (size_t)a.N * a.K * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
  if (epi == EPI_RESIDUAL && (size_t)a.M * a.N >= ((size_t)1 << 30)) return hipErrorInvalidValue;      // 32-bit element index into x
  return t == OP_F16 ? launch_typed<f16>(epi, a, s) : launch_typed<bf16>(epi, a, s);
}

}  // namespace pio
