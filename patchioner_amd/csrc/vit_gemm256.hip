// ViT linear layers at M >= 8192 rows: 256 x 256 x 64 tiles, 8 waves in two ping-ponging groups (gfx950).
//
//   C[M][N] = A[M][K] * W[N][K]^T    same operands, same epilogues and the SAME per-element arithmetic as k_vit_gemm
//                                     (v_mfma_f32_32x32x16, k ascending in steps of 16): an output element does not
//                                     depend on which of the two kernels, or which tile, produced it.
//
// Why a second kernel: a 128 x 128 tile stages 32 KiB per 2.1 MFLOP and the L2 -> LDS path (about 64 B/clk/CU) is then
// co-critical with the MFMA pipe; a 256 x 256 tile stages 64 KiB per 8.4 MFLOP.  One workgroup fills a CU: 512 threads,
// <= 256 VGPRs, 128 KiB of LDS.
//
// Tile geometry.  A K-tile (64 deep) is four 16-KiB half-tiles: A0 / A1 = rows 0-127 / 128-255 of the tile, B0 / B1 =
// W rows (output columns) 0-127 / 128-255, each [128 rows][128 B] with the 16-B chunk index XORed by (row >> 1) & 7
// (conflict-free ds_read_b128 fragment reads; the XOR sits on the per-lane SOURCE address of the LDS-DMA, the LDS image
// itself is written lane-linear).  Two K-tiles are resident (2 x 64 KiB).
// Wave (wr, wc) = (wid >> 2, wid & 3) owns a 128 x 64 patch in four quadrants (i, j): rows i * 128 + wr * 64 + [0, 64) x
// columns j * 128 + wc * 32 + [0, 32): quadrant (i, j) needs 64 rows of A-half i and 32 rows of B-half j, so every wave
// reads every half-tile exactly once per K-tile (fragments stay in registers across the two quadrants that share them).
//
// Schedule (the details, the LDS-DMA pipeline and its WAR / RAW arguments are spelled out at the main loop).  A K-tile is four
// phases, one quadrant each, in the order (A0,B0) (A0,B1) (A1,B1) (A1,B0); ONE workgroup barrier per phase.  Between two
// barriers the two wave groups run different programs: waves 0-3 the 8 MFMAs of phase k (operands read in the interval before),
// then the fragment reads of phase k + 1 and the LDS-DMA issue; waves 4-7 the reads of phase k, the LDS-DMA, then the MFMAs of
// phase k -- so on every SIMD one wave multiplies while its partner reads.  (Round 2's first schedule, two barriers per phase
// with the second group one segment behind, measured 1-4 % slower and was removed from the tree in round 4: git history, tools/microbench/attic/.)
#include "common.h"
#include "kernels.h"

namespace pio {

#ifndef PIO_G256_NOSTAGGER      // diagnostic: both groups in lock-step
#define PIO_G256_NOSTAGGER 0
#endif
#ifndef PIO_G256_SETPRIO
#define PIO_G256_SETPRIO 1
#endif

#ifdef PIO_G256_STAMPS           // diagnostic builds only: s_memtime at kernel entry / first operands landed / main loop done / end,
__device__ unsigned long long* g256_stamps = nullptr;   // 4 words per workgroup, written by wave 0 (never read by the kernel)
#define G256_STAMP(i)                                                                                              \
  do {                                                                                                             \
    if (g256_stamps != nullptr && tid == 0) g256_stamps[4 * blockIdx.x + (i)] = __builtin_readcyclecounter();      \
  } while (0)
#else
#define G256_STAMP(i) do { } while (0)
#endif

namespace g256 {

static constexpr int TM = 256, TN = 256, TK = 64;
static constexpr int HALF = 128 * TK * 2;      // 16 KiB
static constexpr int LDS_BYTES = 8 * HALF;     // 128 KiB: [A0 A1](buffer 0) [A0 A1](buffer 1) [B0 B1](buffer 0) [B0 B1](buffer 1)
// byte offset of half-tile (operand o = 0 A / 1 W, buffer, half): with this order every fragment read of one operand is
// one base register per k-step plus a 16-bit immediate
__host__ __device__ constexpr int half_off(int o, int buf, int half) { return o * 4 * HALF + buf * 2 * HALF + half * HALF; }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <typename T> struct Vec4h { typedef T type __attribute__((ext_vector_type(4))); };

#define G256_BARRIER()                         \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

// One LDS-DMA half-tile: 2 x 1 KiB per wave (rows 8 wid + (lane >> 3) and 64 + the same) of K-tile kt into (buf, half).
// A rows are clamped per lane (4 offsets); W rows never are: one per-lane offset, the rest rides in the scalar offset.
#define G256_ISSUE_A(buf, half, kt)                                                                                        \
  do {                                                                                                                     \
    char* const _d = smem + half_off(0, (buf), (half)) + wid * 1024;                                                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)_d, 16, a_off[half][0], (kt) * (TK * 2), 0, 0);               \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(_d + 8192), 16, a_off[half][1], (kt) * (TK * 2), 0, 0);      \
  } while (0)
#define G256_ISSUE_W(buf, half, kt)                                                                                        \
  do {                                                                                                                     \
    char* const _d = smem + half_off(1, (buf), (half)) + wid * 1024;                                                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)_d, 16, w_off, (kt) * (TK * 2) + (half) * 128 * wrow, 0, 0);  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(_d + 8192), 16, w_off, (kt) * (TK * 2) + ((half) * 128 + 64) * wrow, 0, 0); \
  } while (0)

}  // namespace g256

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void k_vit_gemm256(const GemmArgs g) {
  using namespace g256;
  static_assert(EPI != EPI_RESIDUAL, "proj / fc2 at 256 x 256 tiles: k_vit_gemm_roll (the old x joins the sum inside the main loop, kernels.h)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Vec8<T>::type frag_t;
  typedef typename Vec4h<T>::type half4_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3, h = lane >> 5, r31 = lane & 31;

  G256_STAMP(0);
  const int ntn = g.N / TN;
  const int ntiles = ((g.M + TM - 1) / TM) * ntn;
  if ((int)blockIdx.x >= ntiles) {            // extra workgroups on idle compute units: the next GEMM's weights (vit_gemm.hip, "cold weights")
    gemm_warm_next(g, blockIdx.x - ntiles, gridDim.x - ntiles, tid, 512);
    return;
  }
  const int bid = xcd_remap(blockIdx.x, ntiles);
  const int tn = bid % ntn, tm = bid / ntn;
  const int m0 = tm * TM, n0 = tn * TN;

  // ---- LDS-DMA source offsets (bytes from A / W; the K-tile advance goes in the scalar offset)
  const int prow = 8 * wid + (lane >> 3);
  const uint32_t kcs = (uint32_t)(((lane & 7) ^ ((4 * wid + (lane >> 4)) & 7)) * 16);
  uint32_t a_off[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int am = m0 + i * 128 + q * 64 + prow;
      am = am < g.M ? am : g.M - 1;            // rows past M: a copy of the last row, computed and never stored
      a_off[i][q] = (uint32_t)am * (uint32_t)(g.lda * 2) + kcs;
    }
  const int wrow = g.K * 2;                    // bytes per W row
  const uint32_t w_off = (uint32_t)(n0 + prow) * (uint32_t)wrow + kcs;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((size_t)g.M * g.lda * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (int)((size_t)g.N * g.K * 2), 0x00020000);

  // ---- fragment read offsets inside a K-tile buffer
  const int sw7 = (lane >> 1) & 7;
  const int a_rd = (wr * 64 + r31) * 128;                       // + half_off(0, buf, i) + rt * 4096
  const int b_rd = half_off(1, 0, 0) + (wc * 32 + r31) * 128;   // + half_off(0, buf, j)
  int co[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) co[s] = ((2 * s + h) ^ sw7) << 4;

  f32x16 acc[2][2][2];     // [A half i][B half j][row tile rt]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][rt][r] = 0.f;

  // V columns of the fused qkv projection keep the token on the accumulator REGISTER (the epilogue stores V transposed
  // straight from registers); every other tile keeps the token on the LANE, so that a lane holds 4 consecutive output
  // columns per register group and the epilogue stages whole 16-B pieces.  Swapping the two MFMA operands changes
  // neither the products nor their order.
  const bool v_block = EPI == EPI_QKV && n0 >= 2 * g.D;      // block-uniform (D % 256 == 0, checked by the launcher)

  const int nk = g.K / TK;
  frag_t fa[2][4], fb0[4], fb1[4];

#define G256_READ_A(buf, i)                                                                              \
  _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) _Pragma("unroll") for (int s = 0; s < 4; ++s)        \
      fa[rt][s] = *(const frag_t*)(smem + half_off(0, (buf), (i)) + rt * 4096 + a_rd + co[s])
#define G256_READ_B(dst, buf, j)                                                                         \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) dst[s] = *(const frag_t*)(smem + half_off(0, (buf), (j)) + b_rd + co[s])
#define G256_MMA(i, j, fb)                                                                               \
  do {                                                                                                   \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)      \
        acc[i][j][rt] = SWAP ? mfma32(fb[s], fa[rt][s], acc[i][j][rt]) : mfma32(fa[rt][s], fb[s], acc[i][j][rt]); \
  } while (0)

  // ------------------------------------------------------------------------------------------------------------
  // ONE barrier per phase, the two groups run DIFFERENT programs between two barriers ("interval" k):
  //   waves 0-3:  MFMAs of phase k (operands read in interval k-1), then the ds_reads of phase k+1, then the LDS-DMA
  //   waves 4-7:  the ds_reads of phase k, the LDS-DMA, then the MFMAs of phase k
  // so one wave of every SIMD multiplies while its partner reads.
  // LDS-DMA in interval 4t+0: A1(t+1), 4t+1: A0(t+2), 4t+2: B0(t+2) then s_waitcnt vmcnt(4), 4t+3: B1(t+2).
  //   WAR  a half-tile read for phase p (by waves 0-3 in interval p-1, by waves 4-7 in interval p; both consumed by
  //        the MFMAs of interval p, i.e. before the barrier that closes it) is overwritten by DMA issued in an
  //        interval >= p+1: A0/B0(t) are read for phase 4t and restaged in 4t+1 / 4t+2, B1(t): 4t+1 -> 4t+3,
  //        A1(t): 4t+2 -> 4t+4.
  //   RAW  the wait of interval 4t+2 retires all of K-tile t+1 (only A0, B0 of t+2 stay in flight); every wave
  //        executes it before that interval's barrier, and K-tile t+1 is first read in interval 4t+3 (waves 0-3,
  //        for phase 4t+4).  K-tile 0: A0, B0, B1 are retired before the loop, A1(0) by an extra wait in interval 0
  //        (first read in interval 1), so the first MFMA waits for 48 KiB of operands, not for 96.
  G256_ISSUE_A(0, 0, 0);
  G256_ISSUE_W(0, 0, 0);
  G256_ISSUE_W(0, 1, 0);
  G256_ISSUE_A(0, 1, 0);
  G256_ISSUE_A(1, 0, 1);
  G256_ISSUE_W(1, 0, 1);
  G256_ISSUE_W(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // A0, B0, B1 of K-tile 0 have landed
  G256_BARRIER();
  G256_STAMP(1);

#define G256_DMA0(t, BUF) if ((t) + 1 < nk) G256_ISSUE_A((BUF) ^ 1, 1, (t) + 1)
#define G256_DMA1(t, BUF) if ((t) + 2 < nk) G256_ISSUE_A(BUF, 0, (t) + 2)
#define G256_DMA2(t, BUF) if ((t) + 2 < nk) G256_ISSUE_W(BUF, 0, (t) + 2)
#define G256_DMA3(t, BUF) if ((t) + 2 < nk) G256_ISSUE_W(BUF, 1, (t) + 2)
#define G256_WAIT2(t)                                                                                    \
  do {                                                                                                   \
  if ((t) + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                   \
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
  } while (0)
#define G256_FIRSTWAIT(t) do { if ((t) == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); } while (0)
#define G256_SB() __builtin_amdgcn_sched_barrier(0)

  // waves 0-3
#define G256_KTILE_G0(t, BUF)                                                                     \
  do {                                                                                                   \
  G256_MMA(0, 0, fb0); G256_SB(); G256_READ_B(fb1, BUF, 1); G256_DMA0(t, BUF); G256_FIRSTWAIT(t); G256_BARRIER(); \
  G256_MMA(0, 1, fb1); G256_SB(); G256_READ_A(BUF, 1); G256_DMA1(t, BUF); G256_BARRIER();              \
  G256_MMA(1, 1, fb1); G256_SB(); G256_DMA2(t, BUF); G256_WAIT2(t); G256_BARRIER();                    \
  G256_MMA(1, 0, fb0); G256_SB();                                                                      \
  if ((t) + 1 < nk) { G256_READ_A((BUF) ^ 1, 0); G256_READ_B(fb0, (BUF) ^ 1, 0); }                     \
  G256_DMA3(t, BUF); G256_BARRIER();                                                                   \
  } while (0)
  // waves 4-7
#define G256_KTILE_G1(t, BUF)                                                                     \
  do {                                                                                                   \
  G256_READ_A(BUF, 0); G256_READ_B(fb0, BUF, 0); G256_DMA0(t, BUF); G256_SB(); G256_MMA(0, 0, fb0); G256_FIRSTWAIT(t); G256_BARRIER(); \
  G256_READ_B(fb1, BUF, 1); G256_DMA1(t, BUF); G256_SB(); G256_MMA(0, 1, fb1); G256_BARRIER();         \
  G256_READ_A(BUF, 1); G256_DMA2(t, BUF); G256_SB(); G256_MMA(1, 1, fb1); G256_WAIT2(t); G256_BARRIER(); \
  G256_DMA3(t, BUF); G256_SB(); G256_MMA(1, 0, fb0); G256_BARRIER();                                   \
  } while (0)
#define G256_MAINLOOP(SWAP_)                                                                             \
  do {                                                                                                   \
  constexpr bool SWAP = SWAP_;                                                                         \
  if (wr == 0) {                                                                                       \
    G256_READ_A(0, 0); G256_READ_B(fb0, 0, 0);                                                         \
    for (int t = 0; t < nk; t += 2) { G256_KTILE_G0(t, 0); G256_KTILE_G0(t + 1, 1); }                  \
  } else {                                                                                             \
    for (int t = 0; t < nk; t += 2) { G256_KTILE_G1(t, 0); G256_KTILE_G1(t + 1, 1); }                  \
  }                                                                                                    \
  } while (0)

  if (v_block) G256_MAINLOOP(false);
  else G256_MAINLOOP(true);
#undef G256_MAINLOOP
#undef G256_KTILE_G0
#undef G256_KTILE_G1
#undef G256_READ_A
#undef G256_READ_B
#undef G256_MMA

  // =================================================================================================== epilogues
  G256_STAMP(2);
  // (image, row inside the image) of tile row r without a per-row integer division: one scalar division per workgroup,
  // then wrap (rows per image >= 256 for every crop from 224^2 on: at most one wrap per tile)
  const int per = EPI == EPI_PATCH_EMBED ? g.n2 : g.Tp;
  const int img0 = m0 / per, row0 = m0 - img0 * per;
#define G256_SPLIT(r, b, t)                 \
  int b = img0, t = row0 + (r);             \
  while (t >= per) { t -= per; ++b; }

  if (v_block) {
    // token on the register: acc[i][j][rt][r] = C[m = m0 + 128 i + 64 wr + 32 rt + row32(r, lane)][n = n0 + 128 j + 32 wc + r31]
    typedef T half2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int nn = n0 + j * 128 + wc * 32 + r31, hd = nn - 2 * g.D, head = hd >> 6, d = hd & 63;
          const int rb = i * 128 + wr * 64 + rt * 32;
          const float bn = g.bias[nn];
          if (g.qkv_last != nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rr = rb + acc_row32(r, lane);
              G256_SPLIT(rr, b, t);
              if (m0 + rr < g.M && t < g.T) g.qkv_last[((size_t)b * g.T + t) * g.N + nn] = acc[i][j][rt][r] + bn;
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
                p2[0] = (T)(acc[i][j][rt][4 * (a + e) + 2 * w2] + bn);
                p2[1] = (T)(acc[i][j][rt][4 * (a + e) + 2 * w2 + 1] + bn);
                x[e][w2] = __builtin_bit_cast(uint32_t, p2);
              }
            const auto s0 = __builtin_amdgcn_permlane32_swap(x[0][0], x[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(x[0][1], x[1][1], false, false);
            const int rr = rb + 8 * (a + h);
            if (m0 + rr >= g.M) continue;
            G256_SPLIT(rr, b, t);
            *(uint4*)((T*)g.vT + ((size_t)(b * g.H + head) * 64 + d) * g.Tk + t) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
        }
    G256_STAMP(3);
    return;
  }

  // token on the lane: acc[i][j][rt][r] = C[m = m0 + 128 i + 64 wr + 32 rt + r31][n = n0 + 128 j + 32 wc + 8 (r >> 2) + 4 h + (r & 3)]
  constexpr bool HALF_OUT = EPI == EPI_GELU || EPI == EPI_QKV;
  const bool capture = EPI == EPI_QKV && g.qkv_last != nullptr;   // last block: the fp32 qkv the reference's hook takes
  if (HALF_OUT && !capture) {
    // ---- operand-precision outputs: bias (+ GELU) in registers, one [256][256] half image in LDS (512-B rows, the 16-B
    //      chunk index XORed with row & 15), read back as 16 B per lane: two whole 512-B output rows per wave-instruction.
    float4 bq[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int a = 0; a < 4; ++a) bq[j][a] = *(const float4*)(g.bias + n0 + j * 128 + wc * 32 + 8 * a + 4 * h);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 b4 = bq[j][a];
        const int c = j * 16 + wc * 4 + a;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            const int ml = i * 128 + wr * 64 + rt * 32 + r31;
            float v0 = acc[i][j][rt][4 * a + 0] + b4.x, v1 = acc[i][j][rt][4 * a + 1] + b4.y;
            float v2 = acc[i][j][rt][4 * a + 2] + b4.z, v3 = acc[i][j][rt][4 * a + 3] + b4.w;
            if constexpr (EPI == EPI_GELU) {
              if (g.act == 1) {
                v0 = quick_gelu(v0); v1 = quick_gelu(v1); v2 = quick_gelu(v2); v3 = quick_gelu(v3);
              } else {
                const pio_f32x2 g01 = gelu_erf2((pio_f32x2){v0, v1}), g23 = gelu_erf2((pio_f32x2){v2, v3});
                v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
              }
            }
            half4_t o;
            o[0] = (T)v0; o[1] = (T)v1; o[2] = (T)v2; o[3] = (T)v3;
            *(half4_t*)(smem + ml * 512 + ((c ^ (ml & 15)) << 4) + 8 * h) = o;
          }
      }
    __syncthreads();
    const int cc = tid & 31;
    const int n = n0 + 8 * cc;
    const int which = n0 >= g.D ? 1 : 0;
    const int hd = n - which * g.D, head = hd >> 6, d = hd & 63;
    T* const qk = which == 0 ? (T*)g.q : (T*)g.k;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int ml = (tid >> 5) + 16 * it, m = m0 + ml;
      if (m >= g.M) continue;
      const uint4 v = *(const uint4*)(smem + ml * 512 + ((cc ^ (ml & 15)) << 4));
      if constexpr (EPI == EPI_GELU) {
        *(uint4*)((T*)g.out16 + (size_t)m * g.N + n) = v;
      } else {
        G256_SPLIT(ml, b, t);
        *(uint4*)(qk + ((size_t)(b * g.H + head) * g.Tk + t) * 64 + d) = v;
      }
    }
    G256_STAMP(3);
    return;
  }

  // ---- fp32 outputs (residual stream, patch embedding, the captured qkv): two passes of 128 rows through a
  //      [128][256] fp32 image (1-KiB rows, chunk index XORed with row & 7); a wave-instruction reads back one whole row
  //      (row = wave-uniform: its bounds test and image split are scalar).  What the read-back combines with (the position
  //      rows) is loaded BEFORE the pass is staged, all 16 rows of the wave at once, and the
  //      second pass's rows while the first pass is read back: one exposed memory latency per tile instead of eight.
  const int n = n0 + 4 * lane;
  const float4 b4 = *(const float4*)(g.bias + n);
  constexpr bool PRE = EPI == EPI_PATCH_EMBED;
  float4 pre0[PRE ? 16 : 1], pre1[PRE ? 16 : 1];
#define G256_PRELOAD(dst, i)                                                                            \
  do {                                                                                                  \
    if constexpr (PRE) {                                                                                \
      _Pragma("unroll") for (int it = 0; it < 16; ++it) {                                               \
        const int rr = (i) * 128 + wid * 16 + it;                                                       \
        dst[it] = make_float4(0.f, 0.f, 0.f, 0.f);                                                      \
        if (m0 + rr < g.M) {                                                                            \
          G256_SPLIT(rr, b, p);                                                                         \
          (void)b;                                                                                      \
          dst[it] = *(const float4*)(g.pos + (size_t)(1 + p) * g.D + n);                                \
        }                                                                                               \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)
#define G256_STAGE32(i)                                                                                 \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)       \
  _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                       \
    const int ml = wr * 64 + rt * 32 + r31, c = j * 32 + wc * 8 + 2 * a + h;                            \
    *(float4*)(smem + ml * 1024 + ((c ^ (ml & 7)) << 4)) =                                              \
        make_float4(acc[i][j][rt][4 * a], acc[i][j][rt][4 * a + 1], acc[i][j][rt][4 * a + 2], acc[i][j][rt][4 * a + 3]); \
  }
#define G256_READOUT32(i, pre)                                                                          \
  _Pragma("unroll") for (int it = 0; it < 16; ++it) {                                                   \
    const int ml = wid * 16 + it, rr = (i) * 128 + ml, m = m0 + rr;     /* wave-uniform row */          \
    if (m >= g.M) continue;                                                                             \
    float4 v = *(const float4*)(smem + ml * 1024 + ((lane ^ (ml & 7)) << 4));                           \
    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;                                                 \
    if constexpr (EPI == EPI_PATCH_EMBED) {                                                             \
      G256_SPLIT(rr, b, p);                                                                             \
      const float4 ps = pre[it];                                                                        \
      *(float4*)(g.x + (size_t)(b * g.Tp + g.G + p) * g.D + n) = make_float4(v.x + ps.x, v.y + ps.y, v.z + ps.z, v.w + ps.w); \
    } else if constexpr (EPI == EPI_QKV) {       /* q or k columns of the last block */                 \
      const int which = n0 >= g.D ? 1 : 0;                                                              \
      const int hd = n - which * g.D, head = hd >> 6, d = hd & 63;                                      \
      G256_SPLIT(rr, b, t);                                                                             \
      half4_t o;                                                                                        \
      o[0] = (T)v.x; o[1] = (T)v.y; o[2] = (T)v.z; o[3] = (T)v.w;                                       \
      *(half4_t*)((which == 0 ? (T*)g.q : (T*)g.k) + ((size_t)(b * g.H + head) * g.Tk + t) * 64 + d) = o; \
      if (t < g.T) *(float4*)(g.qkv_last + ((size_t)b * g.T + t) * g.N + n) = v;                        \
    }                                                                                                   \
  }

  G256_PRELOAD(pre0, 0);
  G256_STAGE32(0);
  __syncthreads();
  G256_PRELOAD(pre1, 1);
  G256_READOUT32(0, pre0);
  __syncthreads();                       // pass 0 has been read out
  G256_STAGE32(1);
  __syncthreads();
  G256_READOUT32(1, pre1);
#undef G256_PRELOAD
#undef G256_STAGE32
#undef G256_READOUT32
#undef G256_SPLIT
  G256_STAMP(3);
}

template <typename T, int EPI>
static hipError_t launch256_one(const GemmArgs& a, hipStream_t s) {
  static DeviceOnce attr_once; bool& attr_set = attr_once.flag();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_vit_gemm256<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       g256::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  // one workgroup owns a compute unit: extra (weight-warming) workgroups only where the tiles leave compute units idle
  const int tiles = ceil_div(a.M, g256::TM) * (a.N / g256::TN);
  const int spare = tiles < 256 ? (256 - tiles < 16 ? 256 - tiles : 16) : 0;
  const int grid = tiles + (PIO_GEMM_WARM_NEXT && a.pf != nullptr && a.pf_bytes > 0 ? spare : 0);
  hipLaunchKernelGGL((k_vit_gemm256<T, EPI>), dim3(grid), dim3(512), g256::LDS_BYTES, s, a);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch256_typed(GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  switch (epi) {
    case EPI_PATCH_EMBED: return launch256_one<T, EPI_PATCH_EMBED>(a, s);
    case EPI_QKV: return launch256_one<T, EPI_QKV>(a, s);
    case EPI_RESIDUAL: return hipErrorInvalidValue;      // k_vit_gemm_roll
    case EPI_GELU: return launch256_one<T, EPI_GELU>(a, s);
  }
  return hipErrorInvalidValue;
}

bool vit_gemm256_fits(GemmEpilogue epi, const GemmArgs& a) {
  if (epi == EPI_RESIDUAL) return false;      // proj / fc2 at 256 x 256 tiles are the rolling kernel's (vit_gemm_roll.hip)
  if (a.N % g256::TN != 0 || a.K % (2 * g256::TK) != 0 || a.lda % 8 != 0) return false;
  if (epi == EPI_QKV && (a.D % g256::TN != 0 || a.Tp % 8 != 0)) return false;
  if ((size_t)a.M * a.lda * 2 >= ((size_t)1 << 31) || (size_t)a.N * a.K * 2 >= ((size_t)1 << 31)) return false;
  return true;
}

hipError_t launch_vit_gemm256(OperandType t, GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || !vit_gemm256_fits(epi, a)) return hipErrorInvalidValue;
  return t == OP_F16 ? launch256_typed<f16>(epi, a, s) : launch256_typed<bf16>(epi, a, s);
}


}  // namespace pio
