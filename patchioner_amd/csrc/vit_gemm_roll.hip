// ViT linear layers with many tiles per CU (qkv, fc1 from ~48 images per launch on): PERSISTENT 256 x 256 x 64 workgroups whose
// epilogue ROLLS under the next tile's multiplies (gfx950).
//
//   C[M][N] = A[M][K] * W[N][K]^T    same operands, same tile geometry, same LDS-DMA ring, same two-group schedule and the SAME
//                                     per-element arithmetic as k_vit_gemm256 (vit_gemm256.hip, whose header describes all of
//                                     that): an output element does not depend on which kernel produced it.
//
// Why.  k_vit_gemm256 spends 26-50 % of a tile's life in its epilogue, un-overlapped: one workgroup owns the CU, all of them
// reach their stores at the same time and the fabric, not the CU, sets the epilogue's length (profiles/r02_*).  Here one
// workgroup per CU walks a list of tiles and never drains its pipeline:
//   * the LDS-DMA stream runs on across the tile boundary (the last two K-tiles of a tile stage the first two of the next
//     one): no prologue but the first;
//   * the accumulator is four quadrants Q(i,j) of 128 x 128 (32 registers per wave each), finished one per interval in the
//     last K-tile, in the order the first K-tile of the next tile re-uses them.  Each quadrant leaves through a 32-KiB staging
//     image in the LDS the ring leaves free -- E12: bias (+ GELU), convert, ds_write; one barrier (the main loop's own);
//     E3: ds_read 16 B per lane, global store of whole rows -- in the intervals between its last MFMA and its next first one:
//         interval   L0    L1       L2      L3       F0      F1       F2      F3            S0
//         MFMA into  Q00   Q01      Q11     Q10      Q00'    Q01'     Q11'    Q10'          Q00''
//         hook             E12 Q00  E3 Q00  E12 Q01  E3 Q01  E12 Q11  E3 Q11  E12 Q10 (*)   E3 Q10
//     (*) before that interval's MFMAs.  (L = last K-tile of the tile, F / S = first / second of the next.)
//     So the matrix pipe keeps running through the transition, the stores of a tile are spread over two K-tiles instead of
//     arriving from all CUs at once, and what remains exposed is the hooks' VALU time (plain, un-packed VALU: another wave's
//     plain VALU does run under MFMAs on gfx950, packed fp32 does not -- tools/microbench/mfma_valu_overlap2.hip).
//   Waves 0-3 run their hooks after their MFMAs (beside waves 4-7's), waves 4-7 before theirs (beside waves 0-3's).
//
// V columns of the qkv projection (stored transposed, [b][h][d][t]) keep the token on the lane like every other tile: their
// quadrant is written TRANSPOSED into the staging image (neighbouring lanes exchange halves: DPP + v_perm_b32) and leaves as
// whole 256-B rows of V^T, so one code path and one tile list serve q, k and V tiles alike.
//
// Hazards added to those of k_vit_gemm256 (same rules: by construction, never by "it ran clean"):
//   staging RAW  E12's ds_writes are retired (lgkmcnt(0)) before the barrier that closes their interval; E3 reads in the next.
//   staging WAR  E3's ds_reads feed its own global stores, so they have returned before the barrier that closes E3's interval;
//                the next E12 writes one interval later.
//   vmcnt        E3's stores are VMEM operations in the same in-order counter as the LDS-DMA.  They are issued AFTER the
//                interval's DMA and wait, and every later wait keeps its k_vit_gemm256 count: a count of 4 still means "all
//                but the two youngest half-tiles", and only ever forces OLDER stores to have completed as well (safe).
//   bias         scalar loads (s_load, lgkmcnt) through inline asm with their own wait: no VGPR-destination load in the loop.
//
// proj / fc2 (EPI_RESIDUAL, round 5): x[m][n] += acc + b' with LayerScale folded into W and b at load.  The fp32 residual stream is 8 B
// per element of traffic where q / k / fc1 outputs are 2, and in k_vit_gemm256 that read-modify-write was all exposed (proj: 31.6 k of a
// tile's 62.8 k cycles).  Here neither half of it waits for the other or for the multiplies:
//   * READ: the old x JOINS the running sum in the middle of the main loop (kernels.h, resid_join_ktile: one definition of the order for
//     every GEMM kernel, keyed to the row's index mod 8 so that it does not depend on where an image sits in a launch).  One row class
//     per K-tile: the wave's 32 rows of the class x its 32 columns of both tile halves (4 KiB) are fetched by four LDS-DMA pieces into
//     the wave's OWN 4 KiB of the staging image (issued behind a K-tile's counted wait, so the next K-tile's wait retires them: every
//     vmcnt keeps its meaning), read back a K-tile later and added by the lane half that holds the class (32 VALU per wave).
//   * WRITE: the finished 32 x 32 blocks leave one per interval in the last K-tile and the next tile's first one: bias (a VGPR: the
//     column is the lane), 16 ds_write_b32 into the same 4 KiB as a row-major image, four lane-linear ds_read_b128, four 16-B buffer
//     stores of 8 rows x 128 B.  Everything a wave stages it reads back itself: no barrier, no cross-wave hazard; LDS operations of one
//     wave execute in order, so a unit's reads precede the next unit's writes without a wait.
//   Output slots (u = 2 q + rt; L = last K-tile, F = first of the next tile; waves 0-3 run hooks after an interval's MFMAs, waves 4-7 before):
//       waves 0-3:  L0 u0  L1 u1  L2 u2  L3 u3  F0 u4  F1 u5  F2 u6  F3 (before its MFMAs) u7
//       waves 4-7:  L1 u0  L2 u1  L3 u2  F0 u3  F1 u4  F2 u5  F3 (before its MFMAs) u6, u7
//   a block is final after interval L(q) and its registers restart from zero in F(q): every slot lies between the two.
//   What it buys (tools/microbench/gemm256_bench, profiles/r05_*): see DESIGN.md section 5, round 5.
#include "common.h"
#include "kernels.h"

namespace pio {

#ifdef PIO_ROLL_STAMPS           // diagnostic builds only (tools/microbench/gemm256_bench.hip): s_memtime per workgroup, 64 words each
__device__ unsigned long long* roll_stamps = nullptr;
#define ROLL_STAMP(i)                                                                                              \
  do {                                                                                                             \
    if (roll_stamps != nullptr && tid == 0 && (i) < 64) roll_stamps[64 * blockIdx.x + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define ROLL_STAMP(i) do { } while (0)
#endif

#ifndef PIO_ROLL_GELU_PACKED     // GELU in the hooks on packed fp32 (fewer issue cycles, no overlap with MFMAs) or plain VALU
#define PIO_ROLL_GELU_PACKED 0   // measured: no difference (fc1 at 80 images 123.9 vs 122.9 us; last K-tile 13.1 k vs 12.6 k cycles)
#endif
#ifndef PIO_ROLL_ABL             // diagnostic ablations (bit 0: bias = 0 without its scalar loads; bit 1: E3 without its global stores; bit 2: no residual bias load; bit 3: no x-in DMA)
#define PIO_ROLL_ABL 0
#endif

namespace groll {

static constexpr int TM = 256, TN = 256, TK = 64;
static constexpr int HALF = 128 * TK * 2;        // 16 KiB
static constexpr int RING_BYTES = 8 * HALF;      // 128 KiB, laid out as in vit_gemm256.hip
static constexpr int STAGE_BYTES = 128 * 256;    // 32 KiB: one quadrant in operand precision, [128 rows][256 B]
static constexpr int LDS_BYTES = RING_BYTES + STAGE_BYTES;   // 160 KiB: the whole CU
__host__ __device__ constexpr int half_off(int o, int buf, int half) { return o * 4 * HALF + buf * 2 * HALF + half * HALF; }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <typename T> struct Vec4h { typedef T type __attribute__((ext_vector_type(4))); };
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// plain (never packed) fp32 VALU: v_pk_* does not run beside another wave's MFMAs, and hipcc's SLP vectoriser packs
// neighbouring fp32 operations when it is left to choose
__device__ __forceinline__ float fma_plain(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float mul_plain(float a, float b) {
  float d;
  asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// gelu_erf (common.h) instruction by instruction on plain VALU: bit-identical
__device__ __forceinline__ float gelu_erf_plain(float v) {
  float u, q, e, hh, m, r;
  asm("v_min_f32 %0, |%1|, %2" : "=v"(u) : "v"(v), "v"(6.0811183f));
  q = fma_plain(-1.971039006e-05f, u, 6.613329563e-04f);
  q = fma_plain(q, u, -7.757447031e-03f);
  q = fma_plain(q, u, 5.296219534e-02f);
  q = fma_plain(q, u, 4.590671448e-01f);
  q = fma_plain(q, u, 1.151118979e+00f);
  e = fma_plain(u, q, 1.0f);
  // one statement: a VALU that reads a transcendental's result needs a wait state in between, which hipcc inserts for its own
  // instructions and not for inline asm (round 4: vit_attention.hip produced run-to-run differences when the scheduler happened to
  // put such a pair back to back; here it never had, by luck)
  asm("v_exp_f32 %0, -%2\n\ts_nop 0\n\tv_mul_f32 %1, %3, %0" : "=&v"(e), "=v"(hh) : "v"(e), "v"(v));
  asm("v_max_f32 %0, %1, 0" : "=v"(m) : "v"(v));
  asm("v_sub_f32 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(hh));
  return r;
}

}  // namespace groll

// the empty asm statements keep the COMPILER from moving LDS accesses (the staging image) across the barrier; they emit nothing
#define ROLL_BARRIER()                         \
  do {                                         \
    asm volatile("" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("" ::: "memory");             \
  } while (0)
#define ROLL_SB() __builtin_amdgcn_sched_barrier(0)

// One LDS-DMA half-tile (vit_gemm256.hip): AO = the tile's four per-lane A row offsets, WO = its per-lane W offset
#define ROLL_ISSUE_A(AO, buf, half, kt)                                                                                    \
  do {                                                                                                                     \
    char* const _d = smem + half_off(0, (buf), (half)) + wid * 1024;                                                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)_d, 16, AO[half][0], (kt) * (TK * 2), 0, 0);                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(_d + 8192), 16, AO[half][1], (kt) * (TK * 2), 0, 0);         \
  } while (0)
#define ROLL_ISSUE_W(WO, buf, half, kt)                                                                                    \
  do {                                                                                                                     \
    char* const _d = smem + half_off(1, (buf), (half)) + wid * 1024;                                                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)_d, 16, WO, (kt) * (TK * 2) + (half) * 128 * wrow, 0, 0);     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(_d + 8192), 16, WO, (kt) * (TK * 2) + ((half) * 128 + 64) * wrow, 0, 0); \
  } while (0)

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void k_vit_gemm_roll(const GemmArgs g) {
  using namespace groll;
  static_assert(EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_RESIDUAL, "rolling epilogue: q / k / V, fc1 + GELU, or the fp32 residual update");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const stage = smem + RING_BYTES;
  typedef typename Vec8<T>::type frag_t;
  typedef typename Vec4h<T>::type half4_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3, h = lane >> 5, r31 = lane & 31;
  ROLL_STAMP(0);

  // ---- this workgroup's tile list: ids id0, id0 + grid, ...; id -> (row tile, column tile), columns fastest; each XCD
  //      (workgroups b, b + 8, ...) gets a contiguous run of ids per round, so neighbouring tiles share their A panel in one L2
  const int ntn = g.N / TN, ntiles = ((g.M + TM - 1) / TM) * ntn, gc = (int)gridDim.x;
  int id = xcd_remap((int)blockIdx.x, gc);
  if (id >= ntiles) return;
  const int gq = gc / ntn, gr = gc - gq * ntn;           // id + gc -> (tm + gq, tn + gr) with one carry: no division in the loop
  int tmC = id / ntn, tnC = id - tmC * ntn;              // the tile being STAGED
  // bytes of an output tensor (the buffer stores' bound): q, k, vT [B][H][Tk][64] / out16 [M][N], operand precision
  const int out_bytes = EPI == EPI_RESIDUAL ? (int)((size_t)g.M * g.N * 4)
                      : EPI == EPI_GELU     ? (int)((size_t)g.M * g.N * 2) : (int)((size_t)(g.M / g.Tp) * g.H * g.Tk * 128);

  const int prow = 8 * wid + (lane >> 3);
  const uint32_t kcs = (uint32_t)(((lane & 7) ^ ((4 * wid + (lane >> 4)) & 7)) * 16);
  const int wrow = g.K * 2;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((size_t)g.M * g.lda * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (int)((size_t)g.N * g.K * 2), 0x00020000);
  uint32_t aoC[2][2], woC;      // LDS-DMA source offsets of the tile being STAGED (the next tile from the last-but-one K-tile on)
  int m0, n0;
#define ROLL_OFFSETS(AO, WO, _tm, _tn)                                                                   \
  do {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q) {        \
      int am = _tm * TM + i * 128 + q * 64 + prow;                                                       \
      am = am < g.M ? am : g.M - 1;                                                                      \
      AO[i][q] = (uint32_t)am * (uint32_t)(g.lda * 2) + kcs;                                             \
    }                                                                                                    \
    WO = (uint32_t)(_tn * TN + prow) * (uint32_t)wrow + kcs;                                             \
  } while (0)
  ROLL_OFFSETS(aoC, woC, tmC, tnC);
  m0 = tmC * TM; n0 = tnC * TN;

  const int sw7 = (lane >> 1) & 7;
  const int a_rd = (wr * 64 + r31) * 128;
  const int b_rd = half_off(1, 0, 0) + (wc * 32 + r31) * 128;
  int co[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) co[s] = ((2 * s + h) ^ sw7) << 4;

  // quadrants in phase order: 0 = (A0,B0), 1 = (A0,B1), 2 = (A1,B1), 3 = (A1,B0); [row tile rt].  Token on the lane:
  // acc[q][rt][4a+e] = C[m0 + 128 I + 64 wr + 32 rt + r31][n0 + 128 J + 32 wc + 8a + 4h + e]
  f32x16 acc[4][2];
  float res_bias[2] = {0.f, 0.f};        // EPI_RESIDUAL: this tile's bias of the lane's column, J = 0 / 1 (RES_BIAS_LOAD)
  float res_xt[EPI == EPI_RESIDUAL ? 16 : 1];   // ... and the class of x that joins next, between RES_XREAD and RES_XADD
  const int nk = g.K / TK;
  frag_t fa[2][4], fb0[4], fb1[4];

  // ---- epilogue context: the tile whose quadrants are leaving (set at the head of a tile's last K-tile)
  int e_m0 = 0, e_n0 = 0, e_img0 = 0, e_row0 = 0;
  const int per = g.Tp;
#define ROLL_SPLIT(r, b, t)                     \
  int b = e_img0, t = e_row0 + (r);             \
  while (t >= per) { t -= per; ++b; }

  // ---- hooks -------------------------------------------------------------------------------------------------------------
  // E12: bias from SGPRs (lanes 0-31 / 32-63 hold different columns: two exec-masked runs of plain v_add_f32), GELU,
  // conversion, then the quadrant's staging image:
  //   q / k / fc1 tiles   [128 token rows][256 B = 128 columns], 16-B chunk index XORed with row & 15; a lane writes its 4
  //                       consecutive columns (8 B) per register group
  //   V tiles (qkv)       TRANSPOSED, [128 columns][256 B = 128 tokens]: neighbouring lanes (tokens r, r ^ 1) exchange halves
  //                       (DPP quad_perm + v_perm_b32) so that a lane writes one column's token PAIR (4 B); odd columns
  //                       sit 64 B further (bank-conflict-free: even lanes fill banks 0-15, odd lanes 16-31)
  // bias of quadrant q (both 32-row halves) from SGPRs: lanes 0-31 / 32-63 hold different columns, two exec-masked runs
#define ROLL_BIAS(q, J, _h)                                                                                                 \
  do {                                                                                                                      \
    const float* const _bp = g.bias + e_n0 + (J) * 128 + wc * 32;                                                           \
    _Pragma("unroll") for (int ah = 0; ah < 2; ++ah) {       /* 16 columns at a time: 16 SGPRs of bias */                   \
      i32x16 s_b;                                                                                                           \
      if (PIO_ROLL_ABL & 1) { _Pragma("unroll") for (int z = 0; z < 16; ++z) s_b[z] = 0; (void)_bp; }                       \
      else asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(s_b) : "s"(_bp), "n"(64 * ah) : "memory"); \
      typedef int i32x2 __attribute__((ext_vector_type(2)));                                                                \
      typedef float f32x2 __attribute__((ext_vector_type(2)));                                                              \
      if (_h == 0) {      /* v_pk_add_f32 with an SGPR pair: two IEEE adds per instruction, half the issue slots */         \
        _Pragma("unroll") for (int a2 = 0; a2 < 2; ++a2) _Pragma("unroll") for (int e = 0; e < 4; e += 2)                   \
          _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                                                \
            f32x2 _x = {acc[q][rt][4 * (2 * ah + a2) + e], acc[q][rt][4 * (2 * ah + a2) + e + 1]};                          \
            const i32x2 _b = {s_b[8 * a2 + e], s_b[8 * a2 + e + 1]};                                                        \
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(_x) : "s"(_b));                                                   \
            acc[q][rt][4 * (2 * ah + a2) + e] = _x[0]; acc[q][rt][4 * (2 * ah + a2) + e + 1] = _x[1];                       \
          }                                                                                                                 \
      } else {                                                                                                              \
        _Pragma("unroll") for (int a2 = 0; a2 < 2; ++a2) _Pragma("unroll") for (int e = 0; e < 4; e += 2)                   \
          _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                                                \
            f32x2 _x = {acc[q][rt][4 * (2 * ah + a2) + e], acc[q][rt][4 * (2 * ah + a2) + e + 1]};                          \
            const i32x2 _b = {s_b[8 * a2 + 4 + e], s_b[8 * a2 + 4 + e + 1]};                                                \
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(_x) : "s"(_b));                                                   \
            acc[q][rt][4 * (2 * ah + a2) + e] = _x[0]; acc[q][rt][4 * (2 * ah + a2) + e + 1] = _x[1];                       \
          }                                                                                                                 \
      }                                                                                                                     \
    }                                                                                                                       \
  } while (0)
#define ROLL_E12(q, I, J)                                                                                                   \
  do {                                                                                                                      \
    int _ln = lane;   /* opaque copy: keeps the hook's address arithmetic INSIDE the hook (hoisted out of the tile loop it */ \
    asm volatile("" : "+v"(_ln));   /* would sit in registers the main loop needs and spill) */                            \
    const int _h = _ln >> 5, _r31 = _ln & 31;                                                                               \
    ROLL_BIAS(q, J, _h);                                                                                                    \
    if (EPI == EPI_QKV && e_n0 >= 2 * g.D) {          /* V tile: transposed image */                                        \
      const uint32_t _sel = (_ln & 1) ? 0x03020706u : 0x05040100u;                                                          \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) _Pragma("unroll") for (int a = 0; a < 4; ++a) {                      \
        typedef T half2_t __attribute__((ext_vector_type(2)));                                                              \
        half2_t p01, p23;                                                                                                   \
        p01[0] = (T)acc[q][rt][4 * a]; p01[1] = (T)acc[q][rt][4 * a + 1];                                                   \
        p23[0] = (T)acc[q][rt][4 * a + 2]; p23[1] = (T)acc[q][rt][4 * a + 3];                                               \
        const uint32_t o01 = __builtin_bit_cast(uint32_t, p01), o23 = __builtin_bit_cast(uint32_t, p23);                    \
        const uint32_t n01 = (uint32_t)__builtin_amdgcn_mov_dpp((int)o01, 0xB1, 0xF, 0xF, true);   /* lane ^ 1 */           \
        const uint32_t n23 = (uint32_t)__builtin_amdgcn_mov_dpp((int)o23, 0xB1, 0xF, 0xF, true);                            \
        const uint32_t wa = __builtin_amdgcn_perm(n01, o01, _sel), wb = __builtin_amdgcn_perm(n23, o23, _sel);              \
        const int tk = wr * 64 + rt * 32 + (_r31 & ~1);                 /* even token of the pair */                       \
        const int col = wc * 32 + 8 * a + 4 * _h + (_ln & 1);           /* image row of wa; wb: col + 2 */                  \
        const int pos = (2 * tk) ^ ((_ln & 1) << 6);                                                                        \
        *(uint32_t*)(stage + col * 256 + pos) = wa;                                                                         \
        *(uint32_t*)(stage + (col + 2) * 256 + pos) = wb;                                                                   \
      }                                                                                                                     \
    } else {                                                                                                                \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) _Pragma("unroll") for (int a = 0; a < 4; ++a) {                      \
        float v0 = acc[q][rt][4 * a], v1 = acc[q][rt][4 * a + 1], v2 = acc[q][rt][4 * a + 2], v3 = acc[q][rt][4 * a + 3];   \
        if constexpr (EPI == EPI_GELU) {                                                                                    \
          if (g.act == 1) { v0 = quick_gelu(v0); v1 = quick_gelu(v1); v2 = quick_gelu(v2); v3 = quick_gelu(v3); }           \
          else if (PIO_ROLL_GELU_PACKED) {                                                                                  \
            const pio_f32x2 g01 = gelu_erf2((pio_f32x2){v0, v1}), g23 = gelu_erf2((pio_f32x2){v2, v3});                     \
            v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];                                                             \
          } else { v0 = gelu_erf_plain(v0); v1 = gelu_erf_plain(v1); v2 = gelu_erf_plain(v2); v3 = gelu_erf_plain(v3); }    \
        }                                                                                                                   \
        half4_t o;                                                                                                          \
        o[0] = (T)v0; o[1] = (T)v1; o[2] = (T)v2; o[3] = (T)v3;                                                             \
        const int rq = wr * 64 + rt * 32 + _r31, c = wc * 4 + a;                                                            \
        *(half4_t*)(stage + rq * 256 + ((c ^ (rq & 15)) << 4) + 8 * _h) = o;                                               \
      }                                                                                                                     \
    }                                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                      \
  } while (0)
  // E3: every wave reads 16 image rows (4 per instruction: rows it * 32 + 4 wid + lane / 16, so the swizzle term is the same
  // for every `it`), 16 B per lane, and stores whole 256-B rows.  Branch-free: buffer stores with 32-bit byte offsets; a row
  // past M gets an offset beyond the buffer (the hardware drops the store).
#define ROLL_E3(I, J)                                                                                                       \
  do {                                                                                                                      \
    int _ln = lane;                                                                                                         \
    asm volatile("" : "+v"(_ln));                                                                                           \
    const int cc = _ln & 15, rl = wid * 4 + (_ln >> 4);                                                                     \
    constexpr uint32_t OOB = 0x80000000u;              /* every output tensor is smaller than 2 GiB (launcher) */            \
    if (EPI == EPI_QKV && e_n0 >= 2 * g.D) {           /* V tile: image row = column, chunk = 8 tokens; vT [b][D][Tk] */     \
      const int ml = (I) * 128 + 8 * cc;                                                                                    \
      int t = e_row0 + ml;                                                                                                  \
      const bool wrap = t >= per;                                                                                           \
      t -= wrap ? per : 0;                                                                                                  \
      const int bD = (e_img0 + (wrap ? 1 : 0)) * g.D + (e_n0 + (J) * 128 - 2 * g.D) + rl;                                   \
      uint32_t off = ((uint32_t)bD * (uint32_t)g.Tk + (uint32_t)t) * 2u;                                                    \
      off = e_m0 + ml < g.M ? off : OOB;                                                                                    \
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(g.vT, 0, out_bytes, 0x00020000);                                    \
      _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                                    \
        const int c = it * 32 + rl;                                                                                         \
        const u32x4 v = *(const u32x4*)(stage + c * 256 + ((cc ^ ((c & 1) << 2)) << 4));                                    \
        if (PIO_ROLL_ABL & 2) { asm volatile("" :: "v"(v)); continue; }                                                     \
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + (uint32_t)(it * 64) * (uint32_t)g.Tk, 0, 0);                    \
      }                                                                                                                     \
    } else {                                                                                                                \
      const bool isk = EPI == EPI_QKV && e_n0 >= g.D;                                                                       \
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(EPI == EPI_GELU ? g.out16 : (isk ? g.k : g.q), 0, out_bytes, 0x00020000); \
      /* GELU: out16[m][N]: (m N + n0 + 128 J) 2 + 16 cc;  q / k [b][H][Tk][64]: ((b H + head) Tk + t) 128 + 16 (cc & 7), */ \
      /* head = (n0 + 128 J - which D) / 64 + cc / 8 */                                                                     \
      const int hb = EPI == EPI_QKV ? (e_n0 + (J) * 128 - (isk ? g.D : 0)) >> 6 : 0;                                        \
      const uint32_t lc = EPI == EPI_GELU ? (uint32_t)((e_n0 + (J) * 128) * 2 + 16 * cc)                                    \
                                          : (uint32_t)(((cc >> 3) * g.Tk) * 128 + (cc & 7) * 16);                           \
      _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                                    \
        const int rq = it * 32 + rl, ml = (I) * 128 + rq, m = e_m0 + ml;                                                    \
        const u32x4 v = *(const u32x4*)(stage + rq * 256 + ((cc ^ (rq & 15)) << 4));                                        \
        if (PIO_ROLL_ABL & 2) { asm volatile("" :: "v"(v)); continue; }                                                     \
        uint32_t off;                                                                                                       \
        if constexpr (EPI == EPI_GELU) {                                                                                    \
          off = (uint32_t)m * (uint32_t)(2 * g.N) + lc;                                                                     \
        } else {                                                                                                            \
          int t = e_row0 + ml;                                                                                              \
          const bool wrap = t >= per;                                                                                       \
          t -= wrap ? per : 0;                                                                                              \
          off = (uint32_t)(((e_img0 + (wrap ? 1 : 0)) * g.H + hb) * g.Tk + t) * 128u + lc;                                  \
        }                                                                                                                   \
        off = m < g.M ? off : OOB;                                                                                          \
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);                                                           \
      }                                                                                                                     \
    }                                                                                                                       \
  } while (0)
  // ---- EPI_RESIDUAL hooks (file header); everything is private to the wave (its 4 KiB of the staging image).
  // The accumulator keeps the TOKEN ON THE REGISTER here (operands of the MFMA swapped: same products, same order):
  //     acc[q][rt][r] = C[m0 + 128 I + 64 wr + 32 rt + 8 (r >> 2) + 4 h + (r & 3)][n0 + 128 J + 32 wc + r31]
  // so that a class E = (row mod 4) + 4 J (kernels.h) is the registers r = E & 3 (mod 4) of EVERY lane in the four blocks (I, rt) of column
  // half J = E >> 2: 16 registers per lane.
  //   x-in, class E of the tile at (m0, n0): the wave's 32 rows of the class in each of its ... four blocks hold 8 (rows 8 g + 4 h' + c,
  //   g = 0 .. 3, h' = 0 / 1) -- times its 32 columns of half J: 32 row segments of 128 B = four LDS-DMA pieces p = 2 I + rt of
  //   [8 segments h', g][128 B]; lane L deposits the 16-B chunk L & 7 of segment L >> 3.  Read back by ds_read_b32 (lanes 0-31 = the 32
  //   columns of segment (h' = 0, g), lanes 32-63 of (h' = 1, g): conflict-free) one slot BEFORE the adds, so that the LDS latency passes
  //   under the interval's MFMAs or the barrier.
  //   out, unit (q, rt): bias from a VGPR (the column is the lane), 16 ds_write_b32 into a row-major [32 rows][128 B] image, four lane-linear
  //   ds_read_b128, four 16-B stores of 8 rows x 128 B.  Rows come in aligned groups of 8 and M % 8 == 0: a piece is inside M or outside.
#define RES_XDMA(E)                                                                                                         \
  do {                                                                                                                      \
    int _ln = lane;                                                                                                         \
    asm volatile("" : "+v"(_ln));                                                                                           \
    const auto _rsx = __builtin_amdgcn_make_buffer_rsrc((void*)g.x, 0, out_bytes, 0x00020000);                              \
    char* const _img = stage + wid * 4096;                                                                                  \
    const int _rl = 64 * wr + 8 * ((_ln >> 3) & 3) + 4 * (_ln >> 5) + ((E) & 3);      /* row inside a 128-row half tile, block rt = 0 */ \
    const uint32_t _cb = (uint32_t)((n0 + 128 * ((E) >> 2) + 32 * wc) * 4 + 16 * (_ln & 7));                                \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                                         \
      const int _row = m0 + 128 * (p >> 1) + 32 * (p & 1) + _rl;                                                            \
      uint32_t _off = (uint32_t)_row * (uint32_t)(g.N * 4) + _cb;                                                           \
      _off = _row < g.M ? _off : 0x80000000u;            /* past M: beyond the buffer (x < 2 GiB, launcher): nothing is fetched */ \
      if (PIO_ROLL_ABL & 8) continue;                                                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(_rsx, (lds_ptr_t)(_img + p * 1024), 16, _off, 0, 0, 0);                      \
    }                                                                                                                       \
  } while (0)
  // this tile's bias, one column per lane and J: an asm load (no compiler-visible VGPR-destination load in the loop: its wait would
  // drain the LDS-DMA queue); issued with the first x-in DMA, so the next K-tile's counted wait retires it long before its first use
#define RES_BIAS_LOAD()                                                                                                     \
  do {                                                                                                                      \
    int _ln = lane;                                                                                                         \
    asm volatile("" : "+v"(_ln));                                                                                           \
    const uint32_t _bo = (uint32_t)((n0 + 32 * wc + (_ln & 31)) * 4);                                                       \
    if (PIO_ROLL_ABL & 4) break;                                                                                            \
    /* s_nop 4: the base may have just been rebuilt by v_readlane (an SGPR spill reload), and a VMEM instruction that reads an SGPR */ \
    /* written by a VALU needs five wait states -- which hipcc inserts for its own instructions, not inside inline asm (round 5: a  */ \
    /* stale high half of the pointer = a memory fault; tests/test_isa_hazards_cpu.py scans the built library for it) */    \
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:512"                       \
                 : "=&v"(res_bias[0]), "=&v"(res_bias[1]) : "v"(_bo), "s"(g.bias) : "memory");                              \
  } while (0)
#define RES_XREAD(E)                                                                                                        \
  do {                                                                                                                      \
    int _ln = lane;                                                                                                         \
    asm volatile("" : "+v"(_ln));                                                                                           \
    const char* const _b = stage + wid * 4096 + (_ln >> 5) * 512 + (_ln & 31) * 4;                                          \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) _Pragma("unroll") for (int gq = 0; gq < 4; ++gq)                          \
      res_xt[4 * p + gq] = *(const float*)(_b + p * 1024 + gq * 128);                                                       \
  } while (0)
#define RES_XADD(E)                                                                                                         \
  do {                                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     /* RES_XREAD's values are here (and the image is free for the next class's DMA) */ \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) _Pragma("unroll") for (int gq = 0; gq < 4; ++gq)                          \
      acc[(p >> 1) == 0 ? ((E) >> 2) : 3 - ((E) >> 2)][p & 1][4 * gq + ((E) & 3)] += res_xt[4 * p + gq];                    \
  } while (0)
#define RES_OUT(U)                                                                                                          \
  do {                                                                                                                      \
    constexpr int _q = (U) >> 1, _rt = (U) & 1, _I = _q >> 1, _J = (_q == 1 || _q == 2) ? 1 : 0;                             \
    int _ln = lane;                                                                                                         \
    asm volatile("" : "+v"(_ln));                                                                                           \
    const auto _rsx = __builtin_amdgcn_make_buffer_rsrc((void*)g.x, 0, out_bytes, 0x00020000);                              \
    char* const _img = stage + wid * 4096;                                                                                  \
    const int _rowb = e_m0 + 128 * _I + 64 * wr + 32 * _rt;                                                                 \
    const uint32_t _sbase = (uint32_t)_rowb * (uint32_t)(g.N * 4) + (uint32_t)((e_n0 + 128 * _J + 32 * wc) * 4);            \
    const uint32_t _vl = (uint32_t)(_ln >> 3) * (uint32_t)(g.N * 4) + (uint32_t)((_ln & 7) << 4);                           \
    char* const _w = _img + (_ln >> 5) * 512 + (_ln & 31) * 4;                                                              \
    const float _bj = res_bias[_J];                                                                                         \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) *(float*)(_w + (8 * (r >> 2) + (r & 3)) * 128) = acc[_q][_rt][r] + _bj;   \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                                      \
      const u32x4 v = *(const u32x4*)(_img + it * 1024 + 16 * _ln);                                                         \
      if (PIO_ROLL_ABL & 2) { asm volatile("" :: "v"(v)); continue; }                                                       \
      if (_rowb + 8 * it < g.M)                                                                                             \
        __builtin_amdgcn_raw_buffer_store_b128(v, _rsx, _vl, _sbase + (uint32_t)(8 * it) * (uint32_t)(g.N * 4), 0);         \
    }                                                                                                                       \
  } while (0)
#define RES_U(U) ((U) < 0 ? 0 : ((U) > 7 ? 7 : (U)))
  // One slot of the schedule: interval k of a K-tile at position POS; PRE = 1 before the interval's MFMAs, 0 at the group's usual place,
  // 2 behind the MFMAs of interval 3 (waves 4-7 only).  U: K-tile 1 carries 0, the K-tiles 2 .. 9 of a tile 1 .. 8, every other one -1.
  // Class e joins after K-tile e + 1 (kernels.h):
  //   waves 0-3 (hooks after MFMAs): K-tile e + 1, interval 3: reads before its MFMAs, adds behind them, then the DMA of class e + 1;
  //                                  the DMA of class 0 (and the bias) in the last interval of K-tile 0, behind the previous tile's last unit
  //   waves 4-7 (hooks before MFMAs): reads behind the MFMAs of K-tile e + 1's interval 3, adds in the FIRST interval of K-tile e + 2 (before
  //                                  its MFMAs), then the DMA of class e + 1; the DMA of class 0 (and the bias) in the first interval of K-tile 1
  // either way between the MFMAs of K-tile e + 1 and those of K-tile e + 2 on every quadrant, and a DMA is retired by the counted wait
  // of the K-tile that follows its issue (it is older than that K-tile's A0 / B0 requests), before the reads of its class are issued.
#define RES_SLOT(POS, k, PRE, GRP, U)                                                                                       \
  do {                                                                                                                      \
    if constexpr (EPI == EPI_RESIDUAL) {                                                                                    \
      if constexpr ((GRP) == 0) {                                                                                           \
        if constexpr ((POS) == 4 && (PRE) == 0) RES_OUT(k);                                                                 \
        if constexpr ((POS) == 0 && (PRE) == 0 && (k) <= 2) { if (has_prev) RES_OUT(4 + (k)); }                             \
        if constexpr ((POS) == 0 && (PRE) == 1) { if (has_prev) RES_OUT(7); }                                               \
        if constexpr ((POS) == 0 && (PRE) == 0 && (k) == 3) { RES_XDMA(0); RES_BIAS_LOAD(); }                               \
        if constexpr (((POS) == 1 || (POS) == 2) && (U) >= 0 && (U) <= 7 && (k) == 3) {                                     \
          if constexpr ((PRE) == 1) RES_XREAD(RES_U(U));                                                                    \
          if constexpr ((PRE) == 0) {                                                                                       \
            RES_XADD(RES_U(U));                                                                                             \
            if constexpr ((U) < 7) RES_XDMA(RES_U(U) + 1);                                                                  \
          }                                                                                                                 \
        }                                                                                                                   \
      } else {                                                                                                              \
        if constexpr ((POS) == 4 && (PRE) == 0 && (k) >= 1) RES_OUT(((k) - 1) & 7);                                         \
        if constexpr ((POS) == 0 && (PRE) == 0 && (k) <= 2) { if (has_prev) RES_OUT(3 + (k)); }                             \
        if constexpr ((POS) == 0 && (PRE) == 1) { if (has_prev) { RES_OUT(6); RES_OUT(7); } }                               \
        if constexpr ((POS) == 1 && (PRE) == 0 && (k) == 0) { RES_XDMA(0); RES_BIAS_LOAD(); }                               \
        if constexpr (((POS) == 1 || (POS) == 2) && (U) >= 0 && (U) <= 7 && (PRE) == 2) RES_XREAD(RES_U(U));                \
        if constexpr ((POS) == 2 && (U) >= 1 && (PRE) == 0 && (k) == 0) {                                                   \
          RES_XADD(RES_U((U) - 1));                                                                                         \
          if constexpr ((U) <= 7) RES_XDMA(RES_U(U));                                                                       \
        }                                                                                                                   \
      }                                                                                                                     \
    }                                                                                                                       \
  } while (0)
  // the eight hook slots of the table in the file header
#define ROLL_HOOK(slot)                                                                                                     \
  do {                                                                                                                      \
    if constexpr ((slot) == 1) ROLL_E12(0, 0, 0);                                                                           \
    if constexpr ((slot) == 2) ROLL_E3(0, 0);                                                                               \
    if constexpr ((slot) == 3) ROLL_E12(1, 0, 1);                                                                           \
    if constexpr ((slot) == 4) ROLL_E3(0, 1);                                                                               \
    if constexpr ((slot) == 5) ROLL_E12(2, 1, 1);                                                                           \
    if constexpr ((slot) == 6) ROLL_E3(1, 1);                                                                               \
    if constexpr ((slot) == 7) ROLL_E12(3, 1, 0);                                                                           \
    if constexpr ((slot) == 8) ROLL_E3(1, 0);                                                                               \
  } while (0)

#define ROLL_READ_A(buf, i)                                                                              \
  _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) _Pragma("unroll") for (int s = 0; s < 4; ++s)        \
      fa[rt][s] = *(const frag_t*)(smem + half_off(0, (buf), (i)) + rt * 4096 + a_rd + co[s])
#define ROLL_READ_B(dst, buf, j)                                                                         \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) dst[s] = *(const frag_t*)(smem + half_off(0, (buf), (j)) + b_rd + co[s])
  // ZERO: the first K-tile of a tile starts its quadrant from 0 (the old contents have left through the hooks)
#define ROLL_MMA(q, fb, ZERO)                                                                            \
  do {                                                                                                   \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {     \
      f32x16 _c = acc[q][rt];                                                                            \
      if constexpr (ZERO) { if (s == 0) { _Pragma("unroll") for (int r = 0; r < 16; ++r) _c[r] = 0.f; } } \
      if constexpr (EPI == EPI_RESIDUAL) acc[q][rt] = mfma32(fa[rt][s], fb[s], _c);      /* token on the register (RES_* hooks) */ \
      else acc[q][rt] = mfma32(fb[s], fa[rt][s], _c);                                    /* token on the lane */     \
    }                                                                                                    \
  } while (0)

  // K-tile `t` of the stream in buffer BUF.  POS: 0 = first of a tile, 1 = second, 2 = middle, 3 = last but one, 4 = last.
  // The LDS-DMA of an interval stages (k_vit_gemm256, schedule 1)  0: A1(t+1)  1: A0(t+2)  2: B0(t+2) + wait  3: B1(t+2);
  // past the end of the tile these are the first K-tiles of the NEXT tile (offsets aoN / woN), if there is one.
// The current tile's offsets are last used by DMA0 of its last-but-one K-tile (A1 of the last K-tile); right after it
// aoC / woC are recomputed for the next tile, so one set of five registers serves both.
#define ROLL_DMA0(POS, t, BUF)                                                                           \
  do {                                                                                                   \
    if constexpr ((POS) == 4) { if (has_next) ROLL_ISSUE_A(aoC, (BUF) ^ 1, 1, 0); }                      \
    else ROLL_ISSUE_A(aoC, (BUF) ^ 1, 1, (t) + 1);                                                       \
    if constexpr ((POS) == 3) {                                                                          \
      if (has_next) {                                                                                    \
        tmC += gq; tnC += gr;                                                                            \
        if (tnC >= ntn) { tnC -= ntn; ++tmC; }                                                           \
        ROLL_OFFSETS(aoC, woC, tmC, tnC);                                                                \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)
#define ROLL_DMA1(POS, t, BUF)                                                                           \
  do {                                                                                                   \
    if constexpr ((POS) >= 3) { if (has_next) ROLL_ISSUE_A(aoC, BUF, 0, (POS) - 3); }                    \
    else ROLL_ISSUE_A(aoC, BUF, 0, (t) + 2);                                                             \
  } while (0)
#define ROLL_DMA2(POS, t, BUF)                                                                           \
  do {                                                                                                   \
    if constexpr ((POS) >= 3) { if (has_next) ROLL_ISSUE_W(woC, BUF, 0, (POS) - 3); }                    \
    else ROLL_ISSUE_W(woC, BUF, 0, (t) + 2);                                                             \
  } while (0)
#define ROLL_DMA3(POS, t, BUF)                                                                           \
  do {                                                                                                   \
    if constexpr ((POS) >= 3) { if (has_next) ROLL_ISSUE_W(woC, BUF, 1, (POS) - 3); }                    \
    else ROLL_ISSUE_W(woC, BUF, 1, (t) + 2);                                                             \
  } while (0)
#define ROLL_WAIT2(POS)                                                                                  \
  do {                                                                                                   \
    if ((POS) < 3 || has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                          \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
  } while (0)
  // the very first K-tile of the workgroup: A1 of K-tile 0 is still in flight after the prologue's wait
#define ROLL_FIRSTWAIT(POS) do { if constexpr ((POS) == 0) { if (!has_prev) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); } } while (0)
  // hooks of interval k (0..3) of a K-tile at position POS: slots L1..L3 in the last K-tile, F0..F3 in the first, S0 in the second
#define ROLL_HOOKS(POS, k, GRP, U)                                                                       \
  do {                                                                                                   \
    if constexpr (EPI == EPI_RESIDUAL) {                                                                 \
      RES_SLOT(POS, k, 0, GRP, U);                                                                       \
    } else {                                                                                             \
      if constexpr ((POS) == 4 && (k) >= 1) ROLL_HOOK((k));                                              \
      if constexpr ((POS) == 0 && (k) <= 2) { if (has_prev) ROLL_HOOK(4 + (k)); }                        \
      if constexpr ((POS) == 1 && (k) == 0) { if (has_prev) ROLL_HOOK(8); }                              \
    }                                                                                                    \
  } while (0)
  // slot 7 (E12 of Q10) comes BEFORE the MFMAs of interval F3, which overwrite Q10
#define ROLL_PREHOOK3(POS, GRP, U)                                                                       \
  do {                                                                                                   \
    if constexpr (EPI == EPI_RESIDUAL) {                                                                 \
      RES_SLOT(POS, 3, 1, GRP, U);                                                                       \
    } else {                                                                                             \
      if constexpr ((POS) == 0) { if (has_prev) ROLL_HOOK(7); }                                          \
    }                                                                                                    \
  } while (0)

  // waves 0-3: MFMAs of interval k, then the fragment reads of interval k + 1, the LDS-DMA, the hooks
#define ROLL_KTILE_G0(POS, t, BUF, U)                                                                    \
  do {                                                                                                   \
    ROLL_MMA(0, fb0, (POS) == 0); ROLL_SB(); ROLL_READ_B(fb1, BUF, 1); ROLL_DMA0(POS, t, BUF); ROLL_FIRSTWAIT(POS); \
    ROLL_SB(); ROLL_HOOKS(POS, 0, 0, U); ROLL_BARRIER();                                                 \
    ROLL_MMA(1, fb1, (POS) == 0); ROLL_SB(); ROLL_READ_A(BUF, 1); ROLL_DMA1(POS, t, BUF);                \
    ROLL_SB(); ROLL_HOOKS(POS, 1, 0, U); ROLL_BARRIER();                                                 \
    ROLL_MMA(2, fb1, (POS) == 0); ROLL_SB(); ROLL_DMA2(POS, t, BUF); ROLL_WAIT2(POS);                    \
    ROLL_SB(); ROLL_HOOKS(POS, 2, 0, U); ROLL_BARRIER();                                                 \
    ROLL_PREHOOK3(POS, 0, U); ROLL_SB();                                                                 \
    ROLL_MMA(3, fb0, (POS) == 0); ROLL_SB();                                                             \
    if ((POS) < 4 || has_next) { ROLL_READ_A((BUF) ^ 1, 0); ROLL_READ_B(fb0, (BUF) ^ 1, 0); }            \
    ROLL_DMA3(POS, t, BUF);                                                                              \
    ROLL_SB(); ROLL_HOOKS(POS, 3, 0, U); ROLL_BARRIER();                                                 \
  } while (0)
  // waves 4-7: the hooks, the fragment reads of interval k, the LDS-DMA, then the MFMAs of interval k
#define ROLL_KTILE_G1(POS, t, BUF, U)                                                                    \
  do {                                                                                                   \
    ROLL_HOOKS(POS, 0, 1, U); ROLL_SB();                                                                 \
    ROLL_READ_A(BUF, 0); ROLL_READ_B(fb0, BUF, 0); ROLL_DMA0(POS, t, BUF); ROLL_SB(); ROLL_MMA(0, fb0, (POS) == 0); \
    ROLL_FIRSTWAIT(POS); ROLL_BARRIER();                                                                 \
    ROLL_HOOKS(POS, 1, 1, U); ROLL_SB();                                                                 \
    ROLL_READ_B(fb1, BUF, 1); ROLL_DMA1(POS, t, BUF); ROLL_SB(); ROLL_MMA(1, fb1, (POS) == 0); ROLL_BARRIER(); \
    ROLL_HOOKS(POS, 2, 1, U); ROLL_SB();                                                                 \
    ROLL_READ_A(BUF, 1); ROLL_DMA2(POS, t, BUF); ROLL_SB(); ROLL_MMA(2, fb1, (POS) == 0); ROLL_WAIT2(POS); ROLL_BARRIER(); \
    ROLL_PREHOOK3(POS, 1, U); ROLL_HOOKS(POS, 3, 1, U); ROLL_SB();                                       \
    ROLL_DMA3(POS, t, BUF); ROLL_SB(); ROLL_MMA(3, fb0, (POS) == 0); ROLL_SB(); RES_SLOT(POS, 3, 2, 1, U); ROLL_BARRIER(); \
  } while (0)

  // ---- the tile walk of one wave group
#define ROLL_WALK(KTILE, GRP)                                                                            \
  do {                                                                                                   \
    bool has_prev = false;                                                                               \
    int _ti = 0; (void)_ti;                                                                               \
    for (;;) {                                                                                           \
      const int nid = id + gc;                                                                           \
      const bool has_next = nid < ntiles;                                                                \
      ROLL_STAMP(2 + 5 * _ti);                                                                           \
      KTILE(0, 0, 0, -1);                                                                                \
      KTILE(1, 1, 1, 0);                                                                                 \
      ROLL_STAMP(3 + 5 * _ti);                                                                           \
      if constexpr (EPI == EPI_RESIDUAL) {      /* K-tiles 1 .. 9: the old x joins, one row class per K-tile (nk >= 12: launcher) */ \
        KTILE(2, 2, 0, 1); KTILE(2, 3, 1, 2); KTILE(2, 4, 0, 3); KTILE(2, 5, 1, 4);                      \
        KTILE(2, 6, 0, 5); KTILE(2, 7, 1, 6); KTILE(2, 8, 0, 7); KTILE(2, 9, 1, 8);                      \
        for (int t = 10; t < nk - 2; t += 2) { KTILE(2, t, 0, -1); KTILE(2, t + 1, 1, -1); }             \
      } else {                                                                                           \
        for (int t = 2; t < nk - 2; t += 2) { KTILE(2, t, 0, -1); KTILE(2, t + 1, 1, -1); }              \
      }                                                                                                  \
      ROLL_STAMP(4 + 5 * _ti);                                                                           \
      KTILE(3, nk - 2, 0, -1);                                                                           \
      ROLL_STAMP(5 + 5 * _ti);                                                                           \
      e_m0 = m0; e_n0 = n0; e_img0 = m0 / per; e_row0 = m0 - e_img0 * per;                               \
      KTILE(4, nk - 1, 1, -1);                                                                           \
      ROLL_STAMP(6 + 5 * _ti); ++_ti;                                                                    \
      has_prev = true;                                                                                   \
      if (!has_next) break;                                                                              \
      id = nid;                                                                                          \
      m0 = tmC * TM; n0 = tnC * TN;                                                                      \
    }                                                                                                    \
    /* drain: the last tile's slots F0 .. S0 with nothing left to multiply */                            \
    if constexpr (EPI == EPI_RESIDUAL) {                                                                 \
      if constexpr ((GRP) == 1) RES_OUT(3);                                                              \
      RES_OUT(4); RES_OUT(5); RES_OUT(6); RES_OUT(7);                                                    \
    } else {                                                                                             \
      ROLL_HOOK(4); ROLL_BARRIER();                                                                      \
      ROLL_HOOK(5); ROLL_BARRIER();                                                                      \
      ROLL_HOOK(6); ROLL_BARRIER();                                                                      \
      ROLL_HOOK(7); ROLL_BARRIER();                                                                      \
      ROLL_HOOK(8);                                                                                      \
    }                                                                                                    \
    ROLL_STAMP(2 + 5 * _ti);                                                                             \
  } while (0)

  // ---- prologue of the FIRST tile only (k_vit_gemm256, schedule 1)
  ROLL_ISSUE_A(aoC, 0, 0, 0);
  ROLL_ISSUE_W(woC, 0, 0, 0);
  ROLL_ISSUE_W(woC, 0, 1, 0);
  ROLL_ISSUE_A(aoC, 0, 1, 0);
  ROLL_ISSUE_A(aoC, 1, 0, 1);
  ROLL_ISSUE_W(woC, 1, 0, 1);
  ROLL_ISSUE_W(woC, 1, 1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // A0, B0, B1 of K-tile 0 have landed
  ROLL_BARRIER();
  ROLL_STAMP(1);

  if (wr == 0) {
    ROLL_READ_A(0, 0); ROLL_READ_B(fb0, 0, 0);
    ROLL_WALK(ROLL_KTILE_G0, 0);
  } else {
    ROLL_WALK(ROLL_KTILE_G1, 1);
  }
}

#undef ROLL_WALK
#undef ROLL_KTILE_G0
#undef ROLL_KTILE_G1
#undef ROLL_PREHOOK3
#undef ROLL_HOOKS
#undef ROLL_FIRSTWAIT
#undef ROLL_WAIT2
#undef ROLL_DMA0
#undef ROLL_DMA1
#undef ROLL_DMA2
#undef ROLL_DMA3
#undef ROLL_MMA
#undef ROLL_READ_A
#undef ROLL_READ_B
#undef ROLL_HOOK
#undef ROLL_E3
#undef ROLL_E12
#undef ROLL_BIAS
#undef RES_SLOT
#undef RES_U
#undef RES_OUT
#undef RES_XADD
#undef RES_XREAD
#undef RES_XDMA
#undef RES_BIAS_LOAD
#undef ROLL_SPLIT
#undef ROLL_OFFSETS

// The rolling kernel serves the GEMMs whose tiles outnumber the CUs enough to give every workgroup a second tile to hide
// the first one's epilogue under (qkv without the fp32 capture, fc1): from 1.5 tiles per CU on; and proj / fc2 (EPI_RESIDUAL) at every
// size the 256 x 256 tile is worth taking: the old x joins the sum inside the main loop even when a workgroup has one tile.
bool vit_gemm_roll_fits(GemmEpilogue epi, const GemmArgs& a) {
  using namespace groll;
  if (epi == EPI_RESIDUAL) {
    if (a.M <= 0 || a.M % 8 != 0 || a.N % TN != 0 || a.K % (2 * TK) != 0 || a.K / TK < 12 || a.lda % 8 != 0) return false;
    if ((size_t)a.M * a.lda * 2 >= ((size_t)1 << 31) || (size_t)a.N * a.K * 2 >= ((size_t)1 << 31)) return false;
    return (size_t)a.M * a.N * 4 < ((size_t)1 << 31);      // 32-bit byte offsets into x
  }
  if (epi != EPI_QKV && epi != EPI_GELU) return false;
  if (epi == EPI_QKV && (a.qkv_last != nullptr || a.D % TN != 0 || a.Tp % 8 != 0 || a.N != 3 * a.D)) return false;
  if (a.M <= 0 || a.N % TN != 0 || a.K % (2 * TK) != 0 || a.K / TK < 4 || a.lda % 8 != 0) return false;
  if ((size_t)a.M * a.lda * 2 >= ((size_t)1 << 31) || (size_t)a.N * a.K * 2 >= ((size_t)1 << 31)) return false;
  if (a.Tp < TM) return false;                      // a tile's rows span at most two images
  if ((size_t)a.M * a.N * 2 >= ((size_t)1 << 31)) return false;     // 32-bit store offsets, "past the end" = bit 31
  return true;
}

static constexpr int ROLL_GRID = 256;               // one workgroup per CU (the kernel takes all 160 KiB of LDS)

template <typename T, int EPI>
static hipError_t launch_roll_one(const GemmArgs& a, hipStream_t s) {
  using namespace groll;
  static bool attr_set[64] = {};                     // function attributes are per device
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (!attr_set[dev & 63]) {
    e = hipFuncSetAttribute((const void*)k_vit_gemm_roll<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  // As many workgroups as give every one of them the same number of tiles: the launch takes ceil(tiles / 256) rounds either way, and
  // the compute units a balanced grid leaves free (7 of 256 at 80 images: 747 and 996 tiles are 3 and 4 rounds of 249) serve the
  // decode kernels of the other streams, which otherwise find none while a persistent GEMM is resident.  Measured: isolated qkv
  // 81.1 against 82.2 us, fc1 123.5 against 123.9; pipelined throughput unchanged (8.19 against 8.20 k captions/s).
  static const bool balanced = [] { const char* e = getenv("PIO_ROLL_BALANCED"); return e == nullptr || atoi(e) != 0; }();
  // PIO_ROLL_MAX_GRID (diagnostic, round 5): fewer persistent workgroups than CUs, so that two streams' GEMMs can sit side by side on
  // the chip (tools/microbench/r5_two_streams.sh)
  static const int max_grid = [] { const char* e = getenv("PIO_ROLL_MAX_GRID"); const int v = e ? atoi(e) : 0; return v > 0 && v < ROLL_GRID ? v : ROLL_GRID; }();
  const int ntiles = ceil_div(a.M, TM) * (a.N / TN);
  const int rounds = ceil_div(ntiles, max_grid);
  const int grid = balanced ? ceil_div(ntiles, rounds) : max_grid;
  hipLaunchKernelGGL((k_vit_gemm_roll<T, EPI>), dim3(grid), dim3(512), LDS_BYTES, s, a);
  return hipGetLastError();
}

hipError_t launch_vit_gemm_roll(OperandType t, GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
  if (!vit_gemm_roll_fits(epi, a)) return hipErrorInvalidValue;
  if (epi == EPI_QKV) return t == OP_F16 ? launch_roll_one<f16, EPI_QKV>(a, s) : launch_roll_one<bf16, EPI_QKV>(a, s);
  if (epi == EPI_RESIDUAL) return t == OP_F16 ? launch_roll_one<f16, EPI_RESIDUAL>(a, s) : launch_roll_one<bf16, EPI_RESIDUAL>(a, s);
  return t == OP_F16 ? launch_roll_one<f16, EPI_GELU>(a, s) : launch_roll_one<bf16, EPI_GELU>(a, s);
}

}  // namespace pio
