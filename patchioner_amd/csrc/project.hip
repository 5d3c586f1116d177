// Memory-bank projection: Im2TxtProjector.project (P/src/decap/im2txtprojection/im2txtprojection.py:353-385)
// as ONE pass over the bank with an online softmax (the reference re-normalises the bank, takes
// sim = q.bank_n^T, softmaxes sim/T and multiplies by the raw bank: three passes + a bank-sized temp).
//
//   sim[n][r]  = <q_n/|q_n|, bank[r]> * inv_norm[r]
//   out[n][:]  = sum_r softmax_r(sim[n][r] / T) * bank[r][:]          (optionally L2-normalised)
//
// HBM-bound for <= 16 queries: the bank (M x D fp32, 1.8 GB for the COCO bank) is read exactly once.
// All arithmetic is exact fp32 on v_mfma_f32_16x16x4_f32 (T = 0.01 multiplies cosine error by 100, so
// reduced-precision operands are not used here).
//
// One 256-thread workgroup per CU walks a contiguous slab of rows in tiles of 16 rows, register-staged
// into double-buffered LDS (row stride D*4+16 B => conflict-free for both access patterns below).  The 4
// waves split D:  wave w owns channels [w*D/4, (w+1)*D/4).
//   GEMM1  S[row][n]   partial over the wave's channels (A = bank rows, B = queries, both ds_read_b128),
//          summed across the 4 waves through LDS (bitwise identical in every wave).
//   online softmax     query n sits on lane&15, rows on (lane>>4, reg): tile max / sum = 4 regs + 2 shuffles.
//   GEMM2  Acc[n][d] += P[row][n] * bank[row][d]: the S accumulator register t IS the A operand of the
//          t-th MFMA when k-slot kq means row 4*kq+t, so P never moves; B = bank[4kq+t][d0+lane&15] (ds_read_b32).
// Each workgroup ends with a partial (max, sum, Acc) that k_project_combine merges.
#include "common.h"
#include "kernels.h"

namespace pio {

#ifndef PIO_PROJECT_Q32       // 32 queries per bank pass when more than 16 are left
#define PIO_PROJECT_Q32 1
#endif
#ifndef PIO_PROJECT_Q48       // 48 queries per bank pass when more than 32 are left (D = 768)
#define PIO_PROJECT_Q48 1
#endif
#ifndef PIO_PROJECT_RT32      // 16-row tiles per loop iteration of the 32-query form
#define PIO_PROJECT_RT32 2
#endif
#ifndef PIO_PROJECT_NT
#define PIO_PROJECT_NT 1
#endif
#ifndef PIO_PROJECT_SPLIT16   // 1: the 16-query pass takes the split-fp16 form as well
#define PIO_PROJECT_SPLIT16 1
#endif
__device__ __forceinline__ float4 ld_stream4(const float* p) {
  if (PIO_PROJECT_NT) {
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
  }
  return *(const float4*)p;
}


typedef __attribute__((address_space(3))) void* pr_lds_ptr_t;

#ifdef PIO_PROJ_STAMPS            // diagnostic builds only: cycles per phase of k_project2's loop, summed over a workgroup's tiles by
__device__ unsigned long long* proj_stamps = nullptr;   // wave 0 -> [workgroup][8] (never read by the kernel)
#define PROJ_STAMP(i)                                                                                          \
  do {                                                                                                         \
    const unsigned long long _now = __builtin_readcyclecounter();                                              \
    st_acc[i] += _now - st_last; st_last = _now;                                                               \
  } while (0)
#else
#define PROJ_STAMP(i) do { } while (0)
#endif

static constexpr int PR_ROWS = 16;   // bank rows per tile
static constexpr int PR_Q = 16;      // queries per pass

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// L2-normalise rows in place (reference line 368 mutates the caller's tensor)
__global__ __launch_bounds__(256) void k_l2norm_rows(float* x, int D) {
  __shared__ float red[4];
  float* r = x + (size_t)blockIdx.x * D;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) s += r[i] * r[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  for (int i = threadIdx.x; i < D; i += 256) r[i] = r[i] / nrm;
}

__global__ __launch_bounds__(256) void k_row_inv_norm(const float* __restrict__ bank, int64_t M, int D, float* inv) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4* r = (const float4*)(bank + row * D);
  float s = 0.f;
  for (int c = threadIdx.x & 63; c < (D >> 2); c += 64) {
    const float4 v = r[c];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) inv[row] = 1.0f / sqrtf(s);
}

// max |x| over a buffer, as the bit pattern of a non-negative float (ordering of non-negative floats = ordering of their bits)
__global__ __launch_bounds__(256) void k_abs_max(const float* __restrict__ x, int64_t n, uint32_t* out) {
  uint32_t m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t b = __float_as_uint(x[i]) & 0x7FFFFFFFu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(m, o); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// The bank as split fp16 operands, built once when the bank is loaded: out[row] = [hi(x S) for the D channels | lo(x S) for the D
// channels], hi = fp16(x S) (round to nearest), lo = fp16(x S - hi); S = bank_scale, a power of two (see k_project2's split form).
__global__ __launch_bounds__(256) void k_split_bank(const float* __restrict__ bank, int64_t M, int D, float S, _Float16* __restrict__ out) {
  const int64_t pairs = M * (int64_t)(D / 2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / (D / 2);
    const int d = (int)(i - row * (D / 2)) * 2;
    const float2 x = *(const float2*)(bank + row * D + d);
    const float x0 = x.x * S, x1 = x.y * S;
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    *(h2*)(out + row * 2 * D + d) = (h2){h0, h1};
    *(h2*)(out + row * 2 * D + D + d) = (h2){l0, l1};
  }
}

// ---- round 2: the same pass, software-pipelined -------------------------------------------------------------------
// k_project spends a 16-row tile as  [GEMM1 -> partial S to LDS] barrier [sum, soft-max, GEMM2] barrier [registers -> LDS]
// barrier: three barriers, a register-staged single buffer, and the MFMA pipe idle around each of them (PMC: 39-49 % busy;
// 0.83 ms per 32-query pass against 0.37 ms of fp32 MFMA issue).  k_project2 keeps every arithmetic operation and its order
// -- the outputs are bit-identical, tests/test_gpu_parity.py compares the two -- and changes the plumbing:
//   * two tile buffers filled by LDS-DMA (global_load_lds, 16 B per lane, non-temporal), no staging registers, no store phase;
//   * iteration t = GEMM2(t) | wait own DMA of tile t+1, barrier A (tile t free, tile t+1 complete) | DMA(t+2) into tile t's
//     buffer | GEMM1(t+1) -> partial S | barrier B | sum + soft-max(t+1).  Two barriers per tile, the load of tile t+2 has a
//     whole iteration to land, GEMM2(t) / GEMM1(t+1) issue back to back (96 MFMAs per wave);
//   * the rescale of the accumulators (4 lane exchanges + D/16 multiplies) is skipped while no query's running maximum moved
//     (multiplying by exp(0) = 1 is the identity, so skipping it changes no bit); the row-sum exchanges of the soft-max wait
//     until GEMM2's MFMAs have been issued; inv_norm of tile t+1 is fetched under GEMM2(t).
// Measured (one MI355X, 591 753 x 768 bank, kernel time under rocprofv3 --pmc, tools/microbench/pmc_project.sh): 32 queries
// 650 us against 779 us (MFMA busy 61 % against 50 % of the SIMD cycles at 2.23 GHz; 399 us is the issue time of the pass's
// 192 fp32 MFMAs per tile and SIMD); whole call 0.67-0.71 against 0.84-0.88 ms; 16 queries 0.53-0.57 against 0.55-0.59 ms.
// The time per tile does not depend on where the bank lives (4.5 us per tile and CU from 100 MB = cache-resident to 1.8 GB,
// tools/microbench/project_latency_probe.py): the pass is bound by its own serial chain, not by HBM.
// Tried on top and dropped: the two query groups of a workgroup half an iteration apart (group 0 soft-max + GEMM2 while
// group 1 runs GEMM1 and vice versa, three tile buffers, bit-identical): 726 us, MFMA busy 55 % -- the SIMD issues the older
// wave's MFMAs first, so the groups do not interleave the way the schedule wants.
//
// ---- round 3: both GEMMs on split fp16 operands (SPLIT = true; what every pass takes unless the context is in the exact mode) ------
// The fp32 matrix instructions run at the fp32 VECTOR rate (64 FLOP/clk/SIMD) and nothing of the wave's VALU work overlaps them
// (stamps: tools/microbench/project_stamps.hip), so a 32-query pass was bound by its 192 fp32 MFMAs of 32 cycles per tile and SIMD
// (399 us of 650) plus the soft-max beside them.  An fp16 MFMA does 16x the work per cycle, but T = 0.01 multiplies a cosine's error by
// 100, so the operands keep 22 significant bits: every factor is split into TWO fp16 values, x S = hi + lo with hi = fp16(x S) and
// lo = fp16(x S - hi) (S a power of two: lo is a normal fp16 for every |x| >= 2^-17 max|bank|), and the products (Ah + Al)(Bh + Bl)
// are fp16 MFMAs with fp32 accumulation: fp16 x fp16 is exact in fp32, what is lost is below 2^-22 of each factor -- against an fp64
// evaluation the pass is as close as the fp32 form was (2.1e-7 vs 3.9e-7 max on unit-norm outputs, tools/microbench/project_time.py).
// The BANK IS SPLIT ONCE, when it is loaded (k_split_bank: [row][hi plane D | lo plane D] fp16 of bank * bank_scale, the same 4 D
// bytes per row; the fp32 bank stays for the exact mode and the top-k similarities), so the pass has NO conversion work:
//   * the LDS-DMA image of a tile is unchanged (3 x 1 KiB per row), row stride 4 D + 32 B (8 dwords mod 64);
//   * GEMM1: a lane's A fragment = 8 consecutive channels of ITS row: one ds_read_b128 from the hi plane, one from the lo plane
//     (conflict-free with that stride); per 32 channels Bh.qh on one accumulator and Bh.ql + Bl.qh on another (Bl.ql is 2^-22 of
//     the result); the query fragments (q 2^14 = qh + ql) are split once per pass;
//   * GEMM2: the B fragment needs a COLUMN of the tile (4 rows of one channel, hi and lo): two ds_read_b64_tr_b16 (gfx950's
//     transposing LDS read, 4 rows x 16 channels per 16 lanes, conflict-free with the same stride); the 32 k-slots of one
//     v_mfma_f32_16x16x32_f16 hold the tile's 16 rows twice, [Bh | Bl], against [Ph | Ph] and then [Pl | Pl] (P 2^14 = Ph + Pl):
//     a lane's own four P values are its A fragment, no exchange;
//   * powers of two scale exactly: they are divided out of the similarities (inv_norm * unscale) and of the stored partials.
template <int D, int NQG, bool SPLIT>
__global__ __launch_bounds__(256 * NQG, 1) void k_project2(const float* __restrict__ bank, const float* __restrict__ inv_norm,
                                                           int64_t M, const float* __restrict__ q, int N, int q0,
                                                           float temperature, float* part_acc, float* part_ml, int parts,
                                                           int slab_unit, float bank_scale) {
  constexpr int STRIDE = SPLIT ? D + 8 : D + 4;  // floats; +16 B (exact) / +32 B (split: see above) skews rows across the 64 banks
  constexpr int DW = D / 4;                      // channels per wave
  constexpr int NW = 4 * NQG;                    // waves
  constexpr int NQ = PR_Q * NQG;                 // queries per pass
  constexpr int OPR = (D * 4 + 1023) / 1024;     // LDS-DMA operations per bank row (1 KiB each; the last one may be partial)
  constexpr int OPS = PR_ROWS * OPR;             // ... per tile
  static_assert(D % 64 == 0 && OPS % NW == 0, "D");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NBUF = 2;                         // tile buffers
  float* s_tile = lds;                            // [NBUF][16][STRIDE]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wid = wv & 3, grp = wv >> 2;          // channel slice of the wave, query group of the wave
  float* s_red = lds + NBUF * PR_ROWS * STRIDE + grp * 4 * 256;   // [NQG][4][256]
  const int li = lane & 15, kq = lane >> 4;
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef __fp16 trh4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));   // what ds_read_b64_tr_b16's builtin returns
  typedef __attribute__((address_space(3))) trh4_t* tr_ptr_t;
  typedef float f2_t __attribute__((ext_vector_type(2)));
  typedef Vec8<f16>::type h8_t;
  constexpr float P_SCALE = 16384.0f;
  const float unscale = SPLIT ? 1.0f / (bank_scale * P_SCALE) : 1.0f;       // powers of two: exact

  // slab of rows for this workgroup, in units of slab_unit = 16 RT rows (the boundaries round 1's kernel had)
  const int64_t units_total = (M + slab_unit - 1) / slab_unit;
  const int64_t units_per = (units_total + parts - 1) / parts;
  const int tpu = slab_unit / PR_ROWS;
  const int64_t t_begin = (int64_t)blockIdx.x * units_per * tpu;
  int64_t u_end = ((int64_t)blockIdx.x + 1) * units_per;
  if (u_end > units_total) u_end = units_total;
  const int64_t t_end = u_end * tpu;              // tiles past M hold copies of row M-1, masked in the soft-max (as in k_project)

  float4 qreg[SPLIT ? 1 : DW / 16];             // exact form: the query's channels as fp32 B operands
  h8_t qfh[SPLIT ? DW / 32 : 1], qfl[SPLIT ? DW / 32 : 1];   // split form: (q 2^14) = qh + ql, 8 consecutive channels per fragment
  if constexpr (!SPLIT) {
#pragma unroll
    for (int c = 0; c < DW / 16; ++c) {
      qreg[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q0 + 16 * grp + li < N)
        qreg[c] = *(const float4*)(q + (size_t)(q0 + 16 * grp + li) * D + wid * DW + 16 * c + 4 * kq);
    }
  } else {
    static_assert(!SPLIT || DW % 32 == 0, "split form: 32 channels per MFMA");
#pragma unroll
    for (int c = 0; c < DW / 32; ++c) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (q0 + 16 * grp + li < N) {
        const float* qp = q + (size_t)(q0 + 16 * grp + li) * D + wid * DW + 32 * c + 8 * kq;
        const float4 v0 = *(const float4*)qp, v1 = *(const float4*)(qp + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = v[e] * P_SCALE;            // |q| <= 1 (normalised rows)
        const _Float16 hh = (_Float16)x;
        qfh[c][e] = hh;
        qfl[c][e] = (_Float16)(x - (float)hh);
      }
    }
  }

  // tile t -> buffer b: operation o covers 1 KiB of bank row o / OPR; wave wv issues o = wv, wv + NW, ...
  // Issued through inline asm: a builtin LDS-DMA makes hipcc put s_waitcnt vmcnt(0) in front of the next ds_read of ANY part
  // of this LDS array (it cannot tell the two tile buffers apart) -- the load of tile t+2 would be waited for right after its
  // issue.  Hidden like this, hipcc's own vmcnt arithmetic for ordinary loads can only over-wait, never under-wait.
  const uint32_t lds_tile0 = (uint32_t)(uintptr_t)(pr_lds_ptr_t)s_tile;
#define PIO_DMA_ASM_S(voff, sbase, ldsaddr)      /* address = 64-bit scalar base + per-lane 32-bit offset */     \
  do {                                                                                                         \
    uint32_t _keep;                                                                                            \
    if (PIO_PROJECT_NT)                                                                                        \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" \
                   : "=&s"(_keep) : "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory");                           \
    else                                                                                                       \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                   : "=&s"(_keep) : "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory");                           \
  } while (0)
#define PIO_DMA_ASM_V(vaddr, ldsaddr)            /* address = per-lane 64-bit pointer */                         \
  do {                                                                                                         \
    uint32_t _keep;                                                                                            \
    if (PIO_PROJECT_NT)                                                                                        \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" \
                   : "=&s"(_keep) : "v"(vaddr), "s"(ldsaddr) : "memory");                                      \
    else                                                                                                       \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                   : "=&s"(_keep) : "v"(vaddr), "s"(ldsaddr) : "memory");                                      \
  } while (0)
  // per-lane byte offsets of this wave's operations inside a tile (constant), for the scalar-base form of the load
  uint32_t dma_off[OPS / NW];
#pragma unroll
  for (int i = 0; i < OPS / NW; ++i) {
    const int o = wv + NW * i, row = o / OPR, part = o % OPR;
    dma_off[i] = (uint32_t)((row * D + part * 256 + lane * 4) * 4);
  }
#define PIO_DMA_TILE(t, b)                                                                                     \
  do {                                                                                                         \
    if (((t) + 1) * PR_ROWS <= M) {            /* every row exists: scalar tile base + the constant lane offsets */ \
      const float* _base = bank + (t) * PR_ROWS * D;                                                           \
      _Pragma("unroll") for (int _i = 0; _i < OPS / NW; ++_i) {                                                \
        const int _o = wv + NW * _i, _row = _o / OPR, _part = _o % OPR;                                        \
        const uint32_t _dst = __builtin_amdgcn_readfirstlane(lds_tile0 + (uint32_t)((((b) * PR_ROWS + _row) * STRIDE + _part * 256) * 4)); \
        if (_part * 64 + lane < D / 4) PIO_DMA_ASM_S(dma_off[_i], _base, _dst);                               \
      }                                                                                                        \
    } else {                                   /* the bank's last tile: rows past M read row M-1 (masked later) */ \
      _Pragma("unroll") for (int _i = 0; _i < OPS / NW; ++_i) {                                                \
        const int _o = wv + NW * _i, _row = _o / OPR, _part = _o % OPR;                                        \
        int64_t _g = (t) * PR_ROWS + _row;                                                                     \
        _g = _g < M ? _g : M - 1;                                                                              \
        const float* _src = bank + _g * D + _part * 256 + lane * 4;                                            \
        const uint32_t _dst = __builtin_amdgcn_readfirstlane(lds_tile0 + (uint32_t)((((b) * PR_ROWS + _row) * STRIDE + _part * 256) * 4)); \
        if (_part * 64 + lane < D / 4) PIO_DMA_ASM_V(_src, _dst);                                              \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)

  f32x4 acc[DW / 16];
#pragma unroll
  for (int j = 0; j < DW / 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;          // for query n = li (replicated over kq and over waves)
  float p[4] = {0.f, 0.f, 0.f, 0.f}, alpha = 1.0f, rs_part = 0.f;
  float inv4[4];

  // partial S of the tile in buffer b over this wave's channels -> s_red
#define PIO_GEMM1(b)                                                                                           \
  do {                                                                                                         \
    const float* sb = s_tile + (b) * PR_ROWS * STRIDE;                                                         \
    f32x4 sp = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                    \
    _Pragma("unroll") for (int c = 0; c < DW / 16; ++c) {                                                      \
      const int d = wid * DW + 16 * c + 4 * kq;                                                                \
      const float4 a = *(const float4*)(sb + li * STRIDE + d);                                                 \
      const float4 bq = qreg[c];                                                                               \
      sp = mfma16(a.x, bq.x, sp);                                                                              \
      sp = mfma16(a.y, bq.y, sp);                                                                              \
      sp = mfma16(a.z, bq.z, sp);                                                                              \
      sp = mfma16(a.w, bq.w, sp);                                                                              \
    }                                                                                                          \
    *(f32x4*)(s_red + wid * 256 + lane * 4) = sp;                                                              \
  } while (0)
  // SPLIT: partial S over this wave's channels from the split tile in buffer b (hi plane at byte 0 of a row, lo plane at 2 D)
#define PIO_GEMM1_SPLIT(b)                                                                                     \
  do {                                                                                                         \
    const char* rb = (const char*)(s_tile + (b) * PR_ROWS * STRIDE) + li * (STRIDE * 4) + (wid * DW + 8 * kq) * 2; \
    f32x4 sp0 = (f32x4){0.f, 0.f, 0.f, 0.f}, sp1 = (f32x4){0.f, 0.f, 0.f, 0.f};                                \
    _Pragma("unroll") for (int c = 0; c < DW / 32; ++c) {                                                      \
      const h8_t ah = *(const h8_t*)(rb + 64 * c), al = *(const h8_t*)(rb + 64 * c + 2 * D);                   \
      sp0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qfh[c], sp0, 0, 0, 0);                                  \
      sp1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qfl[c], sp1, 0, 0, 0);                                  \
      sp1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qfh[c], sp1, 0, 0, 0);                                  \
    }                                                                                                          \
    *(f32x4*)(s_red + wid * 256 + lane * 4) = sp0 + sp1;                                                       \
  } while (0)
  // sum of the four partials, online soft-max for query n = li (this lane's rows are 4 kq + i): p[], alpha, m_run; the
  // row sum stays partial (rs_part) until PIO_ROWSUM
#define PIO_SOFTMAX(t)                                                                                         \
  do {                                                                                                         \
    f32x4 sfull = *(const f32x4*)(s_red + lane * 4);                                                           \
    _Pragma("unroll") for (int w = 1; w < 4; ++w) {                                                            \
      const f32x4 o = *(const f32x4*)(s_red + w * 256 + lane * 4);                                             \
      sfull += o;                                                                                              \
    }                                                                                                          \
    float tmax = -INFINITY;                                                                                    \
    const int rows_left = (int)(M - (t) * PR_ROWS < PR_ROWS ? M - (t) * PR_ROWS : PR_ROWS);   /* 16 but in the bank's last tile */ \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
      float z = (sfull[i] * (SPLIT ? inv4[i] * unscale : inv4[i])) / temperature;   /* a power of two: inv4 * unscale is exact */ \
      if (4 * kq + i >= rows_left) z = -INFINITY;                                                              \
      p[i] = z;                                                                                                \
      tmax = fmaxf(tmax, z);                                                                                   \
    }                                                                                                          \
    tmax = xor16_max(tmax);                                                                                    \
    tmax = xor32_max(tmax);                                                                                    \
    const float m_new = fmaxf(m_run, tmax);                                                                    \
    alpha = expf(m_run - m_new);                                                                               \
    float rs = 0.f;                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
      p[i] = expf(p[i] - m_new);                                                                               \
      rs += p[i];                                                                                              \
    }                                                                                                          \
    rs_part = rs;                                                                                              \
    m_run = m_new;                                                                                             \
  } while (0)
  // Acc[n][d] = Acc * alpha + P^T . bank over the tile in buffer b (this lane's accumulator register i belongs to query 4 kq + i)
#define PIO_GEMM2(b)                                                                                           \
  do {                                                                                                         \
    const float* sb = s_tile + (b) * PR_ROWS * STRIDE;                                                         \
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {                                                     \
      float al[4];                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) al[i] = __shfl(alpha, 4 * kq + i);                         \
      _Pragma("unroll") for (int j = 0; j < DW / 16; ++j)                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[j][i] *= al[i];                                      \
    }                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < DW / 16; ++j) {                                                      \
      f32x4 a4 = acc[j];                                                                                       \
      const float* bp = sb + (4 * kq) * STRIDE + wid * DW + 16 * j + li;                                       \
      _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) a4 = mfma16(p[tt], bp[tt * STRIDE], a4);               \
      acc[j] = a4;                                                                                             \
    }                                                                                                          \
  } while (0)
  // SPLIT: Acc = Acc * alpha + (Ph + Pl)^T . (Bh + Bl) over the split tile in buffer b.  B fragment of 16 channels: lane 4 q + p of
  // a 16-lane group hands ds_read_b64_tr_b16 the address of row 4 kq + q, channels d0 + 4 p .. + 3 and receives channel d0 + li of
  // rows 4 kq .. 4 kq + 3 (EXEC is all ones here: no divergent code around the loop).
#define PIO_GEMM2_SPLIT(b)                                                                                     \
  do {                                                                                                         \
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {                                                     \
      float al[4];                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) al[i] = __shfl(alpha, 4 * kq + i);                         \
      _Pragma("unroll") for (int j = 0; j < DW / 16; ++j)                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[j][i] *= al[i];                                      \
    }                                                                                                          \
    float px[4];                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) px[i] = p[i] * P_SCALE;                                      \
    const h2_t ph01 = __builtin_convertvector((f2_t){px[0], px[1]}, h2_t), ph23 = __builtin_convertvector((f2_t){px[2], px[3]}, h2_t); \
    const h2_t pl01 = __builtin_convertvector((f2_t){px[0] - (float)ph01[0], px[1] - (float)ph01[1]}, h2_t);   \
    const h2_t pl23 = __builtin_convertvector((f2_t){px[2] - (float)ph23[0], px[3] - (float)ph23[1]}, h2_t);   \
    const h8_t ah = {ph01[0], ph01[1], ph23[0], ph23[1], ph01[0], ph01[1], ph23[0], ph23[1]};                  \
    const h8_t al8 = {pl01[0], pl01[1], pl23[0], pl23[1], pl01[0], pl01[1], pl23[0], pl23[1]};                 \
    const char* cb = (const char*)(s_tile + (b) * PR_ROWS * STRIDE) + (4 * kq + (li >> 2)) * (STRIDE * 4) +    \
                     (wid * DW + 4 * (li & 3)) * 2;                                                            \
    _Pragma("unroll") for (int j = 0; j < DW / 16; ++j) {                                                      \
      const trh4_t bh = __builtin_amdgcn_ds_read_tr16_b64_v4f16((tr_ptr_t)(cb + 32 * j));                     \
      const trh4_t bl = __builtin_amdgcn_ds_read_tr16_b64_v4f16((tr_ptr_t)(cb + 32 * j + 2 * D));             \
      const h8_t bf = {(_Float16)bh[0], (_Float16)bh[1], (_Float16)bh[2], (_Float16)bh[3],                     \
                       (_Float16)bl[0], (_Float16)bl[1], (_Float16)bl[2], (_Float16)bl[3]};                    \
      f32x4 a4 = acc[j];                                                                                       \
      a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bf, a4, 0, 0, 0);                                        \
      a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al8, bf, a4, 0, 0, 0);                                       \
      acc[j] = a4;                                                                                             \
    }                                                                                                          \
  } while (0)
#define PIO_ROWSUM()                                                                                           \
  do {                                                                                                         \
    float rs = rs_part;                                                                                        \
    rs = xor16_add(rs);                                                                                        \
    rs = xor32_add(rs);                                                                                        \
    l_run = l_run * alpha + rs;                                                                                \
  } while (0)
#define PIO_LOAD_INV(t)                                                                                        \
  do {                                                                                                         \
    const float* _ib = inv_norm + (t) * PR_ROWS;                /* scalar tile base + a 32-bit lane offset */        \
    const int _last = (int)(M - 1 - (t) * PR_ROWS);                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) inv4[i] = _ib[4 * kq + i < _last ? 4 * kq + i : _last];      \
  } while (0)
#define PIO_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)   /* vmcnt(0), expcnt / lgkmcnt untouched: known to hipcc's own counting */
#define PIO_RAW_BARRIER()                                                                                      \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)

  {
    if (t_begin < t_end) {
      PIO_DMA_TILE(t_begin, 0);
      if (t_begin + 1 < t_end) PIO_DMA_TILE(t_begin + 1, 1);
      PIO_LOAD_INV(t_begin);
      PIO_WAIT_VM0();
      PIO_RAW_BARRIER();
      if constexpr (SPLIT) PIO_GEMM1_SPLIT(0);
      else PIO_GEMM1(0);
      PIO_RAW_BARRIER();
      PIO_SOFTMAX(t_begin);
    }
    int cur = 0;
#ifdef PIO_PROJ_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#endif
    for (int64_t t = t_begin; t < t_end; ++t) {
      const bool more = t + 1 < t_end;
      float inv_next[4];
      if (more) {
        const float* ib = inv_norm + (t + 1) * PR_ROWS;
        const int last = (int)(M - 1 - (t + 1) * PR_ROWS);      // >= 0: tile t+1 starts inside the bank
#pragma unroll
        for (int i = 0; i < 4; ++i) inv_next[i] = ib[4 * kq + i < last ? 4 * kq + i : last];
      }
      if constexpr (SPLIT) PIO_GEMM2_SPLIT(cur);
      else PIO_GEMM2(cur);
      PIO_ROWSUM();                     // l_run of tile t, in the shadow of the MFMAs just issued
      PROJ_STAMP(0);
      PIO_WAIT_VM0();                   // this wave's share of tile t+1 (issued an iteration ago) and inv_next
      PROJ_STAMP(1);
      PIO_RAW_BARRIER();                // A: tile t+1 is complete, nobody reads tile t (or s_red) any more
      PROJ_STAMP(2);
      if (t + 2 < t_end) PIO_DMA_TILE(t + 2, cur);
      if (more) {
#pragma unroll
        for (int i = 0; i < 4; ++i) inv4[i] = inv_next[i];
        if constexpr (SPLIT) PIO_GEMM1_SPLIT(cur ^ 1);
        else PIO_GEMM1(cur ^ 1);
        PROJ_STAMP(3);
        PROJ_STAMP(4);
        PIO_RAW_BARRIER();              // B: the four partials of the query group are in s_red
        PROJ_STAMP(5);
        PIO_SOFTMAX(t + 1);
        PROJ_STAMP(6);
      }
#ifdef PIO_PROJ_STAMPS
      st_acc[7] += 1;
#endif
      cur ^= 1;
    }
#ifdef PIO_PROJ_STAMPS
    if (proj_stamps != nullptr && tid == 0) {
      for (int i = 0; i < 7; ++i) proj_stamps[8 * blockIdx.x + i] = st_acc[i];
      proj_stamps[8 * blockIdx.x + 7] = st_acc[7];
    }
#endif
  }
#undef PIO_DMA_TILE
#undef PIO_DMA_ASM_S
#undef PIO_DMA_ASM_V
#undef PIO_GEMM1
#undef PIO_GEMM1_SPLIT
#undef PIO_SOFTMAX
#undef PIO_GEMM2
#undef PIO_GEMM2_SPLIT
#undef PIO_ROWSUM
#undef PIO_LOAD_INV
#undef PIO_WAIT_VM0
#undef PIO_RAW_BARRIER

  // ---- partial results: part_acc[block][n][D], part_ml[block][n][2] ----
  float* pa = part_acc + ((size_t)blockIdx.x * NQ + 16 * grp) * D;
#pragma unroll
  for (int j = 0; j < DW / 16; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) pa[(size_t)(4 * kq + i) * D + wid * DW + 16 * j + li] = SPLIT ? acc[j][i] * unscale : acc[j][i];
  if (wid == 0 && kq == 0) {
    part_ml[((size_t)blockIdx.x * NQ + 16 * grp + li) * 2 + 0] = m_run;
    part_ml[((size_t)blockIdx.x * NQ + 16 * grp + li) * 2 + 1] = l_run;
  }
}

// Merge the per-workgroup partials: out[n][d] = sum_b e^{m_b - M} acc_b[n][d] / sum_b e^{m_b - M} l_b.
// One workgroup per (query, 64 channels): lane = channel, the 4 waves split the partials.
__global__ __launch_bounds__(256) void k_project_combine(const float* __restrict__ part_acc,
                                                         const float* __restrict__ part_ml, int parts, int D, int q0,
                                                         int qstride, float* out) {
  __shared__ float s_w[1024];
  __shared__ float red[4];
  __shared__ float s_acc[4][64];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float mx = -INFINITY;
  for (int b = tid; b < parts; b += 256) mx = fmaxf(mx, part_ml[((size_t)b * qstride + n) * 2]);
  mx = wave_max(mx);
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float ls = 0.f;
  for (int b = tid; b < parts; b += 256) {
    const float mb = part_ml[((size_t)b * qstride + n) * 2];
    const float w = (mb == -INFINITY) ? 0.f : expf(mb - mx);   // workgroups with an empty slab
    s_w[b] = w;
    ls += w * part_ml[((size_t)b * qstride + n) * 2 + 1];
  }
  ls = wave_sum(ls);
  if (lane == 0) red[wid] = ls;
  __syncthreads();
  const float inv_l = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  const int d = blockIdx.y * 64 + lane;
  float a = 0.f;
#pragma unroll 8
  for (int b = wid; b < parts; b += 4) a += s_w[b] * part_acc[((size_t)b * qstride + n) * D + d];
  s_acc[wid][lane] = a;
  __syncthreads();
  if (wid == 0) out[(size_t)(q0 + n) * D + d] = ((s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane])) * inv_l;
}

// Cosine similarities of 16 queries against every row + per-query top-k (return_n_best_sims path,
// im2txtprojection.py:371-375, 382-383).  One wave per bank row; k <= 16.
__global__ __launch_bounds__(256) void k_project_sims(const float* __restrict__ bank, const float* __restrict__ inv_norm,
                                                      int64_t M, int D, const float* __restrict__ q, int N, float* sims) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* r = bank + row * D;
  for (int n = 0; n < N; ++n) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += r[d] * q[(size_t)n * D + d];
    s = wave_sum(s);
    if (lane == 0) sims[(size_t)n * M + row] = s * inv_norm[row];
  }
}

__global__ __launch_bounds__(256) void k_topk_desc(const float* __restrict__ sims, int64_t M, int k, float* out,
                                                   long long* out_rows = nullptr) {
  // one workgroup per query; k rounds of block-wide arg-max with exclusion (k is tiny)
  __shared__ float s_v[4];
  __shared__ long long s_i[4];
  __shared__ long long taken[16];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* r = sims + (size_t)n * M;
  for (int round = 0; round < k; ++round) {
    float bv = -INFINITY;
    long long bi = -1;
    for (int64_t i = tid; i < M; i += 256) {
      bool used = false;
      for (int u = 0; u < round; ++u) used |= (taken[u] == i);
      const float v = r[i];
      if (!used && (v > bv || bi < 0)) { bv = v; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const long long oi = __shfl_xor(bi, o);
      if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      float fv = s_v[0]; long long fi = s_i[0];
      for (int w = 1; w < 4; ++w)
        if (s_i[w] >= 0 && (fi < 0 || s_v[w] > fv || (s_v[w] == fv && s_i[w] < fi))) { fv = s_v[w]; fi = s_i[w]; }
      taken[round] = fi;
      out[(size_t)n * k + round] = fv;
      if (out_rows != nullptr) out_rows[(size_t)n * k + round] = fi;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_revert(const float* __restrict__ x, const float* __restrict__ b,
                                                const float* __restrict__ A, int D, int P, float* out) {
  // out[n][p] = sum_d (x[n][d] - b[d]) * A_pinv[p][d]; one wave per output element
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y;
  if (p >= P) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += (x[(size_t)n * D + d] - b[d]) * A[(size_t)p * D + d];
  s = wave_sum(s);
  if (lane == 0) out[(size_t)n * P + p] = s;
}

template <int D, int NQG, bool SPLIT>
static hipError_t project_pass(const ProjectArgs& a, int q0, int parts, hipStream_t s) {
  constexpr int RT = NQG >= 2 ? PIO_PROJECT_RT32 : 1;           // slab boundaries in units of 16 RT rows (as in rounds 1 and 2)
  const int smem = (2 * PR_ROWS * (D + (SPLIT ? 8 : 4)) + NQG * 4 * 256) * (int)sizeof(float);
  static bool attr_set[64] = {};                                  // function attributes are per device
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (!attr_set[dev & 63]) {
    e = hipFuncSetAttribute((const void*)k_project2<D, NQG, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set[dev & 63] = true;
  }
  hipLaunchKernelGGL((k_project2<D, NQG, SPLIT>), dim3(parts), dim3(256 * NQG), smem, s, SPLIT ? (const float*)a.bank_split : a.bank,
                     a.inv_norm, a.M, a.q, a.N, q0,
                     a.temperature, a.part_acc, a.part_ml, parts, PR_ROWS * RT, a.bank_scale);
  return hipGetLastError();
}

template <int NQG>
static hipError_t project_pass_d(const ProjectArgs& a, int q0, int parts, hipStream_t s) {
  const bool split = a.bank_scale > 0.f && a.bank_split != nullptr && (NQG >= 2 || PIO_PROJECT_SPLIT16);
  if constexpr (NQG == 3) {      // 48 queries per pass: the tile's 48 LDS-DMA operations divide among 12 waves only at D = 768
    if (a.D != 768) return hipErrorInvalidValue;
    return split ? project_pass<768, 3, true>(a, q0, parts, s) : project_pass<768, 3, false>(a, q0, parts, s);
  } else {
    switch (a.D) {
      case 384: return split ? project_pass<384, NQG, true>(a, q0, parts, s) : project_pass<384, NQG, false>(a, q0, parts, s);
      case 512: return split ? project_pass<512, NQG, true>(a, q0, parts, s) : project_pass<512, NQG, false>(a, q0, parts, s);
      case 768: return split ? project_pass<768, NQG, true>(a, q0, parts, s) : project_pass<768, NQG, false>(a, q0, parts, s);
      default: return hipErrorInvalidValue;
    }
  }
}

hipError_t launch_mem_project(const ProjectArgs& a, hipStream_t s) {
  if (a.N <= 0 || a.M <= 0 || a.parts > 1024 || a.parts < 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_l2norm_rows, dim3(a.N), dim3(256), 0, s, a.q, a.D);
  for (int q0 = 0; q0 < a.N;) {
    // more than 16 queries left: one pass of 32 (512-thread workgroups, one per CU: parts / 2 of them, same partial
    // buffers with 32 queries per workgroup); else a pass of 16
    // round 4: more than 32 left (and D = 768, the split image): one pass of 48 -- 768-thread workgroups, three query groups on
    // one tile stream (12 waves x 164 VGPRs fill a CU); the pass is bandwidth-bound, so queries 33..48 ride almost free
    // (80 queries of a 5-batch ViT launch: 48 + 32 instead of 32 + 32 + 16)
    const int left = a.N - q0;
    // (49 .. 64 left: 32 + 32 beats 48 + 16 -- 0.82 against 0.87 ms; tools/microbench/project_time.py)
    const bool q48 = PIO_PROJECT_Q48 && left > 2 * PR_Q && !(left > 3 * PR_Q && left <= 4 * PR_Q) && a.D == 768 &&
                     a.part_rows >= 3 * PR_Q * (a.parts / 2);
    const int groups = q48 ? 3 : (PIO_PROJECT_Q32 && left > PR_Q ? 2 : 1);
    const int cap = groups * PR_Q, nq = left < cap ? left : cap;
    const int parts = groups >= 2 ? a.parts / 2 : a.parts;
    const hipError_t e = groups == 3 ? project_pass_d<3>(a, q0, parts, s) : (groups == 2 ? project_pass_d<2>(a, q0, parts, s) : project_pass_d<1>(a, q0, parts, s));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_project_combine, dim3(nq, a.D / 64), dim3(256), 0, s, a.part_acc, a.part_ml, parts, a.D,
                       q0, cap, a.out);
    q0 += nq;
  }
  if (a.normalize) hipLaunchKernelGGL(k_l2norm_rows, dim3(a.N), dim3(256), 0, s, a.out, a.D);
  if (a.n_best > 0) {
    if (a.n_best > 16 || a.best_sims == nullptr || a.part_best == nullptr) return hipErrorInvalidValue;
    for (int q0 = 0; q0 < a.N; q0 += PR_Q) {   // part_best is the [16][M] similarity scratch
      const int nq = (a.N - q0) < PR_Q ? (a.N - q0) : PR_Q;
      hipLaunchKernelGGL(k_project_sims, dim3((unsigned)((a.M + 3) / 4)), dim3(256), 0, s, a.bank, a.inv_norm, a.M, a.D,
                         a.q + (size_t)q0 * a.D, nq, a.part_best);
      hipLaunchKernelGGL(k_topk_desc, dim3(nq), dim3(256), 0, s, a.part_best, a.M, a.n_best,
                         a.best_sims + (size_t)q0 * a.n_best);
    }
  }
  return hipGetLastError();
}

hipError_t launch_mem_topk(const float* bank, const float* inv_norm, int64_t M, int D, float* q, int N, int k, float* sims_scratch,
                           float* best_sims, int64_t* best_rows, hipStream_t s) {
  if (N <= 0 || M <= 0 || k < 1 || k > 16 || (int64_t)k > M) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_l2norm_rows, dim3(N), dim3(256), 0, s, q, D);
  for (int q0 = 0; q0 < N; q0 += PR_Q) {
    const int nq = (N - q0) < PR_Q ? (N - q0) : PR_Q;
    hipLaunchKernelGGL(k_project_sims, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, bank, inv_norm, M, D, q + (size_t)q0 * D, nq,
                       sims_scratch);
    hipLaunchKernelGGL(k_topk_desc, dim3(nq), dim3(256), 0, s, sims_scratch, M, k, best_sims + (size_t)q0 * k,
                       (long long*)best_rows + (size_t)q0 * k);
  }
  return hipGetLastError();
}

hipError_t launch_l2norm_rows(float* x, int N, int D, hipStream_t s) {
  hipLaunchKernelGGL(k_l2norm_rows, dim3(N), dim3(256), 0, s, x, D);
  return hipGetLastError();
}

hipError_t launch_row_inv_norm(const float* bank, int64_t M, int D, float* inv_norm, hipStream_t s) {
  hipLaunchKernelGGL(k_row_inv_norm, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, bank, M, D, inv_norm);
  return hipGetLastError();
}

hipError_t launch_split_bank(const float* bank, int64_t M, int D, float scale, void* out, hipStream_t s) {
  hipLaunchKernelGGL(k_split_bank, dim3(4096), dim3(256), 0, s, bank, M, D, scale, (_Float16*)out);
  return hipGetLastError();
}

hipError_t launch_abs_max(const float* x, int64_t n, uint32_t* out, hipStream_t s) {
  hipError_t e = hipMemsetAsync(out, 0, 4, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_abs_max, dim3(2048), dim3(256), 0, s, x, n, out);
  return hipGetLastError();
}

hipError_t launch_revert(const float* x, const float* b, const float* A_pinv, int N, int D, int P, float* out,
                         hipStream_t s) {
  hipLaunchKernelGGL(k_revert, dim3(ceil_div(P, 4), N), dim3(256), 0, s, x, b, A_pinv, D, P, out);
  return hipGetLastError();
}

}  // namespace pio
