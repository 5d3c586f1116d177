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
//   * the GEMMs that end a residual branch (attn.c_proj, mlp.c_proj) add bias + result into the residual
//     stream in place; mlp.c_proj (K = 3072) is split over 4 workgroups per column group whose partial tiles
//     are summed in slice order by the last arriver (ticket counter, agent-scope release/acquire):
//     deterministic, no floating-point atomics.
//   * above 64 prefixes (round 4) the three wide layer GEMMs run on split-fp16 operands (k_dec_gemm_s: fp16 hi / lo pairs, three
//     v_mfma_f32_16x16x32_f16 per product, fp32-class accuracy): at that size the fp32 matrix pipe, 256 flop / clk / CU, is what a
//     workgroup spends its residency on.
//   * the LM head never materialises logits: each workgroup reduces its 16 columns to (max, arg-max,
//     sum-exp) per prefix and k_dec_select merges the 3142 partials into the id / log-prob and writes the
//     next input embedding.
#include "common.h"
#include "kernels.h"
#include <cmath>
#include <cstring>

namespace pio {

// (Round 5: the timing ablations that produced wrong results by design -- PIO_DABL_NOX / NOSPLIT / NOMFMA, PIO_LMF16_ABL, PIO_DEC_DENSE -- and the
// fence-based split-K ticket that the relaxed-atomic one replaced have left this file; their numbers are in DESIGN.md section 5 and
// profiles/r04_*, the code in the git history of round 4.)
// minimum waves per SIMD requested for k_dec_gemm.  (5 => <= 96 VGPRs would let decode waves sit beside two
// 208-register ViT GEMM waves when batches are pipelined on several streams; measured: it spills and is slower
// both alone, 6.6 vs 5.8 ms per 30 steps, and pipelined, 2586 vs 2984 captions/s.)
#ifndef PIO_DEC_GEMM_WAVES
#define PIO_DEC_GEMM_WAVES 2
#endif
#ifndef PIO_DEC_WAVES_RG1    // waves per k_dec_gemm workgroup at <= 16 prefixes (K = 768 GEMMs)
#define PIO_DEC_WAVES_RG1 4
#endif
#ifndef PIO_DEC_WAVES_RG4    // waves per k_dec_gemm workgroup at 33..64 prefixes
#define PIO_DEC_WAVES_RG4 8
#endif
#ifndef PIO_LMF16_NT          // which row-group counts of k_lmhead_f16 stream their weights non-temporally
#define PIO_LMF16_NT(RG) ((RG) >= 4)
#endif
#ifndef PIO_LMF16_DEEP        // 64 / 128 prefixes: three X~ / weight chunks in flight instead of one / two (measured: 56.6 vs 50.3 us, off)
#define PIO_LMF16_DEEP 0
#endif
#ifndef PIO_LMF16_WAVES8      // waves per workgroup of k_lmhead_f16 at 128 prefixes (4 = round 1)
#define PIO_LMF16_WAVES8 8
#endif
#ifndef PIO_LMF16_WAVES4      // ... at 33..64 prefixes
#define PIO_LMF16_WAVES4 4
#endif
#ifndef PIO_LMF16_CPW8        // 16-column groups per wave of k_lmhead_f16 at 65 .. 128 prefixes (2: half the X~ fragment reads per MFMA, 197 instead of
#define PIO_LMF16_CPW8 1      // 393 workgroups; measured 6.82 against 6.75 ms per decode(128) and no difference through the pipeline: 1)
#endif
#ifndef PIO_LMF16_FUSED       // <= 16 prefixes: statistics / fp16 conversion inside the head kernel
#define PIO_LMF16_FUSED 1
#endif
#ifndef PIO_LMHEAD_FILTER     // greedy ids through the fp16 filter + exact re-evaluation (log-probabilities: exact head)
#define PIO_LMHEAD_FILTER 1
#endif
#ifndef PIO_DEC_TILED           // above 16 prefixes: k_dec_gemm_b (X tiles through LDS) instead of k_dec_gemm
#define PIO_DEC_TILED 1
#endif
#ifndef PIO_DEC_XLDS
#define PIO_DEC_XLDS 1
#endif
#ifndef PIO_LMHEAD_WIDE
#define PIO_LMHEAD_WIDE 1
#endif
#ifndef PIO_LMHEAD_CG
#define PIO_LMHEAD_CG 1
#endif

static constexpr int DEC_MAX_COLGROUPS = 128;  // split-K counters / slabs: Nout <= 1024 at <= 128 prefixes, fc2's 24 x 4 tiles at 256 (kernels.h: DEC_SPLITK_*)
static constexpr size_t DEC_SPLITK_WS_FLOATS = (size_t)DEC_SPLITK_COUNTERS * 4 * 8 * 256;   // api.cpp: splitk_ws
static_assert(DEC_MAX_COLGROUPS == DEC_SPLITK_COUNTERS, "kernels.h");

enum DecEpi { DE_STORE = 0, DE_RESID = 1, DE_GELU = 2, DE_EMBED = 3, DE_ARGMAX = 4 };

// diagnostic build only (tools/microbench/dec_bench.hip stamps): shader-clock stamps of wave 0 of every k_dec_gemm workgroup
#ifdef PIO_DEC_STAMPS
__device__ unsigned long long g_dec_stamps[8][512][8];     // [kind = EPI + 4 (KS > 1)][workgroup][stamp]
#define PIO_STAMP(i) do { if (threadIdx.x == 0) g_dec_stamps[(EPI & 3) + (KS > 1 ? 4 : 0)][(blockIdx.y * gridDim.x + blockIdx.x) & 511][i] = __builtin_readcyclecounter(); } while (0)
#else
#define PIO_STAMP(i) do {} while (0)
#endif

typedef _Float16 dec_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 dec_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 dec_h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16f(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// torch.argmax order: NaN counts as the maximum, ties (and several NaNs) go to the lowest index.  A prefix of NaNs
// (the reference's mean over an empty box region) therefore decodes to token 0, as in the reference, instead of
// leaving the arg-max at its sentinel.
__device__ __forceinline__ bool arg_better(float v, int i, float best, int bi) {
  const bool vn = v != v, bn = best != best;
  if (vn != bn) return vn;
  if (vn) return i < bi;
  return v > best || (v == best && i < bi);
}

// max over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in every lane: four v_max_f32_dpp instead of four
// ds_bpermute round trips (__shfl_xor), which serialised the fp16 head's epilogue (32 such reductions per lane).
template <int CTRL> __device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));     // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f32<0x4E>(v));     // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f32<0x141>(v));    // row_half_mirror: lane i <-> 7 - i
  v = fmaxf(v, dpp_f32<0x140>(v));    // row_mirror:      lane i <-> 15 - i
  return v;
}

// sum over the wave without the LDS crossbar: four DPP steps inside each row of 16 lanes, then the rows through v_permlane16/32_swap
// (a __shfl_xor butterfly is six dependent ds_bpermute round trips, ~130 cycles each)
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);     // row_half_mirror
  v += dpp_f32<0x140>(v);     // row_mirror
  return xor32_add(xor16_add(v));
}

// A split-K partial tile crosses XCDs (each has its own L2).  Round 1 published it with an agent-scope RELEASE fence (an L2 write-back)
// and read it behind an ACQUIRE fence (an L2 invalidate); tools/microbench/persist_probe.hip: the same exchange costs half as much when
// the DATA itself moves by relaxed agent-scope atomic accesses (sc1 stores write through, sc1 loads read the coherent copy; 4.2 against
// 8.0 us per all-to-all exchange, every value checked) and no fence is executed at all.  Order: slab stores -> s_waitcnt vmcnt(0) (written
// through) -> workgroup barrier -> ticket; the last arrival's slab loads are issued after it has seen the ticket.
__device__ __forceinline__ void st_agent(float* p, f32x4 v) {
  typedef unsigned long long u64;
  const u64 lo = (u64)__float_as_uint(v[0]) | ((u64)__float_as_uint(v[1]) << 32), hi = (u64)__float_as_uint(v[2]) | ((u64)__float_as_uint(v[3]) << 32);
  __hip_atomic_store((u64*)p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((u64*)p + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 ld_agent(const float* p) {
  typedef unsigned long long u64;
  const u64 lo = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), hi = __hip_atomic_load((const u64*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (f32x4){__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32))};
}

__device__ __forceinline__ float gelu_new(float x) {
  // transformers "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}

// (Round 4, measured and removed: extra workgroups per kernel that warm L2 with the NEXT kernel's weights -- lines do survive the kernel
//  boundary and a same-XCD warm-up takes 1.0 us off a pure streaming consumer, tools/microbench/l2_prefetch_probe.hip -- gave 0.4-0.5 us on
//  c_attn / c_fc, nothing on the others and 4.33 against 4.33 ms per decode: by the stamps below a layer GEMM spends 2.6 of its 5.5 us
//  waiting for X, the previous kernel's output, to cross from the other XCDs' L2 through memory; the weights are not what it waits for.
//  Also measured and removed: producer -> consumer pairs on ONE XCD where the data flow allows it (c_attn -> attention per head,
//  c_fc -> mlp.c_proj per k-slice; ids dealt so that only XCDs 0-3 work): the consumers did not get faster at all (attention 5.21
//  against 5.20 us -- what a kernel leaves in its XCD's L2 does not survive the kernel boundary as a hit for the next one) and the
//  GEMMs that then stream through half the XCDs lose 1.3-1.5 us each: 5.1 against 4.5 ms per decode.
//  And, built and removed in the same round: attn.c_proj folded into the value projection (v'_h = W_p[:, h] W_v[h], one fused matrix
//  per head built at load in fp64; the attention then finishes the residual branch itself and a layer is four kernels instead of
//  five).  Ids identical to the oracle on every decoder test -- and 4.68 against 4.18 ms per decode: the (2 + H) E = 4 608-column
//  c_attn takes 7.9 instead of 5.5 us in the chain and the attention that reads H x the value bytes (8 workgroups per prefix, each
//  needing all four heads' keys) 11.0 instead of 5.2: more than the 5.4 us of the launch it removes.)
// out[n][j] = epilogue( sum_k X[n][k] * W[j][k] )        W [Nout][K] ([out][in]), X [N][K]
//   grid = (ceil(Nout/16), KS) workgroups of NWV waves; each wave owns CPW*16 k's: K = KS * NWV * CPW * 16
//   (KS = 1: K = 768 or 512; KS = 4: K = 3072).  NWV = 4; 8 at more than 32 prefixes, where the X loads (all of
//   X per workgroup, 196 KB at 64 rows) set the time: twice the waves = twice the bytes in flight per CU and half
//   the MFMA chain per wave.
//   lane (li = lane&15, kq = lane>>4): B operand W[col0+li][k0 + 16c + 4kq + t], A operand X[16g+li][same k]
//   (the k order inside a chunk is free as long as A and B agree); C: column li, row 4kq+i.
//   LN != 0: X is the raw residual stream; the LayerNorm is applied algebraically in the epilogue
//            (W is pre-scaled by ln_w; cvec / dvec as in the file header).
//   KS > 1 : in-launch split-K.  Every k-slice workgroup stores its partial tile to `ws`, publishes it with
//            an agent-scope release + one relaxed atomic ticket per column group; the workgroup that draws the
//            last ticket acquires, re-reads ALL KS slabs in slice order (so the sum does not depend on which
//            workgroup was last: deterministic) and runs the epilogue; it also re-arms the counter.
template <int RG, int CPW, int KS, int EPI, int LN, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? PIO_DEC_GEMM_WAVES : 1) void k_dec_gemm(const float* __restrict__ W, const float* __restrict__ X, int N,
                                                  int Nout, int K, const float* __restrict__ bias, float* out,
                                                  const float* __restrict__ extra, const float* __restrict__ cvec,
                                                  float eps, float* ws, unsigned* cnt) {
  __shared__ __attribute__((aligned(16))) float red[NWV * RG * 256];
  __shared__ float s_sum[LN ? NWV : 1][RG * 16], s_sq[LN ? NWV : 1][RG * 16];
  __shared__ int s_last;
  PIO_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int col0 = blockIdx.x * 16;
  const int j = col0 + li;
  const int jc = j < Nout ? j : Nout - 1;
  const int k0 = (blockIdx.y * NWV + wid) * (16 * CPW) + 4 * kq;
  const float* wp = W + (size_t)jc * K + k0;
  float4 w4[CPW];
  f32x4 acc[RG];
  constexpr bool XLDS = PIO_DEC_XLDS && RG == 1 && NWV * CPW == 48;             // the workgroup's K range: 768 columns of X
  const int kbase = blockIdx.y * (NWV * 16 * CPW);
  __shared__ __attribute__((aligned(16))) float s_x[XLDS ? 16 * 772 : 4];
  // Loads return in the order they were issued: the epilogue vectors first (three dwords, needed last), then X, whose staging through
  // LDS then runs under the weight stream instead of behind it, then the weights.  Stamps of a layer GEMM at <= 16 prefixes
  // (tools/microbench/dec_bench.hip built with -DPIO_DEC_STAMPS, shader cycles of wave 0, ~1.9 GHz): entry -> X landed and staged 5.4 k
  // (2.7 us: the previous kernel's output has to cross from the other XCDs' L2 through memory; the same for the 2.4-MB attn.c_proj
  // and the 9.4-MB c_fc, so it is latency, not the weight stream), fragment reads + 48 MFMAs 1.9 k, reduction 0.4-1.1 k, epilogue
  // until the stores have landed 1.7-2.5 k; the split-K ticket of mlp.c_proj another 2.5 k.
  const float bj = EPI == DE_ARGMAX && !LN ? 0.f : bias[jc];
  const float cj = LN ? cvec[jc] : 0.f;
  const float ej = EPI == DE_EMBED ? extra[jc] : 0.f;
  if constexpr (XLDS) {
    // <= 16 prefixes: the workgroup's slice of X (16 x 768) goes through LDS with fully coalesced loads instead of
    // 16-row x 64-B fragment-shaped ones (rows padded to 772 floats: the 16 rows of a fragment read fall on 16
    // distinct 16-B slots).  Same values, same MFMA order: bit-identical; 0.25-0.5 us per kernel.  (Above 16 prefixes,
    // one row group at a time with the next one prefetched: two barriers per group and a staging array that hipcc keeps
    // in scratch made it 2x slower -- 19.9 vs 9.9 us for qkv at 64 prefixes; not kept.)
    constexpr int NT = 64 * NWV, XI = 3072 / NT;
    float4 xs[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + NT * i, r = e / 192, c4 = e - r * 192;
      const int rc = r < N ? r : N - 1;
      xs[i] = *(const float4*)(X + (size_t)rc * K + kbase + 4 * c4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CPW; ++c) w4[c] = *(const float4*)(wp + 16 * c);       // the HBM stream: all in flight
    __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the loads next to their MFMAs (2 in flight)
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + NT * i, r = e / 192, c4 = e - r * 192;
      *(float4*)(s_x + r * 772 + 4 * c4) = xs[i];
    }
    PIO_STAMP(1);
    __syncthreads();
    PIO_STAMP(2);
  } else {
#pragma unroll
    for (int c = 0; c < CPW; ++c) w4[c] = *(const float4*)(wp + 16 * c);       // the HBM stream: all in flight
    __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the loads next to their MFMAs (2 in flight)
  }
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    int n = g * 16 + li;
    n = n < N ? n : N - 1;
    const float* xp = X + (size_t)n * K + k0;
    // activations (L2-resident), requested in two halves right behind the weight stream
    constexpr int HC = CPW / 2;
    float4 xa[HC], xb[HC];
    if constexpr (XLDS) {
#pragma unroll
      for (int c = 0; c < HC; ++c) xa[c] = *(const float4*)(s_x + li * 772 + (k0 - kbase) + 16 * c);
#pragma unroll
      for (int c = 0; c < HC; ++c) xb[c] = *(const float4*)(s_x + li * 772 + (k0 - kbase) + 16 * (HC + c));
    } else {
#pragma unroll
    for (int c = 0; c < HC; ++c) xa[c] = *(const float4*)(xp + 16 * c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < HC; ++c) xb[c] = *(const float4*)(xp + 16 * (HC + c));
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = (f32x4){0.f, 0.f, 0.f, 0.f};   // two independent MFMA chains
    float sx = 0.f, sq = 0.f;
    // chunk pairs (cc, cc+1): even chunks feed chain a0, odd ones a1; chunk cc lives in xa (cc < HC) or xb
#define PIO_XC(cc) ((cc) < HC ? xa[(cc) < HC ? (cc) : 0] : xb[(cc) < HC ? 0 : (cc) - HC])
#pragma unroll
    for (int cc = 0; cc < CPW; cc += 2) {
      const float4 x0 = PIO_XC(cc), x1 = PIO_XC(cc + 1);
      a0 = mfma16f(x0.x, w4[cc].x, a0);  a1 = mfma16f(x1.x, w4[cc + 1].x, a1);
      a0 = mfma16f(x0.y, w4[cc].y, a0);  a1 = mfma16f(x1.y, w4[cc + 1].y, a1);
      a0 = mfma16f(x0.z, w4[cc].z, a0);  a1 = mfma16f(x1.z, w4[cc + 1].z, a1);
      a0 = mfma16f(x0.w, w4[cc].w, a0);  a1 = mfma16f(x1.w, w4[cc + 1].w, a1);
      if (LN) {
        sx += ((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w));
        sq += ((x0.x * x0.x + x0.y * x0.y) + (x0.z * x0.z + x0.w * x0.w)) +
              ((x1.x * x1.x + x1.y * x1.y) + (x1.z * x1.z + x1.w * x1.w));
      }
    }
#undef PIO_XC
    acc[g] = a0 + a1;
#ifdef PIO_DEC_STAMPS
    asm volatile("s_nop 0" : "+v"(acc[g]));
    PIO_STAMP(3);
#endif
    if (LN) {   // row statistics of x: this lane holds 4*CPW values of row 16g+li; sum the 4 kq groups
      sx = xor32_add(xor16_add(sx));            // the other three k-quarters of the row: lanes ^ 16, ^ 32 (v_permlane swaps: same bits as
      sq = xor32_add(xor16_add(sq));            // the __shfl_xor pair, without two ds_bpermute round trips on the kernel's critical path)
      if (kq == 0) { s_sum[wid][g * 16 + li] = sx; s_sq[wid][g * 16 + li] = sq; }
    }
  }
#pragma unroll
  for (int g = 0; g < RG; ++g) *(f32x4*)(red + ((wid * RG + g) * 64 + lane) * 4) = acc[g];
  __syncthreads();
  PIO_STAMP(4);
  if constexpr (KS == 1) PIO_STAMP(5);
  f32x4 sums[(RG + NWV - 1) / NWV];
#pragma unroll
  for (int gi = 0; gi < (RG + NWV - 1) / NWV; ++gi) {
    const int g = wid + NWV * gi;
    if (g < RG) {
      f32x4 s = *(const f32x4*)(red + ((0 * RG + g) * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < NWV; ++w) s += *(const f32x4*)(red + ((w * RG + g) * 64 + lane) * 4);
      sums[gi] = s;
    }
  }
  if constexpr (KS > 1) {
    float* slab = ws + ((size_t)blockIdx.x * KS) * RG * 256;
#pragma unroll
    for (int gi = 0; gi < (RG + NWV - 1) / NWV; ++gi) {
      const int g = wid + NWV * gi;
      if (g < RG) {
        float* sp = slab + ((size_t)blockIdx.y * RG + g) * 256 + lane * 4;
        st_agent(sp, sums[gi]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its stores
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(cnt + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned)(KS - 1));
    }
    __syncthreads();
    PIO_STAMP(5);
    if (!s_last) return;
    if (tid == 0) __hip_atomic_store(cnt + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
#pragma unroll
    for (int gi = 0; gi < (RG + NWV - 1) / NWV; ++gi) {
      const int g = wid + NWV * gi;
      if (g < RG) {
        const float* sp = slab + (size_t)g * 256 + lane * 4;
        f32x4 s = ld_agent(sp);
#pragma unroll
        for (int y = 1; y < KS; ++y) s += ld_agent(sp + (size_t)y * RG * 256);
        sums[gi] = s;
      }
    }
  }
#pragma unroll
  for (int gi = 0; gi < (RG + NWV - 1) / NWV; ++gi) {
    const int g = wid + NWV * gi;
    if (g >= RG) continue;
    f32x4 s = sums[gi];
    if (LN) {   // s_i <- r_n (s_i - mu_n c_j) + d_j  for row n = 16g + 4kq + i
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = g * 16 + 4 * kq + i;
        float tx = (s_sum[0][rr] + s_sum[1][rr]) + (s_sum[2][rr] + s_sum[3][rr]);
        float tq = (s_sq[0][rr] + s_sq[1][rr]) + (s_sq[2][rr] + s_sq[3][rr]);
        if constexpr (NWV == 8) {
          tx += (s_sum[4][rr] + s_sum[5][rr]) + (s_sum[6][rr] + s_sum[7][rr]);
          tq += (s_sq[4][rr] + s_sq[5][rr]) + (s_sq[6][rr] + s_sq[7][rr]);
        }
        const float mu = tx / (float)K;
        const float var = fmaxf(tq / (float)K - mu * mu, 0.f);
        s[i] = rsqrtf(var + eps) * (s[i] - mu * cj) + bj;
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
          if (arg_better(ov, oi, mx, idx)) { mx = ov; idx = oi; }
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
        const float v = LN ? s[i] : s[i] + bj;          // LN: the bias is already inside d_j
        float* o = out + (size_t)n * Nout + j;
        if constexpr (EPI == DE_STORE) *o = v;
        else if constexpr (EPI == DE_RESID) *o += v;    // each element has exactly one owner: in place is safe
        else if constexpr (EPI == DE_GELU) *o = gelu_new(v);
        else if constexpr (EPI == DE_EMBED) *o = v + ej;
      }
    }
  }
#ifdef PIO_DEC_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PIO_STAMP(6);
#endif
}

// Causal attention for the new position `pos` of prefix n, head h: appends k,v to the cache and attends
// over positions 0..pos (pos < 128: 30 for DeCap, prompt + 64 for the ViECap head).  One workgroup per (n, head).
//   scores : thread (j = tid>>3, seg = tid&7) takes hd/8 channels of key j -> all key loads of a 32-key
//            pass are in flight at once; 3 shuffles finish the dot product.
//   output : thread (c4 = tid % (hd/4), jg = tid / (hd/4)) accumulates 4 channels over keys j = jg (mod NJ);
//            the NJ partial sums meet in LDS.
// grid.y > 1 (batched prompt prefill, dec_prefill_layers): workgroup (x, y) handles position pos0 + y of prefix n, whose activations are
// row n * rows_per_n + y of qkv / att; the keys and values of ALL those positions have been appended by k_kv_append before the launch
// (the workgroup's own append below then rewrites the values that are there).  Same loads, same order per (prefix, head, position).
__global__ __launch_bounds__(256) void k_dec_attention(const float* __restrict__ qkv, float* kcache, float* vcache,
                                                       int E, int heads, int pos0, int max_steps, float* att, int rows_per_n) {
  __shared__ float s_sc[256];
  __shared__ __attribute__((aligned(16))) float s_o[5][256];
  const int n = blockIdx.x / heads, h = blockIdx.x - n * heads;
  const int pos = pos0 + (int)blockIdx.y;
  const size_t row = (size_t)n * rows_per_n + blockIdx.y;
  const int tid = threadIdx.x;
  const int hd = E / heads;                      // 192; multiple of 32
  const float* q = qkv + row * 3 * E + h * hd;
  const float* kn = q + E;
  const float* vn = q + 2 * E;
  float* kc = kcache + ((size_t)n * max_steps) * E + h * hd;
  float* vc = vcache + ((size_t)n * max_steps) * E + h * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  // The value rows this thread will weight do not depend on the scores: request them first, so that their latency
  // runs beside the score pass instead of behind the soft-max (up to VPRE rows per thread; any further ones are
  // loaded in the output loop).
  const int nc4 = hd >> 2;                          // 48 float4 per head row
  const int NJ = (256 / nc4) < 5 ? (256 / nc4) : 5;   // 5 key groups (240 threads active)
  const int c4 = tid % nc4, jg = tid / nc4;
  constexpr int VPRE = 7;
  float4 vpre[VPRE];
#pragma unroll
  for (int i = 0; i < VPRE; ++i) {
    const int j = jg + NJ * i;
    vpre[i] = (jg < NJ && j <= pos) ? ((const float4*)(j == pos ? vn : vc + (size_t)j * E))[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
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
      s += dpp_f32<0xB1>(s); s += dpp_f32<0x4E>(s); s += dpp_f32<0x141>(s);      // lanes ^ 1, ^ 2, ^ 4 (DPP: same bits as the __shfl_xor form)
      if (seg == 0 && j <= pos) s_sc[j] = s * scale;
    }
  }
  if (tid < hd) {                                  // append the new key / value
    kc[(size_t)pos * E + tid] = kn[tid];
    vc[(size_t)pos * E + tid] = vn[tid];
  }
  __syncthreads();
  if (tid < 64) {                                   // soft-max over the <= 256 scores once, by one wave (four per lane)
    const float sc = tid <= pos ? s_sc[tid] : -INFINITY;
    const float sc2 = tid + 64 <= pos ? s_sc[tid + 64] : -INFINITY;
    const float sc3 = tid + 128 <= pos ? s_sc[tid + 128] : -INFINITY;
    const float sc4 = tid + 192 <= pos ? s_sc[tid + 192] : -INFINITY;
    const float mx = wave_max(fmaxf(fmaxf(sc, sc2), fmaxf(sc3, sc4)));
    const float e = tid <= pos ? expf(sc - mx) : 0.f;
    const float e2 = tid + 64 <= pos ? expf(sc2 - mx) : 0.f;    // a lane's absent terms are +0: e + 0 is e, so the sums of the
    const float e3 = tid + 128 <= pos ? expf(sc3 - mx) : 0.f;   // 30-step DeCap decode (pos < 64) and of the 128-position
    const float e4 = tid + 192 <= pos ? expf(sc4 - mx) : 0.f;   // ViECap search are the bits they were with two per lane
    const float inv = 1.0f / wave_sum(((e + e2) + e3) + e4);
    s_sc[tid] = e * inv;
    s_sc[tid + 64] = e2 * inv;
    s_sc[tid + 128] = e3 * inv;
    s_sc[tid + 192] = e4 * inv;
  }
  __syncthreads();
  if (jg < NJ) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < VPRE; ++i) {
      const int j = jg + NJ * i;
      if (j <= pos) {
        const float p = s_sc[j];
        o.x += p * vpre[i].x; o.y += p * vpre[i].y; o.z += p * vpre[i].z; o.w += p * vpre[i].w;
      }
    }
    for (int j = jg + NJ * VPRE; j <= pos; j += NJ) {
      const float p = s_sc[j];
      const float4 v = ((const float4*)(j == pos ? vn : vc + (size_t)j * E))[c4];
      o.x += p * v.x; o.y += p * v.y; o.z += p * v.z; o.w += p * v.w;
    }
    *(float4*)&s_o[jg][4 * c4] = o;
  }
  __syncthreads();
  if (tid < hd) {
    float o = 0.f;
    for (int g = 0; g < NJ; ++g) o += s_o[g][tid];
    att[row * E + h * hd + tid] = o;
  }
}

// Merge the LM head's per-workgroup (max, arg-max, sum-exp) partials: greedy id (first index on ties, like
// torch.argmax), log-softmax of the chosen logit; then the next step's input x = wte[id] + wpe[step+1].
// One workgroup per prefix.
__global__ __launch_bounds__(256) void k_dec_select(const float* __restrict__ part, int nblk, int N, int E, int step,
                                                    int steps, const float* __restrict__ wte,
                                                    const float* __restrict__ wpe, int32_t* ids, float* logprob,
                                                    float* x, int pos_base) {
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
    if (arg_better(v, i, bv, bi)) { bv = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_v[wid] = bv; s_i[wid] = bi; }
  __syncthreads();
  bv = s_v[0]; bi = s_i[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (arg_better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
  for (int d = tid; d < E; d += 256) x[(size_t)n * E + d] = wte[(size_t)bi * E + d] + wpe[(size_t)(pos_base + step + 1) * E + d];
  if (tid == 0) ids[(size_t)n * steps + step] = bi;
  if (logprob == nullptr) return;                   // block-uniform: the log-probability pass is only run on request
  {
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) se += pr[k].z * expf(pr[k].x - bv);   // padded entries: 0 * exp(-inf) = 0
    se = wave_sum(se);
    if (lane == 0) s_s[wid] = se;
  }
  __syncthreads();
  if (tid == 0) logprob[(size_t)n * steps + step] = -logf((s_s[0] + s_s[1]) + (s_s[2] + s_s[3]));
}

template <int CPW, int KS, int EPI, int LN>
static hipError_t dec_gemm_rg(const float* W, const float* X, int N, int Nout, int K, const float* bias, float* out,
                              const float* extra, const float* cvec, float eps, float* ws, unsigned* cnt,
                              hipStream_t s) {
  const int rg = ceil_div(N, 16);
  const dim3 grid(ceil_div(Nout, 16), KS);
#define PIO_DG(R, C, NW) hipLaunchKernelGGL((k_dec_gemm<R, C, KS, EPI, LN, NW>), grid, dim3(64 * NW), 0, s, W, X, N, Nout, K, bias, out, extra, cvec, eps, ws, cnt)
  if (rg <= 1) {
    // <= 16 prefixes, K = 768: eight waves of 96 k each (round 4, -DPIO_DEC_WAVES_RG1=8) halve the MFMA chain (1.9 k -> 1.3 k cycles by the
    // stamps) and lose it again in the reduction over eight partial tiles: 4.33 against 4.20-4.33 ms per decode, not the default
    if constexpr (PIO_DEC_WAVES_RG1 == 8 && CPW == 12 && KS == 1 && EPI != DE_ARGMAX) PIO_DG(1, CPW / 2, 8);
    else PIO_DG(1, CPW, 4);
  }
  else if (rg <= 2) PIO_DG(2, CPW, 4);
  else if (rg <= 4) {
    if constexpr (PIO_DEC_WAVES_RG4 == 8 && CPW % 4 == 0 && EPI != DE_ARGMAX) PIO_DG(4, CPW / 2, 8);
    else PIO_DG(4, CPW, 4);
  } else if (rg <= 8) {
    if constexpr (EPI != DE_ARGMAX) PIO_DG(8, CPW, 4);     // 65..128 prefixes (ids-only decode): 4 waves, 32 KiB of LDS partials
    else return hipErrorInvalidValue;
  } else return hipErrorInvalidValue;
#undef PIO_DG
  return hipGetLastError();
}


typedef __attribute__((address_space(3))) void* dec_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* dec_gbl_ptr_t;

// ---- decoder GEMMs above 16 prefixes: X tiles through LDS -----------------------------------------------------
// k_dec_gemm re-reads ALL rows of X per 16 output columns in fragment-shaped loads: 442 KB per workgroup at 128
// prefixes, and a CU ingests only ~50 KB/us through L2: every layer GEMM took 16-21 us whatever its grid (48 .. 192
// workgroups), against ~3 us of fp32 MFMA work chip-wide.  Here a workgroup owns [16 RGB rows][16 NCG columns] and the
// whole K = 768 (KS > 1: one 768-slice of K = 3072, met through the same in-launch ticket as k_dec_gemm); wave
// (cg, kw) multiplies column group cg with the kw-th 16 k's of every 64-k chunk.  X chunks ([16 RGB][64] fp32, 16-B
// slots XOR-swizzled with the row as in k_lmhead_wide) stream through an LDS ring by LDS-DMA, shared by all 4 NCG
// waves; W goes straight to registers (12 x 16 B per lane).  Both are inline asm with hand-counted vmcnt (see
// k_lmhead_wide for why); chunk q is consumed while the later chunks are in flight.  Tile shapes are picked per GEMM
// and prefix count from tools/microbench/dec_bench.hip (128 prefixes: 32 x 48 for qkv / fc, 32 x 16 for proj,
// 64 x 32 x 4 k-slices for fc2: 11.6 / 6.1 / 12.2 / 15.0 us against 16.5 / 16.3 / 17.0 / 21.1; 64 prefixes:
// 8.3 / 4.8 / 8.8 / 11.1 against 9.9 / 9.6 / 10.1 / 13.3).
// Measured and dropped: splitting K = 768 as well (bricks of 64 x 64 x 128..384 with the ticket) -- the ticket's
// serial tail (store -> write-back -> cross-XCD atomic -> invalidate -> slab re-read) costs 5-14 us on this part,
// as much as the smaller bricks save.
//   grid = (Nout / (16 NCG), KS, ceil(N / (16 RGB))), 256 NCG threads.
// vm queue of a k_dec_gemm_b wave and the vmcnt that retires everything chunk q needs at the wait of step q.
//   order: W(0) X(0) W(1) X(1) .. W(RB-1) X(RB-1) W(RB) .. W(11) | X(RB) .. X(11), the late X(s+RB-1) issued in step
//   s >= 1 after that step's wait; X(c) is `xops` LDS-DMA operations (0 for a wave that issues none).
__host__ __device__ constexpr int dec_b_allowed(int q, int RB, int xops) {
  int total = 0, need = 0;
  for (int c = 0; c < RB; ++c) {
    total += 1; if (c <= q) need = total;
    total += xops; if (c <= q) need = total;
  }
  for (int c = RB; c < 12; ++c) { total += 1; if (c <= q) need = total; }
  for (int st = 1; st < q; ++st) {
    const int c = st + RB - 1;
    if (c < 12) { total += xops; if (c <= q) need = total; }
  }
  return total - need;
}
template <int C> __device__ __forceinline__ void dec_wait_vm(f32x4& w) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w) : "n"(C) : "memory");
}
template <int RGB, int NCG, int KS, int EPI, int LN>
__global__ __launch_bounds__(256 * NCG, 1) void k_dec_gemm_b(const float* __restrict__ W, const float* __restrict__ X, int N, int Nout, int K,
                                                             const float* __restrict__ bias, float* out,
                                                             const float* __restrict__ cvec, float eps, float* ws, unsigned* cnt) {
  static_assert(!(LN && KS > 1), "LayerNorm row sums need the whole row in one workgroup");
  static_assert(RGB == 1 || RGB == 2 || RGB == 4, "row groups per workgroup");
  constexpr int NW = 4 * NCG, CH = 64, NQ = 12, ROWS = 16 * RGB, XB = ROWS * CH;
  constexpr int RB = RGB == 4 ? 8 : 12;                     // ring depth (RB == NQ: the whole K slice of X is resident)
  constexpr bool RESIDENT = RB == NQ;
  constexpr int PIECES = 4 * RGB;                           // 1-KiB LDS-DMA pieces per chunk
  // RESIDENT: the NQ * PIECES pieces of the slice are dealt round-robin to the waves (piece p = wid + NW i: chunk
  // p / PIECES, rows 4 (p % PIECES) ..), so EVERY wave issues the same number of operations and one set of hand-
  // counted waits serves all of them.  Ring: NW divides PIECES, wave w takes pieces w, w + NW, .. of every chunk.
  constexpr int TP = RESIDENT ? NQ * PIECES / NW : PIECES / NW;
  static_assert(TP * NW == (RESIDENT ? NQ * PIECES : PIECES), "pieces must divide among the waves");
  extern __shared__ __attribute__((aligned(16))) float lsm[];      // [RB][ROWS][64]; afterwards partial tiles [NW][RGB][256]
  __shared__ float s_sum[LN ? 4 : 1][ROWS], s_sq[LN ? 4 : 1][ROWS];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int cg = wid >> 2, kw = wid & 3;
  const int row0 = blockIdx.z * ROWS;
  const int kbase = blockIdx.y * (NQ * CH);
  const int j = (blockIdx.x * NCG + cg) * 16 + li;
  const float* wp = W + (size_t)j * K + kbase + 16 * kw + 4 * kq;
  f32x4 w[NQ];
  // A piece covers rows 4t .. 4t+3 of a chunk (1 KiB); lane l fills slot (l & 15) of row 4t + (l >> 4) and therefore
  // fetches source slot (l & 15) ^ (row & 15).
  const uint32_t lds_base = (uint32_t)(uintptr_t)(dec_lds_ptr_t)lsm;
#define PIO_DMA(gptr, ldsaddr)                                                                                 \
  do {                                                                                                         \
    uint32_t _keep;                                                                                            \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(_keep) : "v"(gptr), "s"(ldsaddr) : "memory");                                         \
  } while (0)
#define PIO_WISSUE(q)                                                                                          \
  do {                                                                                                         \
    const float* _p = wp + (q) * CH;                                                                           \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[q]) : "v"(_p) : "memory");                         \
  } while (0)
  uint32_t xoff[RESIDENT ? 1 : TP];
  if constexpr (RESIDENT) {
    // vm queue: X pieces (L2-hot, TP per wave) | W(0) .. W(11) (the MALL / HBM stream).  Step 0 waits for this wave's
    // pieces and W(0), the barrier makes that "all of X"; step q > 0 only waits for W(q): at most 11 - q outstanding.
#pragma unroll
    for (int i = 0; i < TP; ++i) {
      const int p = wid + NW * i, ch = p / PIECES, t = p % PIECES;
      const int row = 4 * t + (lane >> 4);
      const int rc = row0 + row < N ? row0 + row : N - 1;
      const float* g = X + (size_t)rc * K + kbase + ch * CH + 4 * ((lane & 15) ^ (row & 15));
      const uint32_t l = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((ch * XB + t * 256) * 4));
      PIO_DMA(g, l);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) PIO_WISSUE(q);
  } else {
    // vm queue: W(0) X(0) W(1) X(1) .. W(RB-1) X(RB-1) W(RB) .. W(11) | X(RB) .. X(11), the late X(s+RB-1) issued in
    // step s >= 1; the waits are counted by dec_b_allowed.
#pragma unroll
    for (int i = 0; i < TP; ++i) {
      const int row = 4 * (wid + NW * i) + (lane >> 4);
      const int rc = row0 + row < N ? row0 + row : N - 1;
      xoff[i] = (uint32_t)rc * K + kbase + 4 * ((lane & 15) ^ (row & 15));
    }
  }
#define PIO_XISSUE(q)                                                                                          \
  do {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < TP; ++i) {                                                           \
      const float* _g = X + (q) * CH + xoff[i];                                                                \
      const uint32_t _l = lds_base + (uint32_t)((((q) % RB) * XB + (wid + NW * i) * 256) * 4);                 \
      PIO_DMA(_g, _l);                                                                                         \
    }                                                                                                          \
  } while (0)
  if constexpr (!RESIDENT) {
#pragma unroll
    for (int q = 0; q < RB; ++q) { PIO_WISSUE(q); PIO_XISSUE(q); }
#pragma unroll
    for (int q = RB; q < NQ; ++q) PIO_WISSUE(q);
  }
  f32x4 acc[RGB];
  float sx[RGB], sq[RGB];
#pragma unroll
  for (int g = 0; g < RGB; ++g) { acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; sx[g] = 0.f; sq[g] = 0.f; }
#define PIO_BSTEP(q)                                                                                           \
  do {                                                                                                         \
    if constexpr (RESIDENT) {                                                                                  \
      dec_wait_vm<NQ - 1 - (q)>(w[q]);                                                                         \
      if ((q) == 0) __builtin_amdgcn_s_barrier();      /* every wave's pieces have landed: all of X */          \
    } else {                                                                                                   \
      dec_wait_vm<dec_b_allowed(q, RB, TP)>(w[q]);                                                             \
      __builtin_amdgcn_s_barrier();    /* chunk q landed in every wave; every wave is done with chunk q-1 */    \
      if ((q) >= 1 && (q) + RB - 1 < NQ) PIO_XISSUE((q) + RB - 1);                                             \
    }                                                                                                          \
    const float* _xb = lsm + ((q) % RB) * XB;                                                                  \
    _Pragma("unroll") for (int g = 0; g < RGB; ++g) {                                                          \
      const float4 xf = *(const float4*)(_xb + (16 * g + li) * CH + (((4 * kw + kq) ^ li) << 2));              \
      acc[g] = mfma16f(xf.x, w[q][0], acc[g]);                                                                 \
      acc[g] = mfma16f(xf.y, w[q][1], acc[g]);                                                                 \
      acc[g] = mfma16f(xf.z, w[q][2], acc[g]);                                                                 \
      acc[g] = mfma16f(xf.w, w[q][3], acc[g]);                                                                 \
      if (LN) {          /* every wave keeps the row sums (a few VALU ops, no branch); cg == 0 publishes them */ \
        sx[g] += (xf.x + xf.y) + (xf.z + xf.w);                                                                \
        sq[g] += (xf.x * xf.x + xf.y * xf.y) + (xf.z * xf.z + xf.w * xf.w);                                    \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
  PIO_BSTEP(0); PIO_BSTEP(1); PIO_BSTEP(2); PIO_BSTEP(3); PIO_BSTEP(4); PIO_BSTEP(5);
  PIO_BSTEP(6); PIO_BSTEP(7); PIO_BSTEP(8); PIO_BSTEP(9); PIO_BSTEP(10); PIO_BSTEP(11);
#undef PIO_BSTEP
#undef PIO_XISSUE
#undef PIO_WISSUE
#undef PIO_DMA
  const float bj = bias[j];
  const float cj = LN ? cvec[j] : 0.f;
  if (LN && cg == 0) {
#pragma unroll
    for (int g = 0; g < RGB; ++g) {
      float tx = sx[g], tq = sq[g];
      tx = xor32_add(xor16_add(tx));
      tq = xor32_add(xor16_add(tq));
      if (kq == 0) { s_sum[kw][g * 16 + li] = tx; s_sq[kw][g * 16 + li] = tq; }
    }
  }
  __syncthreads();                                  // every wave is done with the ring: reuse it for the partial tiles
#pragma unroll
  for (int g = 0; g < RGB; ++g) *(f32x4*)(lsm + ((wid * RGB + g) * 64 + lane) * 4) = acc[g];
  __syncthreads();
  // wave (cg, kw < RGB) finishes row group g = kw of its column group: the four k-quarter partials, in order
  const int g = kw;
  const bool fin = kw < RGB;                        // wave-uniform
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (fin) {
    s = *(const f32x4*)(lsm + (((cg * 4 + 0) * RGB + g) * 64 + lane) * 4);
#pragma unroll
    for (int k2 = 1; k2 < 4; ++k2) s += *(const f32x4*)(lsm + (((cg * 4 + k2) * RGB + g) * 64 + lane) * 4);
  }
  if constexpr (KS > 1) {
    constexpr int NP = NCG * RGB;                   // finished (column group, row group) pairs per workgroup
    const int tile = blockIdx.z * gridDim.x + blockIdx.x;
    float* tbase = ws + (size_t)tile * KS * NP * 256;
    const int pr = cg * RGB + g;
    if (fin) {
      float* sp = tbase + ((size_t)blockIdx.y * NP + pr) * 256 + lane * 4;
      st_agent(sp, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its stores
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned)(KS - 1));
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __hip_atomic_store(cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    if (fin) {
      const float* sp = tbase + (size_t)pr * 256 + lane * 4;
      s = ld_agent(sp);
#pragma unroll
      for (int y = 1; y < KS; ++y) s += ld_agent(sp + (size_t)y * NP * 256);
    }
  }
  if (!fin) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = g * 16 + 4 * kq + i, n = row0 + rr;
    if (n >= N) continue;
    float v;
    if (LN) {   // s_i <- r_n (s_i - mu_n c_j) + d_j
      const float tx = (s_sum[0][rr] + s_sum[1][rr]) + (s_sum[2][rr] + s_sum[3][rr]);
      const float tq = (s_sq[0][rr] + s_sq[1][rr]) + (s_sq[2][rr] + s_sq[3][rr]);
      const float mu = tx / (float)K;
      const float var = fmaxf(tq / (float)K - mu * mu, 0.f);
      v = rsqrtf(var + eps) * (s[i] - mu * cj) + bj;
    } else {
      v = s[i] + bj;
    }
    float* o = out + (size_t)n * Nout + j;
    if constexpr (EPI == DE_STORE) *o = v;
    else if constexpr (EPI == DE_RESID) *o += v;
    else if constexpr (EPI == DE_GELU) *o = gelu_new(v);
  }
}

template <int RGB, int NCG, int KS, int EPI, int LN>
static hipError_t dec_gemm_b_launch(const float* W, const float* X, int N, int Nout, int K, const float* bias, float* out,
                                    const float* cvec, float eps, float* ws, unsigned* cnt, hipStream_t s) {
  const dim3 grid(Nout / (16 * NCG), KS, ceil_div(N, 16 * RGB));
  if (K != KS * 768 || Nout % (16 * NCG) != 0 || N < 1 || N > DEC_MAX_PREFIXES) return hipErrorInvalidValue;
  if (KS > 1 && ((int)(grid.x * grid.z) > DEC_MAX_COLGROUPS || (size_t)grid.x * grid.z * KS * NCG * RGB * 256 > DEC_SPLITK_WS_FLOATS || !ws || !cnt))
    return hipErrorInvalidValue;
  constexpr int RB = RGB == 4 ? 8 : 12;
  constexpr int ring = RB * 16 * RGB * 64 * 4, tiles = 4 * NCG * RGB * 1024;
  constexpr int smem = ring > tiles ? ring : tiles;
  static DeviceOnce attr_once; bool& attr_set = attr_once.flag();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_dec_gemm_b<RGB, NCG, KS, EPI, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_dec_gemm_b<RGB, NCG, KS, EPI, LN>), grid, dim3(256 * NCG), smem, s, W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt);
  return hipGetLastError();
}

// ---- layer GEMMs above 64 prefixes (PIO_DEC_SPLIT_MIN_RG = 5 row groups of 16) on split-fp16 operands (PIO_DEC_SPLIT) ------------------------------------------------
// The fp32 MFMA (v_mfma_f32_16x16x4_f32: 256 flop / clk / CU) is what a k_dec_gemm_b workgroup spends its residency on at 64+
// prefixes (tools/microbench: a quarter of the pipe time = -15 % of a 128-prefix decode).  Here both operands are pairs of fp16
// numbers, x S = hi + lo' 2^-11 with hi = fp16(x S) and lo' = fp16((x S - hi) 2^11) -- 22 bits of the 24, the low half kept in the
// normal range by its 2^11 -- and a product is three v_mfma_f32_16x16x32_f16 (hi hi into one accumulator; hi lo' + lo' hi into a
// second one, added with 2^-11 at the end; the lo' lo' term, 2^-22 of the product, is dropped): 48 instead of 256 pipe cycles per
// 32 k.  The weights are split ONCE at load into the same bytes ([column][8-k group][hi x 8 | lo' x 8] fp16: one 32-B read per lane
// and k-step); the activations are split by the workgroup that multiplies them, while it stages its 32 rows into LDS (two fp16
// planes, rows padded to 1552 B: the 16 rows of a fragment read fall on 16 distinct 16-B slots) -- which is also where the
// LayerNorm row sums are taken now, once per element instead of once per column-group wave.
//   grid = (Nout / (16 NCG), KS, ceil(N / 32)), 256 NCG threads: wave (cg, kw) owns 16 columns and the k-steps kw, kw + 4, .. of
//   the workgroup's 768-k slice; the four k-quarter partials meet in LDS in order, then the epilogues of k_dec_gemm_b.
static constexpr float DEC_SPLIT_XS = 1.0f;             // the activations' scale before the split (1: no multiply; see the kernel for rows it does not fit)
__global__ __launch_bounds__(256) void k_dec_split_weights(const float* __restrict__ W, size_t n8, float S, u32x4_t* __restrict__ out) {
  for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n8; g += (size_t)gridDim.x * 256) {
    const float4 a = *(const float4*)(W + g * 8), b = *(const float4*)(W + g * 8 + 4);
    const float v[8] = {a.x * S, a.y * S, a.z * S, a.w * S, b.x * S, b.y * S, b.z * S, b.w * S};
    dec_h8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      hi[i] = (_Float16)v[i];
      lo[i] = (_Float16)((v[i] - (float)hi[i]) * 2048.0f);
    }
    out[2 * g] = __builtin_bit_cast(u32x4_t, hi);
    out[2 * g + 1] = __builtin_bit_cast(u32x4_t, lo);
  }
}
__global__ __launch_bounds__(256) void k_dec_abs_max(const float* __restrict__ x, size_t n, uint32_t* out) {
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint32_t b = __float_as_uint(x[i]) & 0x7FFFFFFFu;      // non-negative floats order like their bits
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(m, o); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
hipError_t launch_dec_split_weights(const float* W, size_t n, float S, void* out, hipStream_t s);
hipError_t dec_split_weights(const float* W, size_t n, void* out, float* unscale, hipStream_t s) {
  uint32_t* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, 4);
  if (e != hipSuccess) return e;
  uint32_t bits = 0;
  e = hipMemsetAsync(d, 0, 4, s);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_dec_abs_max, dim3(512), dim3(256), 0, s, W, n, d); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(&bits, d, 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (e != hipSuccess) return e;
  float amax;
  memcpy(&amax, &bits, 4);
  if (!(amax < 3.0e38f)) return hipErrorInvalidValue;           // NaN / inf weights: keep the fp32 kernels
  int sh = 0;
  if (amax > 0.f) sh = 13 - ilogbf(amax);                       // |W| * 2^sh < 2^14
  const float S = ldexpf(1.0f, sh);
  *unscale = 1.0f / S;
  e = launch_dec_split_weights(W, n, S, out, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}
hipError_t launch_dec_split_weights(const float* W, size_t n, float S, void* out, hipStream_t s) {
  if (n % 8 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_dec_split_weights, dim3(1024), dim3(256), 0, s, W, n / 8, S, (u32x4_t*)out);
  return hipGetLastError();
}

// x (4 floats) -> hi = fp16 toward zero, lo' = fp16((x - hi) 2^11): x 2^11 - hi 2^11 is exact in one FMA (hi is x cut to 11 bits), and the
// hi operand rides in as fp16 (v_fma_mix_f32): a multiply and an FMA per element instead of convert, subtract, multiply.
// lo' is rounded to NEAREST (v_cvt_pk_f16_f32; round 5): cut toward zero like hi it shrank every activation by up to 2^-21 of
// itself, always in the same direction, where the weights' split (round-to-nearest, k_dec_split_weights) has no such bias; now
// |x - (hi + lo' 2^-11)| <= 2^-22 |x| and the error has no sign.  hi stays cut: v_cvt_pkrtz saturates, which the range bookkeeping below relies on.
__device__ __forceinline__ void dec_split4(const float4 x, const float s, dec_h4& hi, dec_h4& lo) {
  const float v0 = x.x * s, v1 = x.y * s, v2 = x.z * s, v3 = x.w * s;
  const dec_h2 h01 = __builtin_bit_cast(dec_h2, __builtin_amdgcn_cvt_pkrtz(v0, v1)), h23 = __builtin_bit_cast(dec_h2, __builtin_amdgcn_cvt_pkrtz(v2, v3));
  const float l0 = __builtin_fmaf((float)h01[0], -2048.0f, v0 * 2048.0f), l1 = __builtin_fmaf((float)h01[1], -2048.0f, v1 * 2048.0f);
  const float l2 = __builtin_fmaf((float)h23[0], -2048.0f, v2 * 2048.0f), l3 = __builtin_fmaf((float)h23[1], -2048.0f, v3 * 2048.0f);
  typedef float dec_f2 __attribute__((ext_vector_type(2)));
  const dec_h2 l01 = __builtin_convertvector((dec_f2){l0, l1}, dec_h2), l23 = __builtin_convertvector((dec_f2){l2, l3}, dec_h2);
  hi = (dec_h4){h01[0], h01[1], h23[0], h23[1]};
  lo = (dec_h4){l01[0], l01[1], l23[0], l23[1]};
}
// the largest |hi| of four halfs, as two packed fp16 maxima (an fp16 inf for an fp32 inf; NaN never wins a maximum)
__device__ __forceinline__ dec_h2 dec_absmax4(const dec_h4 hi, const dec_h2 run) {
  const uint2 b = __builtin_bit_cast(uint2, hi);
  const dec_h2 a = __builtin_bit_cast(dec_h2, b.x & 0x7fff7fffu), c = __builtin_bit_cast(dec_h2, b.y & 0x7fff7fffu);
  return __builtin_elementwise_max(__builtin_elementwise_max(a, c), run);
}

template <int NCG, int KS, int EPI, int LN, int RGB = 2>
__global__ __launch_bounds__(256 * NCG, 1) void k_dec_gemm_s(const u32x4_t* __restrict__ Ws, const float* __restrict__ X, int N, int Nout, int K,
                                                             const float* __restrict__ bias, float* out, const float* __restrict__ cvec,
                                                             float eps, float unscale, float* ws, unsigned* cnt) {
  static_assert(!(LN && KS > 1), "LayerNorm row sums need the whole row in one workgroup");
  constexpr int NW = 4 * NCG, ROWS = 16 * RGB, PSTR = 768 * 2 + 16, PLANE = ROWS * PSTR, NIT = 3 * ROWS / NW;
  static_assert(RGB == 1 || RGB == 2, "16 or 32 rows per workgroup");
  static_assert(NIT * NW == 3 * ROWS, "a row is three 64-lane chunks: they must divide among the waves");
  extern __shared__ __attribute__((aligned(16))) char lss[];       // hi plane, lo plane; afterwards partial tiles [NW][RGB][256] fp32
  __shared__ float s_sum[LN ? ROWS : 1][3], s_sq[LN ? ROWS : 1][3];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int cg = wid >> 2, kw = wid & 3;
  const int row0 = blockIdx.z * ROWS;
  const int kbase = blockIdx.y * 768;
  const int j = (blockIdx.x * NCG + cg) * 16 + li;
  // the weight stream: k-step s = kw + 4 i covers the 8-k groups 4 s .. 4 s + 3, this lane takes group 4 s + kq: 32 B
  const u32x4_t* wp = Ws + ((size_t)j * (K >> 3) + (kbase >> 3) + 4 * kw + kq) * 2;
  u32x4_t wh[6], wl[6];
  // the activations: chunk c = wid + NW it is part c % 3 (64 float4) of row c / 3; split, planes, row sums.
  // Scale.  fp16 holds 65504 (and v_cvt_pkrtz SATURATES, it does not overflow to inf) and is normal down to 2^-14: split as it is, a
  // row keeps its 22 bits when its largest finite magnitude M is in [2^-10, 2^15) -- every activation a GPT-2 style decoder produces
  // (the residual stream's outlier channels reach a few thousand).  Cheap bookkeeping on the way tells whether that can fail; a workgroup where it can -- a caller may hand in any finite
  // prefix, and the fp32 kernels take it -- stages its slice again, every row with its OWN power of two (M 2^-e in [2^13, 2^14);
  // inf / NaN elements do not count and make their row NaN, as in the fp32 kernels), undone in the epilogue.  (Per-row scales for
  // everyone were built first: the second barrier and the conversions waiting behind it cost 2.4 us per kernel; per-chunk maxima in
  // the fast path 1.6 us: three waves share a SIMD, every VALU instruction here is 12 cycles of it.)
  constexpr int XB = NIT % 8 == 0 ? 8 : (NIT % 6 == 0 ? 6 : 4);   // float4 in flight per thread and batch (NIT = 8, 12, 6 or 4)
  static_assert(NIT % XB == 0, "whole batches");
  __shared__ __attribute__((aligned(16))) float s_rmax[ROWS][4];
  __shared__ float s_rs[ROWS];
  __shared__ __attribute__((aligned(16))) float s_am[16];
  float am = 0.f;
  bool small = false;                                   // wave-uniform
  if constexpr (LN) {
    // LayerNorm consumers: a wave stages WHOLE rows (row wid, wid + NW, ..: three float4 per lane and row), so that the row sums are one
    // reduction pair per row instead of one per 64-lane chunk (16 of them per thread were a third of this phase's VALU work, and with three
    // waves on a SIMD every VALU instruction of it is 12 cycles); the last waves stage one row less
    constexpr int RPW = (ROWS + NW - 1) / NW;
    float4 xr[RPW][3];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wid + NW * i;
      if (row < ROWS) {                                   // wave-uniform
        const int rc = row0 + row < N ? row0 + row : N - 1;
#pragma unroll
        for (int part = 0; part < 3; ++part) xr[i][part] = *(const float4*)(X + (size_t)rc * K + kbase + 4 * (64 * part + lane));
      }
    }
    __builtin_amdgcn_sched_barrier(0);                    // loads return in order: the activations first, then the weight stream
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      wh[i] = wp[32 * i];
      wl[i] = wp[32 * i + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wid + NW * i;
      if (row < ROWS) {
        float sx = 0.f, sq = 0.f;
        dec_h2 ra2 = (dec_h2){(_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          const float4 x = xr[i][part];
          dec_h4 hi, lo;
          dec_split4(x, 1.0f, hi, lo);
          char* const d = lss + row * PSTR + 8 * (64 * part + lane);
          *(dec_h4*)d = hi;
          *(dec_h4*)(d + PLANE) = lo;
          ra2 = dec_absmax4(hi, ra2);
          sx += (x.x + x.y) + (x.z + x.w);
          sq += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
        }
        const float ra = fmaxf((float)ra2[0], (float)ra2[1]);
        am = fmaxf(am, ra);
        small |= __builtin_amdgcn_ballot_w64(ra >= 0.0009765625f) == 0ull && __builtin_amdgcn_ballot_w64(ra > 0.f) != 0ull;   // per ROW here
        sx = wave_sum_dpp(sx);
        sq = wave_sum_dpp(sq);
        if (lane == 0) { s_sum[row][0] = sx; s_sum[row][1] = 0.f; s_sum[row][2] = 0.f; s_sq[row][0] = sq; s_sq[row][1] = 0.f; s_sq[row][2] = 0.f; }
      }
    }
  } else {
#pragma unroll
  for (int b0 = 0; b0 < NIT; b0 += XB) {
    float4 xs[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int c = wid + NW * (b0 + i), row = c / 3, part = c - 3 * row;
      const int rc = row0 + row < N ? row0 + row : N - 1;
      xs[i] = *(const float4*)(X + (size_t)rc * K + kbase + 4 * (64 * part + lane));
    }
    if (b0 == 0) {          // loads return in order: the first batch of activations (what the split waits for), then the weight stream
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        wh[i] = wp[32 * i];
        wl[i] = wp[32 * i + 1];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int c = wid + NW * (b0 + i), row = c / 3, part = c - 3 * row;
      const float4 x = xs[i];
      static_assert(DEC_SPLIT_XS == 1.0f, "the fast path splits x itself");
      dec_h4 hi, lo;
      dec_split4(x, 1.0f, hi, lo);
      char* const d = lss + row * PSTR + 8 * (64 * part + lane);
      *(dec_h4*)d = hi;
      *(dec_h4*)(d + PLANE) = lo;
      // range bookkeeping, cheap: the largest magnitude this thread has seen (inf included: it sends the workgroup to the careful path,
      // NaN never wins an fmaxf), and whether some chunk holds values but none of at least 2^-6 (then its row MAY be too small)
      const dec_h2 c2 = dec_absmax4(hi, (dec_h2){(_Float16)0.f, (_Float16)0.f});
      const float a4 = fmaxf((float)c2[0], (float)c2[1]);
      am = fmaxf(am, a4);
      // (2^-10: see above; fc2 / proj add into the residual stream, where the ABSOLUTE error counts and is 2^-35 whatever the row's size, so a
      //  "small" chunk only costs them the careful path)
      small |= __builtin_amdgcn_ballot_w64(a4 >= 0.0009765625f) == 0ull && __builtin_amdgcn_ballot_w64(a4 > 0.f) != 0ull;
    }
  }
  }
  am = row16_max(am);
  am = xor32_max(xor16_max(am));
  if (lane == 0) s_am[wid] = small ? INFINITY : am;     // one word per wave: its largest magnitude, or "look closer"
  __syncthreads();
  bool robust;
  {
    float hiM = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 4) {
      const float4 m4 = *(const float4*)(s_am + w);
      hiM = fmaxf(fmaxf(hiM, fmaxf(m4.x, m4.y)), fmaxf(m4.z, m4.w));
    }
    robust = !(hiM < 32768.f);                          // workgroup-uniform
  }
  if (robust) {
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {                    // the rows' largest FINITE magnitudes
      const int c = wid + NW * it, row = c / 3, part = c - 3 * row;
      const int rc = row0 + row < N ? row0 + row : N - 1;
      const float4 x = *(const float4*)(X + (size_t)rc * K + kbase + 4 * (64 * part + lane));
      const float a0 = fabsf(x.x), a1 = fabsf(x.y), a2 = fabsf(x.z), a3 = fabsf(x.w);
      float m = fmaxf(fmaxf(a0 < 3.0e38f ? a0 : 0.f, a1 < 3.0e38f ? a1 : 0.f), fmaxf(a2 < 3.0e38f ? a2 : 0.f, a3 < 3.0e38f ? a3 : 0.f));
      m = row16_max(m);
      m = xor32_max(xor16_max(m));
      if (lane == 0) s_rmax[row][part] = m;
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
      const int c = wid + NW * it, row = c / 3, part = c - 3 * row;
      const int rc = row0 + row < N ? row0 + row : N - 1;
      const float4 x = *(const float4*)(X + (size_t)rc * K + kbase + 4 * (64 * part + lane));
      const float4 m4 = *(const float4*)s_rmax[row];
      const float m = fmaxf(fmaxf(m4.x, m4.y), m4.z);
      int e = m > 0.f ? ilogbf(m) - 13 : 0;               // m 2^-e in [2^13, 2^14)
      e = e < -100 ? -100 : e;
      const float sc = ldexpf(1.0f, -e);
      if (part == 0 && lane == 0) s_rs[row] = ldexpf(1.0f, e);
      dec_h4 hi, lo;
      dec_split4(x, sc, hi, lo);
      char* const d = lss + row * PSTR + 8 * (64 * part + lane);
      *(dec_h4*)d = hi;
      *(dec_h4*)(d + PLANE) = lo;
    }
    __syncthreads();
  }
  f32x4 a0[RGB], a1[RGB];
#pragma unroll
  for (int g = 0; g < RGB; ++g) { a0[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; a1[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int sgrp = 4 * (kw + 4 * i) + kq;            // this lane's 8-k group of the slice
    const dec_h8 bh = __builtin_bit_cast(dec_h8, wh[i]), bl = __builtin_bit_cast(dec_h8, wl[i]);
#pragma unroll
    for (int g = 0; g < RGB; ++g) {
      const char* rp = lss + (16 * g + li) * PSTR + 16 * sgrp;
      const dec_h8 xh = *(const dec_h8*)rp, xl = *(const dec_h8*)(rp + PLANE);
      a0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bh, a0[g], 0, 0, 0);
      a1[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bl, a1[g], 0, 0, 0);
      a1[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, bh, a1[g], 0, 0, 0);
    }
  }
  const float bj = bias[j];
  const float cj = LN ? cvec[j] : 0.f;
  float rsc[RGB][4];                                // scales of this lane's accumulator rows
#pragma unroll
  for (int g = 0; g < RGB; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) rsc[g][i] = robust ? s_rs[16 * g + 4 * kq + i] : 1.0f / DEC_SPLIT_XS;
  __syncthreads();                                  // every wave is done with the planes: reuse them for the partial tiles
  float* const lsm = (float*)lss;
#pragma unroll
  for (int g = 0; g < RGB; ++g) {
    f32x4 t = (a0[g] + a1[g] * (1.0f / 2048.0f)) * unscale;         // unscale: the weights' scale; the row's own one (a power of two) follows
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] *= rsc[g][i];
    *(f32x4*)(lsm + ((wid * RGB + g) * 64 + lane) * 4) = t;
  }
  __syncthreads();
  const int g = kw;
  const bool fin = kw < RGB;                        // wave-uniform: wave (cg, kw < 2) finishes row group kw of its column group
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (fin) {
    s = *(const f32x4*)(lsm + (((cg * 4 + 0) * RGB + g) * 64 + lane) * 4);
#pragma unroll
    for (int k2 = 1; k2 < 4; ++k2) s += *(const f32x4*)(lsm + (((cg * 4 + k2) * RGB + g) * 64 + lane) * 4);
  }
  if constexpr (KS > 1) {
    constexpr int NP = NCG * RGB;
    const int tile = blockIdx.z * gridDim.x + blockIdx.x;
    float* tbase = ws + (size_t)tile * KS * NP * 256;
    const int pr = cg * RGB + g;
    if (fin) st_agent(tbase + ((size_t)blockIdx.y * NP + pr) * 256 + lane * 4, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned)(KS - 1));
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __hip_atomic_store(cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    if (fin) {
      const float* sp = tbase + (size_t)pr * 256 + lane * 4;
      s = ld_agent(sp);
#pragma unroll
      for (int y = 1; y < KS; ++y) s += ld_agent(sp + (size_t)y * NP * 256);
    }
  }
  if (!fin) return;
  const float rK = 1.0f / (float)K;                 // (a multiplication per row instead of an IEEE division sequence: this kernel's sums are its own anyway)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = g * 16 + 4 * kq + i, n = row0 + rr;
    if (n >= N) continue;
    float v;
    if (LN) {
      const float tx = (s_sum[rr][0] + s_sum[rr][1]) + s_sum[rr][2];
      const float tq = (s_sq[rr][0] + s_sq[rr][1]) + s_sq[rr][2];
      const float mu = tx * rK;
      const float var = fmaxf(tq * rK - mu * mu, 0.f);
      v = rsqrtf(var + eps) * (s[i] - mu * cj) + bj;
    } else {
      v = s[i] + bj;
    }
    float* o = out + (size_t)n * Nout + j;
    if constexpr (EPI == DE_STORE) *o = v;
    else if constexpr (EPI == DE_RESID) *o += v;
    else if constexpr (EPI == DE_GELU) *o = gelu_new(v);
  }
}

template <int NCG, int KS, int EPI, int LN, int RGB = 2>
static hipError_t dec_gemm_s_launch(const void* Ws, float unscale, const float* X, int N, int Nout, int K, const float* bias, float* out,
                                    const float* cvec, float eps, float* ws, unsigned* cnt, hipStream_t s) {
  const dim3 grid(Nout / (16 * NCG), KS, ceil_div(N, 16 * RGB));
  if (K != KS * 768 || Nout % (16 * NCG) != 0 || N < 1 || N > DEC_MAX_PREFIXES || Ws == nullptr) return hipErrorInvalidValue;
  if (KS > 1 && ((int)(grid.x * grid.z) > DEC_MAX_COLGROUPS || (size_t)grid.x * grid.z * KS * NCG * RGB * 256 > DEC_SPLITK_WS_FLOATS || !ws || !cnt))
    return hipErrorInvalidValue;
  constexpr int planes = 2 * 16 * RGB * (768 * 2 + 16), tiles = 4 * NCG * RGB * 1024;
  constexpr int smem = planes > tiles ? planes : tiles;
  static DeviceOnce attr_once; bool& attr_set = attr_once.flag();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_dec_gemm_s<NCG, KS, EPI, LN, RGB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_dec_gemm_s<NCG, KS, EPI, LN, RGB>), grid, dim3(256 * NCG), smem, s, (const u32x4_t*)Ws, X, N, Nout, K, bias, out, cvec, eps,
                     unscale, ws, cnt);
  return hipGetLastError();
}

// K = 768 (4 waves x 12 chunks), 512 (4 x 8) or 3072 (4 workgroups x 4 waves x 12, split-K with `ws` / `cnt`)
#ifndef PIO_DEC_SPLIT_NCG_FC    // column groups (of 16) per workgroup for c_fc / mlp.c_proj in the split-fp16 form
#define PIO_DEC_SPLIT_NCG_FC 3
#endif
#ifndef PIO_DEC_SPLIT_MIN_RG    // 16-row groups from which the layer GEMMs take the split-fp16 form when the split weights exist (5: 65+ prefixes)
#define PIO_DEC_SPLIT_MIN_RG 5
#endif
template <int EPI, int LN>
static hipError_t dec_gemm(const float* W, const float* X, int N, int Nout, int K, const float* bias, float* out,
                           const float* extra, const float* cvec, float eps, float* ws, unsigned* cnt, hipStream_t s,
                           const void* Wsplit = nullptr, float unscale = 0.f) {
  if (N < 1 || N > DEC_MAX_PREFIXES) return hipErrorInvalidValue;
  if constexpr (EPI == DE_STORE || EPI == DE_RESID || EPI == DE_GELU) {
    if (Wsplit != nullptr && ceil_div(N, 16) >= PIO_DEC_SPLIT_MIN_RG && Nout % 192 == 0) {
      // 48 columns per workgroup (12 waves).  64 (16 waves: 192 instead of 256 workgroups for fc / fc2 at 128 prefixes) is 13-16 % slower alone
      // and measured through the pipeline as PIO_DEC_SPLIT_NCG_FC=4: see profiles/r04_bench_sweep.log
      if (K == 768) {
        if (Nout >= 3072) return dec_gemm_s_launch<PIO_DEC_SPLIT_NCG_FC, 1, EPI, LN>(Wsplit, unscale, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
        return dec_gemm_s_launch<3, 1, EPI, LN>(Wsplit, unscale, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
      }
      if constexpr (EPI == DE_RESID && !LN) {
        if (K == 3072 && Nout <= 768 && ws != nullptr && cnt != nullptr)
          return dec_gemm_s_launch<PIO_DEC_SPLIT_NCG_FC, 4, EPI, LN>(Wsplit, unscale, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
      }
    }
  }
  if constexpr (EPI == DE_EMBED) {
    // the prefix projection runs once per decode on k_dec_gemm (<= 128 rows per launch): 129 .. 256 prefixes as two launches
    if (N > 128) {
      hipError_t e = dec_gemm<EPI, LN>(W, X, 128, Nout, K, bias, out, extra, cvec, eps, ws, cnt, s);
      if (e != hipSuccess) return e;
      return dec_gemm<EPI, LN>(W, X + (size_t)128 * K, N - 128, Nout, K, bias, out + (size_t)128 * Nout, extra, cvec, eps, ws, cnt, s);
    }
  }
  if constexpr (PIO_DEC_TILED != 0 && (EPI == DE_STORE || EPI == DE_RESID || EPI == DE_GELU)) {
    const int rg = ceil_div(N, 16);
    if (rg >= 2 && K == 768 && Nout % 48 == 0) {
      const bool wide = Nout >= 2304;       // qkv / fc: 48 (32) columns per workgroup; proj: 16
      if (rg > 4) return wide ? dec_gemm_b_launch<2, 3, 1, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s)
                              : dec_gemm_b_launch<2, 1, 1, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
      if (rg > 2) return wide ? dec_gemm_b_launch<2, 2, 1, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s)
                              : dec_gemm_b_launch<1, 1, 1, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
      return wide ? dec_gemm_b_launch<1, 2, 1, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s)
                  : dec_gemm_b_launch<1, 1, 1, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
    }
    if constexpr (EPI == DE_RESID && !LN) {
      if (rg >= 2 && K == 3072 && Nout % 32 == 0 && Nout <= 1024 && ws != nullptr && cnt != nullptr) {
        if (rg > 4) return dec_gemm_b_launch<4, 2, 4, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
        if (rg > 2) return dec_gemm_b_launch<2, 2, 4, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
        return dec_gemm_b_launch<1, 2, 4, EPI, LN>(W, X, N, Nout, K, bias, out, cvec, eps, ws, cnt, s);
      }
    }
  }
  if (K == 768) return dec_gemm_rg<12, 1, EPI, LN>(W, X, N, Nout, K, bias, out, extra, cvec, eps, ws, cnt, s);
  if (K == 512) return dec_gemm_rg<8, 1, EPI, LN>(W, X, N, Nout, K, bias, out, extra, cvec, eps, ws, cnt, s);
  if (K == 384) return dec_gemm_rg<6, 1, EPI, LN>(W, X, N, Nout, K, bias, out, extra, cvec, eps, ws, cnt, s);   // ViT-S prefix
  if (K == 3072 && EPI == DE_RESID && !LN && ws != nullptr && cnt != nullptr && Nout <= 1024)
    return dec_gemm_rg<12, 4, DE_RESID, 0>(W, X, N, Nout, K, bias, out, extra, cvec, eps, ws, cnt, s);
  return hipErrorInvalidValue;
}

hipError_t decoder_init() { return hipSuccess; }   // no function attributes needed any more

#define PIO_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

// ---- LM head, wide form ------------------------------------------------------------------------------------
// The generic kernel above re-reads all of X (N x 768 fp32) per 16 columns: 3142 workgroups x up to 196 KB through
// L2 -> L1 (at 64 prefixes that, not the 154 MB weight stream or the MFMAs, sets the time: 95 us vs 62 us with the X
// loads compiled out).  Here a workgroup owns 64 columns (wave w: 16 of them, the FULL K), K is walked in 12 chunks
// of 64 and the X chunk ([16 RG][64] fp32, 16-B slots XOR-swizzled with the row so that the ds_read_b128 A fragments
// are conflict-free) is staged ONCE per workgroup by LDS-DMA into a double buffer and shared by the four waves: X
// traffic / 4, no cross-wave reduction.  W streams HBM -> registers, three chunk sets deep (two chunks ahead).
// Numerics are BIT-IDENTICAL to k_dec_gemm<.., DE_ARGMAX, LN>: the same lane <-> (column, k) mapping, the same two
// MFMA chains per 192-k slice, slices added in order, LayerNorm row sums built per slice in the same order.
template <int RG>
__global__ __launch_bounds__(256, 2) void k_lmhead_wide(const float* __restrict__ W, const float* __restrict__ X, int N, int V,
                                                        const float* __restrict__ dvec, const float* __restrict__ cvec,
                                                        float eps, float* part) {
  constexpr int K = 768, CH = 64, NCH = K / CH, ROWS = RG * 16, XB = ROWS * CH;
  extern __shared__ __attribute__((aligned(16))) float lsm[];      // [2][ROWS][64] X chunks | s_sum[4][ROWS] | s_sq[4][ROWS]
  float* s_sum = lsm + 2 * XB;
  float* s_sq = s_sum + 4 * ROWS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int blk = blockIdx.x * 4 + wid;            // 16-column group of this wave
  const int j = blk * 16 + li;
  const int jc = j < V ? j : V - 1;
  const float* wp = W + (size_t)jc * K + 4 * kq;

  // LDS-DMA pieces of an X chunk: piece t = wid + 4 i covers rows 4t .. 4t+3 (1 KiB); lane l fills slot (l & 15) of
  // row 4t + (l >> 4) and therefore fetches source chunk (l & 15) ^ (row & 15).
  uint32_t xoff[RG];
#pragma unroll
  for (int i = 0; i < RG; ++i) {
    const int row = 4 * (wid + 4 * i) + (lane >> 4);
    const int rc = row < N ? row : N - 1;
    xoff[i] = (uint32_t)rc * K + 4 * ((lane & 15) ^ (row & 15));
  }
  // (LDS-DMA in inline asm as well: hipcc drains vmcnt(0) before every ds_read that follows a builtin LDS-DMA.
  //  M0 = wave-uniform LDS byte address of the piece; saved and restored inside the statement.)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(dec_lds_ptr_t)lsm + (uint32_t)wid * 1024u;
#define PIO_XISSUE(q, buf)                                                                                     \
  do {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < RG; ++i) {                                                           \
      const float* _g = X + (q) * CH + xoff[i];                                                                \
      const uint32_t _l = lds0 + (uint32_t)((buf) * XB * 4 + i * 4096);                                        \
      uint32_t _keep;                                                                                          \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                   : "=&s"(_keep) : "v"(_g), "s"(_l) : "memory");                                              \
    }                                                                                                          \
  } while (0)
  // The weight loads are inline asm: beside an LDS-DMA in flight hipcc waits vmcnt(0) before the first use of ANY
  // ordinary load result, which would drain the two-chunk prefetch every chunk.  hipcc does not count asm loads, so
  // the vm queue is counted by hand (PIO_WWAIT names the destinations "+v": no consumer is scheduled above it).
  // Queue order per chunk: W(q) | X(q) | W(q+1): with <= 4 outstanding, W(q) and X(q) have landed.
  // 64-prefix form (the grouped decode, which runs beside the 1.8 GB bank stream of the next batches): non-temporal,
  // measured +1.5 % captions/s pipelined; at <= 32 prefixes (one synchronous forward) part of the 154 MB head is
  // still in the Infinity Cache from the previous step and the default policy is 2.5 % faster end to end.
#define PIO_WLOAD(set, q)                                                                                      \
  do {                                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                            \
      const float* _p = wp + (q) * CH + 16 * c;                                                                \
      if constexpr (RG == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(w[set][c]) : "v"(_p) : "memory"); \
      else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[set][c]) : "v"(_p) : "memory");             \
    }                                                                                                          \
  } while (0)
#define PIO_WWAIT(set, cnt)                                                                                    \
  asm volatile("s_waitcnt vmcnt(" #cnt ")" : "+v"(w[set][0]), "+v"(w[set][1]), "+v"(w[set][2]), "+v"(w[set][3]) :: "memory")

  f32x4 w[3][4];
  f32x4 tot[RG], a0[RG], a1[RG];
  float sx[RG], sq[RG];
  PIO_WLOAD(0, 0);
  PIO_XISSUE(0, 0);
  PIO_WLOAD(1, 1);
  // Fully unrolled on purpose: with a runtime loop hipcc rotates the weight sets through v_mov copies at the back
  // edge, i.e. it copies registers whose asm loads have not landed yet (audit: tools/microbench/asm_load_audit.py).
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {                 // 192-k slices
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      a0[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; a1[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      sx[g] = 0.f; sq[g] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {                  // chunk q = 3 sl + r uses weight set r
      const int q = 3 * sl + r;
      // W(q) and X chunk q have landed once at most the 4 weight loads issued after them are outstanding
      if (q + 1 < NCH) PIO_WWAIT(r, 4);
      else PIO_WWAIT(r, 0);
      __builtin_amdgcn_s_barrier();                // ... in every wave; and every wave is done with buffer (q+1)&1
      if (q + 1 < NCH) PIO_XISSUE(q + 1, (q + 1) & 1);
      if (q + 2 < NCH) PIO_WLOAD((r + 2) % 3, q + 2);
      const float* xb = lsm + (q & 1) * XB;
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        float4 xf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xf[c] = *(const float4*)(xb + (16 * g + li) * CH + (((4 * c + kq) ^ li) << 2));
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          a0[g] = mfma16f(xf[c].x, w[r][c][0], a0[g]);  a1[g] = mfma16f(xf[c + 1].x, w[r][c + 1][0], a1[g]);
          a0[g] = mfma16f(xf[c].y, w[r][c][1], a0[g]);  a1[g] = mfma16f(xf[c + 1].y, w[r][c + 1][1], a1[g]);
          a0[g] = mfma16f(xf[c].z, w[r][c][2], a0[g]);  a1[g] = mfma16f(xf[c + 1].z, w[r][c + 1][2], a1[g]);
          a0[g] = mfma16f(xf[c].w, w[r][c][3], a0[g]);  a1[g] = mfma16f(xf[c + 1].w, w[r][c + 1][3], a1[g]);
          if (wid == (g & 3)) {     // the LayerNorm row sums of group g are built by one wave only
            sx[g] += ((xf[c].x + xf[c].y) + (xf[c].z + xf[c].w)) + ((xf[c + 1].x + xf[c + 1].y) + (xf[c + 1].z + xf[c + 1].w));
            sq[g] += ((xf[c].x * xf[c].x + xf[c].y * xf[c].y) + (xf[c].z * xf[c].z + xf[c].w * xf[c].w)) +
                     ((xf[c + 1].x * xf[c + 1].x + xf[c + 1].y * xf[c + 1].y) + (xf[c + 1].z * xf[c + 1].z + xf[c + 1].w * xf[c + 1].w));
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < RG; ++g) {                 // the slice is complete
      const f32x4 p = a0[g] + a1[g];
      tot[g] = sl == 0 ? p : tot[g] + p;
      if (wid == (g & 3)) {
        float tx = sx[g], tq = sq[g];
        tx = xor32_add(xor16_add(tx));
        tq = xor32_add(xor16_add(tq));
        if (kq == 0) { s_sum[sl * ROWS + g * 16 + li] = tx; s_sq[sl * ROWS + g * 16 + li] = tq; }
      }
    }
  }
#undef PIO_XISSUE
#undef PIO_WLOAD
#undef PIO_WWAIT
  const float bj = dvec[jc], cj = cvec[jc];
  __syncthreads();                                 // row sums visible
  if (blk * 16 >= V) return;                       // wave-uniform: no columns
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    f32x4 sacc = tot[g];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = g * 16 + 4 * kq + i;
      const float tx = (s_sum[0 * ROWS + rr] + s_sum[1 * ROWS + rr]) + (s_sum[2 * ROWS + rr] + s_sum[3 * ROWS + rr]);
      const float tq = (s_sq[0 * ROWS + rr] + s_sq[1 * ROWS + rr]) + (s_sq[2 * ROWS + rr] + s_sq[3 * ROWS + rr]);
      const float mu = tx / (float)K;
      const float var = fmaxf(tq / (float)K - mu * mu, 0.f);
      sacc[i] = rsqrtf(var + eps) * (sacc[i] - mu * cj) + bj;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = j < V ? sacc[i] : -INFINITY;
      int idx = j;
      float mx = v;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float ov = __shfl_xor(mx, o);
        const int oi = __shfl_xor(idx, o);
        if (arg_better(ov, oi, mx, idx)) { mx = ov; idx = oi; }
      }
      float se = j < V ? expf(v - mx) : 0.f;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) se += __shfl_xor(se, o);
      const int n = g * 16 + 4 * kq + i;
      if (li == 0 && n < N) {
        float* p = part + ((size_t)blk * N + n) * 4;
        p[0] = mx; p[1] = __int_as_float(idx); p[2] = se;
      }
    }
  }
}

template <int RG>
static hipError_t launch_lmhead_wide(const float* W, const float* X, int N, int V, const float* dvec, const float* cvec,
                                     float eps, float* part, hipStream_t s) {
  const int smem = (2 * RG * 16 * 64 + 8 * RG * 16) * 4;
  hipLaunchKernelGGL((k_lmhead_wide<RG>), dim3(ceil_div(V, 64)), dim3(256), smem, s, W, X, N, V, dvec, cvec, eps, part);
  return hipGetLastError();
}

// ---- LM head with an fp16 filter (greedy ids without log-probabilities) ---------------------------------------
// At 64 prefixes the exact head is bound by the fp32 MFMA rate (4.9 GFLOP per step), at 16 by its 154 MB of fp32
// weights.  Only the arg-max is needed, so:
//   k_lm_prep        per row: LayerNorm statistics (mu, rstd), a power-of-two scale that brings the row into fp16
//                    range, the fp16 copy of the row, and the row's error bound (below)
//   k_lmhead_f16     APPROXIMATE logits for all columns from fp16 operands (v_mfma_f32_16x16x32_f16, fp32
//                    accumulation, half the bytes, 1/16 of the MFMA time); same workgroup shape as k_lmhead_wide
//   k_dec_select_filter   per row: max of the approximate logits; every column within 2 * bound of it is
//                    re-evaluated EXACTLY (fp32 dot with the fp32 weights) and the arg-max is taken over those.
// Bound.  With u = 2^-11 (fp16 rounding of x~ = x * 2^-e and of W~ = W' * 2^s) and both accumulations in fp32 over
// K = 768 terms (gamma_768 < 4.6e-5 each), |S~ - S| <= (2u + u^2 + 2 gamma) * sum_k |x_k W'_vk| < 1.07e-3 * |x|_2 |W'_v|_2
// (Cauchy-Schwarz); fp16 subnormals add at most 2^-25 per term in scaled units, < 1e-9 of that.  The logit is
// rstd * (S - mu c_v) + d_v, so |logit~ - logit| <= 1.25e-3 * rstd * |x|_2 * max_v |W'_v|_2 =: B (margin included).
// If v* is the exact arg-max then logit~[v*] >= logit[v*] - B >= logit[v~] - B >= logit~[v~] - 2B for the approximate
// arg-max v~: v* (and every exact tie) passes the filter.  A further slack of 1e-5 (1 + |max|) absorbs the fp32
// rounding of the affine step.  NaN rows decode to id 0 like torch.argmax.

__global__ __launch_bounds__(256) void k_lm_prep(const float* __restrict__ x, int K, float eps, float bound_coef,
                                                 _Float16* __restrict__ xh, float* __restrict__ stats) {
  __shared__ float s_a[4], s_b[4], s_c[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float v[3], sum = 0.f, sq = 0.f, amax = 0.f;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int k = tid + 256 * i;
    v[i] = k < K ? x[(size_t)n * K + k] : 0.f;
    sum += v[i]; sq += v[i] * v[i]; amax = fmaxf(amax, fabsf(v[i]));
    bad |= !(fabsf(v[i]) < 3.0e38f);
  }
  sum = wave_sum(sum); sq = wave_sum(sq); amax = wave_max(amax);
  const bool wbad = __any(bad);
  if (lane == 0) { s_a[wid] = sum; s_b[wid] = sq; s_c[wid] = wbad ? INFINITY : amax; }
  __syncthreads();
  sum = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
  sq = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
  amax = fmaxf(fmaxf(s_c[0], s_c[1]), fmaxf(s_c[2], s_c[3]));
  const bool finite = amax < 3.0e38f;
  const float mu = sum / (float)K;
  const float var = fmaxf(sq / (float)K - mu * mu, 0.f);
  const float rstd = rsqrtf(var + eps);
  int e = 0;
  if (finite && amax > 0.f) e = ilogbf(amax) + 1 - 14;          // |x| * 2^-e <= 2^14
  const float down = ldexpf(1.0f, -e);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int k = tid + 256 * i;
    if (k < K) xh[(size_t)n * K + k] = (_Float16)(finite ? v[i] * down : 0.f);
  }
  if (tid == 0) {
    float* st = stats + 4 * n;
    st[0] = mu; st[1] = rstd; st[2] = ldexpf(1.0f, e);
    st[3] = finite ? bound_coef * rstd * sqrtf(sq) : NAN;       // NaN marks a row without finite logits
  }
}

// CPW: 16-column groups per wave (a fragment read from the X~ chunk then feeds CPW MFMAs).  Built to test whether the kernel is bound by its LDS
// fragment reads at 128 rows (16 KB per wave and chunk): it is not -- see PIO_LMF16_CPW8.
template <int RG, int NWV, int CPW = 1>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void k_lmhead_f16(const uint16_t* __restrict__ W16, const _Float16* __restrict__ Xh, int N, int V,
                                                       int Vp, const float* __restrict__ stats, const float* __restrict__ dvec,
                                                       const float* __restrict__ cvec, float w_unscale, float* __restrict__ out,
                                                       float* __restrict__ gmax, int NGp) {
  constexpr int K = 768, CH = 64, NCH = K / CH, ROWS = RG * 16, XB = ROWS * CH * 2;   // XB: bytes per X~ chunk
  extern __shared__ __attribute__((aligned(16))) char lsh[];       // [2][ROWS][64] fp16, 16-B slots XORed with (row>>1)&7
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int blk0 = (blockIdx.x * NWV + wid) * CPW;      // this wave's first column group
  const uint16_t* wp[CPW];
#pragma unroll
  for (int cgi = 0; cgi < CPW; ++cgi) {
    const int jj = (blk0 + cgi) * 16 + li;
    wp[cgi] = W16 + (size_t)(jj < V ? jj : V - 1) * K + 8 * kq;
  }
  // LDS-DMA pieces of an X~ chunk: 1 KiB = 8 rows x 128 B; pieces wid, wid + NWV, ...
  constexpr int NP = ROWS / 8, PPW = (NP + NWV - 1) / NWV;
  uint32_t xoff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int row = 8 * (wid + NWV * i) + (lane >> 3);
    const int rc = row < N ? row : N - 1;
    xoff[i] = ((uint32_t)rc * K) * 2 + 16 * ((lane & 7) ^ ((row >> 1) & 7));
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(dec_lds_ptr_t)lsh + (uint32_t)wid * 1024u;
#define PIO_XISSUE(q, buf)                                                                                     \
  do {                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                          \
      if (NP % NWV == 0 || wid + NWV * i < NP) {                                           \
        const char* _g = (const char*)Xh + (q) * (CH * 2) + xoff[i];                                           \
        const uint32_t _l = lds0 + (uint32_t)((buf) * XB + i * NWV * 1024);                                          \
        uint32_t _keep;                                                                                        \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(_keep) : "v"(_g), "s"(_l) : "memory");                                            \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
  // weights: 2 x 16 B per lane and chunk (k = 64 q + 32 c + 8 kq ..), three chunk sets in flight (inline asm, counted by hand)
#define PIO_WLOAD(set, q)                                                                                      \
  do {                                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 2 * CPW; ++c) {                                                      \
      const uint16_t* _p = wp[c >> 1] + (q) * CH + 32 * (c & 1);                                               \
      if constexpr (PIO_LMF16_NT(RG)) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(w[set][c]) : "v"(_p) : "memory"); \
      else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[set][c]) : "v"(_p) : "memory");            \
    }                                                                                                          \
  } while (0)
#define PIO_WWAIT(set, cnt)                                                                                    \
  do {                                                                                                         \
    if constexpr (CPW == 1) asm volatile("s_waitcnt vmcnt(" #cnt ")" : "+v"(w[set][0]), "+v"(w[set][1]) :: "memory");             \
    else asm volatile("s_waitcnt vmcnt(" #cnt ")" : "+v"(w[set][0]), "+v"(w[set][1]), "+v"(w[set][2 * CPW - 2]), "+v"(w[set][2 * CPW - 1]) :: "memory"); \
  } while (0)
  // RG >= 4 (64 / 128 prefixes): the X~ chunk is 8 / 16 KB and a chunk's MFMAs last ~0.1 us, far less than an L2 round
  // trip, so one chunk of lookahead left the workgroup waiting on every chunk (50 us for 77 MB at 128 prefixes): ring of
  // DEPTH + 1 buffers and weight sets, DEPTH chunks in flight.  Every wave issues PPW pieces per chunk there (NP % 4 == 0).
  constexpr bool DEEP = PIO_LMF16_DEEP != 0 && RG >= 4 && CPW == 1;
  static_assert(CPW == 1 || CPW == 2, "column groups per wave");
  constexpr int DEPTH = DEEP ? 3 : 1, NSET = DEEP ? DEPTH + 1 : 3;
  static_assert(!DEEP || NP % NWV == 0, "uniform vm queue");
  f32x4 w[NSET][2 * CPW];
  f32x4 acc[CPW][RG];
#pragma unroll
  for (int cgi = 0; cgi < CPW; ++cgi)
#pragma unroll
    for (int g = 0; g < RG; ++g) acc[cgi][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define PIO_COMPUTE(q, set, buf)                                                                               \
  do {                                                                                                         \
    const char* xb = lsh + (buf) * XB;                                                                         \
    _Pragma("unroll") for (int g = 0; g < RG; ++g) {                                                           \
      _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                          \
        const int row = 16 * g + li;                                                                           \
        const dec_h8 xf = *(const dec_h8*)(xb + row * 128 + (((4 * c + kq) ^ ((row >> 1) & 7)) << 4));         \
        _Pragma("unroll") for (int cgi = 0; cgi < CPW; ++cgi) {                                                \
          acc[cgi][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, __builtin_bit_cast(dec_h8, w[set][2 * cgi + c]), acc[cgi][g], 0, 0, 0); \
        }                                                                                                      \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
  if constexpr (DEEP) {
    // queue: W(0) X(0) .. W(DEPTH-1) X(DEPTH-1) | step q: [wait chunk q] [barrier] W(q+DEPTH) X(q+DEPTH) [compute q].
    // At the wait of step q chunks q+1 .. min(q+DEPTH-1, 11) are behind chunk q: (2 + PPW) operations each.
#define PIO_DSTEP(q)                                                                                           \
    do {                                                                                                       \
      constexpr int _last = (q) + DEPTH - 1 < NCH - 1 ? (q) + DEPTH - 1 : NCH - 1;                             \
      asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[(q) % NSET][0]), "+v"(w[(q) % NSET][1]) : "n"((_last - (q)) * (2 + PPW)) : "memory"); \
      __builtin_amdgcn_s_barrier();    /* chunk q landed in every wave; every wave is done with chunk q-1 */    \
      if ((q) + DEPTH < NCH) { PIO_WLOAD(((q) + DEPTH) % NSET, (q) + DEPTH); PIO_XISSUE((q) + DEPTH, ((q) + DEPTH) % NSET); } \
      PIO_COMPUTE(q, (q) % NSET, (q) % NSET);                                                                  \
    } while (0)
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) { PIO_WLOAD(q, q); PIO_XISSUE(q, q); }
    PIO_DSTEP(0); PIO_DSTEP(1); PIO_DSTEP(2); PIO_DSTEP(3); PIO_DSTEP(4); PIO_DSTEP(5);
    PIO_DSTEP(6); PIO_DSTEP(7); PIO_DSTEP(8); PIO_DSTEP(9); PIO_DSTEP(10); PIO_DSTEP(11);
#undef PIO_DSTEP
  } else {
    PIO_WLOAD(0, 0);
    PIO_XISSUE(0, 0);
    PIO_WLOAD(1, 1);
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int r = q % 3;
      // queue per chunk: W(q) | X(q) | W(q+1): W(q), X(q) have landed once only the 2 loads of W(q+1) are outstanding
      if (q + 1 < NCH) { if constexpr (CPW == 1) PIO_WWAIT(r, 2); else PIO_WWAIT(r, 4); }
      else PIO_WWAIT(r, 0);
      __builtin_amdgcn_s_barrier();
      if (q + 1 < NCH) PIO_XISSUE(q + 1, (q + 1) & 1);
      if (q + 2 < NCH) PIO_WLOAD((r + 2) % 3, q + 2);
      PIO_COMPUTE(q, r, q & 1);
    }
  }
#undef PIO_COMPUTE
#undef PIO_XISSUE
#undef PIO_WLOAD
#undef PIO_WWAIT
  // the rows' statistics through LDS (one round trip instead of 4 RG loads per lane).  The epilogue was 17 of this
  // kernel's 50 us at 128 prefixes, most of it the four ds_bpermute round trips per group maximum: row16_max (DPP) -> 46 us.
  // (Keeping only the group maxima, no approximate logits: 40.6 us, but k_dec_select_filter then evaluates all 16
  //  columns of a candidate group exactly, 12.1 instead of 5.8 us: no net gain, dropped.)
  __syncthreads();                                  // every wave is done with the X~ ring
  float4* s_st = (float4*)lsh;
  if (tid < ROWS) s_st[tid] = *(const float4*)(stats + 4 * (tid < N ? tid : N - 1));
  __syncthreads();
#pragma unroll
  for (int cgi = 0; cgi < CPW; ++cgi) {
    const int blk = blk0 + cgi, j = blk * 16 + li, jc = j < V ? j : V - 1;
    if (blk * 16 >= V) break;                         // wave-uniform: no columns
    const float cj = cvec[jc], dj = dvec[jc];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = 16 * g + 4 * kq + i;
        const float4 st = s_st[n];
        const float v = j < V ? st.y * (acc[cgi][g][i] * (st.z * w_unscale) - st.x * cj) + dj : -INFINITY;
        if (n < N && j < V) out[(size_t)n * Vp + j] = v;
        const float gm = row16_max(v);                // max over the wave's 16 columns (lanes li = 0..15 of this kq group)
        if (li == 0 && n < N) gmax[(size_t)n * NGp + blk] = gm;
      }
  }
}

// Per row: max of the approximate logits (from the per-16-column group maxima), exact re-evaluation of every column
// within the bound, next-step input.  Groups whose maximum passes the threshold go to a list in LDS (a handful per
// row); should more than CAND groups pass, every group is walked instead (same result, slower).
// One row of the filter: 256 threads.  AGENT: the approximate logits and group maxima were written by other workgroups of THIS launch
// (k_lmhead_f16_fused<true>): relaxed agent-scope loads read the coherent copy, as in the split-K exchange above.
template <bool AGENT>
__device__ __forceinline__ float ld_maybe_agent(const float* p) {
  if constexpr (AGENT) return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
}
template <bool AGENT>
__device__ __forceinline__ void dec_select_row(const int n, const float4 st, const float* approx, const float* gmax, int V, int Vp, int NG, int NGp,
                                               const float* xrow, const float* __restrict__ W /*[V][E] LN-folded*/,
                                               const float* __restrict__ dvec, const float* __restrict__ cvec, int E, int step, int steps,
                                               const float* __restrict__ wte, const float* __restrict__ wpe, int32_t* ids, float* x, int pos_base) {
  constexpr int CAND = 512;
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ int s_cnt;
  __shared__ int s_list[CAND];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* a = approx + (size_t)n * Vp;
  const float* gm = gmax + (size_t)n * NGp;
  int best_i = 0;
  if (st.w == st.w) {                                   // finite row (block-uniform)
    float gv[16];                                       // this thread's group maxima (NG <= 4096), all loads in flight
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int gi = tid + 256 * k;
      gv[k] = gi < NG ? ld_maybe_agent<AGENT>(gm + gi) : -INFINITY;
      mx = fmaxf(mx, gv[k]);
    }
    if (tid == 0) s_cnt = 0;
    mx = wave_max(mx);
    if (lane == 0) s_v[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3]));
    const float thr = mx - 2.0f * st.w - 1e-5f * (1.0f + fabsf(mx));
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (gv[k] >= thr) {
        const int slot = atomicAdd(&s_cnt, 1);
        if (slot < CAND) s_list[slot] = tid + 256 * k;
      }
    __syncthreads();
    const int cnt = s_cnt;
    const bool listed = cnt <= CAND;
    const int ngroups = listed ? cnt : NG;
    // this row of x, 12 values per lane (every wave keeps its own copy)
    float4 xr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) xr[i] = *(const float4*)(xrow + (size_t)n * E + 4 * lane + 256 * i);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = wid; c < ngroups; c += 4) {            // wave-uniform loop over candidate groups
      const int gi = listed ? s_list[c] : c;
      const int v = gi * 16 + (lane & 15);
      const float val = (lane < 16 && v < V) ? ld_maybe_agent<AGENT>(a + v) : -INFINITY;
      unsigned long long m = __ballot(val >= thr);
      while (m) {                                       // all lanes evaluate candidate column vc exactly
        const int vc = gi * 16 + __builtin_ctzll(m);
        m &= m - 1;
        const float* wr = W + (size_t)vc * E + 4 * lane;
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float4 ww = *(const float4*)(wr + 256 * i);
          sacc = fmaf(xr[i].x, ww.x, sacc); sacc = fmaf(xr[i].y, ww.y, sacc);
          sacc = fmaf(xr[i].z, ww.z, sacc); sacc = fmaf(xr[i].w, ww.w, sacc);
        }
        sacc = wave_sum(sacc);
        const float exact = st.y * (sacc - st.x * cvec[vc]) + dvec[vc];
        if (arg_better(exact, vc, bv, bi)) { bv = exact; bi = vc; }
      }
    }
    __syncthreads();
    if (lane == 0) { s_v[wid] = bv; s_i[wid] = bi; }
    __syncthreads();
    bv = s_v[0]; bi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (arg_better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
    best_i = bi < V ? bi : 0;
  }
  for (int d = tid; d < E; d += 256) x[(size_t)n * E + d] = wte[(size_t)best_i * E + d] + wpe[(size_t)(pos_base + step + 1) * E + d];
  if (tid == 0) ids[(size_t)n * steps + step] = best_i;
}

__global__ __launch_bounds__(256) void k_dec_select_filter(const float* __restrict__ approx, const float* __restrict__ gmax, int V, int Vp,
                                                           int NG, int NGp, const float* __restrict__ stats,
                                                           const float* xrow, const float* __restrict__ W /*[V][E] LN-folded*/,
                                                           const float* __restrict__ dvec, const float* __restrict__ cvec, int E, int step,
                                                           int steps, const float* __restrict__ wte, const float* __restrict__ wpe,
                                                           int32_t* ids, float* x, int pos_base) {
  const int n = blockIdx.x;
  dec_select_row<false>(n, *(const float4*)(stats + 4 * n), approx, gmax, V, Vp, NG, NGp, xrow, W, dvec, cvec, E, step, steps, wte, wpe, ids, x, pos_base);
}

// <= 16 prefixes: k_lm_prep folded into the head (one kernel less per step).  Every workgroup recomputes the rows'
// LayerNorm statistics and fp16 copies (49 KB of x from L2) and keeps ALL of X~ (16 x 768 fp16 = 24 KB, the same
// chunked, swizzled image as above) in LDS: nothing to stage per chunk, no barrier in the K loop.  Workgroup 0
// publishes the statistics for k_dec_select_filter.
//
// TAIL (round 5): k_dec_select_filter as the tail of this kernel -- one launch less per step.  Every workgroup stores its approximate
// logits and group maxima by relaxed agent-scope atomic stores (written through: the split-K exchange's idiom above), drains them, and
// draws a ticket; the LAST N arrivals each take one row: they wait until the ticket says that every workgroup has arrived (only the last
// N ever wait, at most N - 1 of them, for workgroups that are running or about to: the grid drains whatever the residency), read
// the maxima / candidates back by agent-scope loads and run dec_select_row.  The statistics of the row are the workgroup's own (every
// workgroup computes all 16 rows').  `tk[0]` = ticket, `tk[1]` = rows done; the tail that finishes last re-arms both for the next step.
// x is read by every workgroup before its ticket and written (the next step's input) only behind the full count.
struct LmTail {
  unsigned* tk; const float* W32; int E, step, steps; const float* wte; const float* wpe; int32_t* ids; float* xnext; int pos_base, NG;
};
template <bool TAIL>
__global__ __launch_bounds__(256, 2) void k_lmhead_f16_fused(const uint16_t* __restrict__ W16, const float* x, int N, int V,
                                                             int Vp, float eps, float bound_coef, float* __restrict__ stats,
                                                             const float* __restrict__ dvec, const float* __restrict__ cvec,
                                                             float w_unscale, float* out, float* gmax, int NGp, const LmTail tl) {
  constexpr int K = 768, CH = 64, NCH = K / CH;
  __shared__ __attribute__((aligned(16))) char xs[NCH * 16 * 128];      // [chunk][row][64 fp16], 16-B slots XORed with (row>>1)&7
  __shared__ __attribute__((aligned(16))) float s_st[16][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int blk = blockIdx.x * 4 + wid;
  const int j = blk * 16 + li;
  const int jc = j < V ? j : V - 1;
  const uint16_t* wp = W16 + (size_t)jc * K + 8 * kq;
#define PIO_WLOAD(set, q)                                                                                      \
  do {                                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                            \
      const uint16_t* _p = wp + (q) * CH + 32 * c;                                                             \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[set][c]) : "v"(_p) : "memory");                  \
    }                                                                                                          \
  } while (0)
#define PIO_WWAIT(set, cnt) asm volatile("s_waitcnt vmcnt(" #cnt ")" : "+v"(w[set][0]), "+v"(w[set][1]) :: "memory")
  f32x4 w[3][2];
  // ---- rows 4 wid .. 4 wid + 3: statistics, scale, fp16 image ----
  float4 xv[4][3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * wid + r, rc = row < N ? row : N - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) xv[r][i] = *(const float4*)(x + (size_t)rc * K + 4 * lane + 256 * i);
  }
  PIO_WLOAD(0, 0);
  PIO_WLOAD(1, 1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * wid + r;
    float sum = 0.f, sq = 0.f, amax = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 v = xv[r][i];
      sum += (v.x + v.y) + (v.z + v.w);
      sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      bad |= !(fabsf(v.x) < 3.0e38f) || !(fabsf(v.y) < 3.0e38f) || !(fabsf(v.z) < 3.0e38f) || !(fabsf(v.w) < 3.0e38f);
    }
    sum = wave_sum(sum); sq = wave_sum(sq); amax = wave_max(amax);
    const bool finite = !__any(bad);
    const float mu = sum / (float)K;
    const float var = fmaxf(sq / (float)K - mu * mu, 0.f);
    const float rstd = rsqrtf(var + eps);
    int e = 0;
    if (finite && amax > 0.f) e = ilogbf(amax) + 1 - 14;
    const float down = ldexpf(1.0f, -e);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int k = 4 * lane + 256 * i;                         // 4 consecutive k inside one 16-B slot
      const float4 v = xv[r][i];
      h4 hv;
      hv[0] = (_Float16)(finite ? v.x * down : 0.f); hv[1] = (_Float16)(finite ? v.y * down : 0.f);
      hv[2] = (_Float16)(finite ? v.z * down : 0.f); hv[3] = (_Float16)(finite ? v.w * down : 0.f);
      const int q = k >> 6, slot = (k & 63) >> 3;
      *(h4*)(xs + q * 2048 + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4) + (k & 7) * 2) = hv;
    }
    if (lane == 0) {
      const float bnd = finite ? bound_coef * rstd * sqrtf(sq) : NAN;
      s_st[row][0] = mu; s_st[row][1] = rstd; s_st[row][2] = ldexpf(1.0f, e); s_st[row][3] = bnd;
      if (blockIdx.x == 0 && row < N) {
        float* st = stats + 4 * row;
        st[0] = mu; st[1] = rstd; st[2] = ldexpf(1.0f, e); st[3] = bnd;
      }
    }
  }
  __syncthreads();
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int r = q % 3;
    if (q + 1 < NCH) PIO_WWAIT(r, 2);            // queue: W(q) | W(q+1)
    else PIO_WWAIT(r, 0);
    if (q + 2 < NCH) PIO_WLOAD((r + 2) % 3, q + 2);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const dec_h8 xf = *(const dec_h8*)(xs + q * 2048 + li * 128 + (((4 * c + kq) ^ ((li >> 1) & 7)) << 4));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, __builtin_bit_cast(dec_h8, w[r][c]), acc, 0, 0, 0);
    }
  }
#undef PIO_WLOAD
#undef PIO_WWAIT
  if (blk * 16 < V) {                                 // wave-uniform (the last workgroup's waves past the vocabulary have no columns)
    const float cj = cvec[jc], dj = dvec[jc];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 4 * kq + i;
      const float4 st = *(const float4*)s_st[n];
      const float v = j < V ? st.y * (acc[i] * (st.z * w_unscale) - st.x * cj) + dj : -INFINITY;
      const float gm = row16_max(v);
      if constexpr (TAIL) {
        if (n < N && j < V) __hip_atomic_store((unsigned*)(out + (size_t)n * Vp + j), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (li == 0 && n < N) __hip_atomic_store((unsigned*)(gmax + (size_t)n * NGp + blk), __float_as_uint(gm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (n < N && j < V) out[(size_t)n * Vp + j] = v;
        if (li == 0 && n < N) gmax[(size_t)n * NGp + blk] = gm;
      }
    }
  }
  if constexpr (TAIL) {
    __shared__ unsigned s_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its stores (written through)
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(tl.tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned total = gridDim.x, t = s_ticket;
    if (t + (unsigned)N < total) return;                        // not one of the last N arrivals (block-uniform)
    const int n = (int)(t + (unsigned)N - total);               // 0 .. N - 1 (the host launches this form with N <= gridDim.x)
    if (tid == 0)
      while (__hip_atomic_load(tl.tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    dec_select_row<true>(n, *(const float4*)s_st[n], out, gmax, V, Vp, tl.NG, NGp, x, tl.W32, dvec, cvec, tl.E, tl.step, tl.steps, tl.wte, tl.wpe, tl.ids,
                         tl.xnext, tl.pos_base);
    if (tid == 0) {
      const unsigned d = __hip_atomic_fetch_add(tl.tk + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d + 1 == (unsigned)N) {                               // every tail has read the full count: re-arm for the next step's launch
        __hip_atomic_store(tl.tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tl.tk + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

template <int RG>
static hipError_t launch_lmhead_f16(const DecoderArgs& a, hipStream_t s) {
  // 128 prefixes (RG = 8): 8 waves = 128 columns per workgroup share an X~ chunk, halving the X~ reads through L2 (77 instead of
  // 154 MB per step); fewer row groups keep 4 waves (two workgroups per CU cover each other's chunk waits)
  constexpr int NWV = RG >= 8 ? PIO_LMF16_WAVES8 : (RG >= 4 ? PIO_LMF16_WAVES4 : 4);
  constexpr int CPW = RG == 8 ? PIO_LMF16_CPW8 : 1;
  const int Vp = round_up(a.vocab, 64);
  const int smem = (PIO_LMF16_DEEP != 0 && RG >= 4 ? 4 : 2) * RG * 16 * 64 * 2;
  const int NGp = round_up(ceil_div(a.vocab, 16), 64);
  static DeviceOnce attr_once; bool& attr_set = attr_once.flag();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_lmhead_f16<RG, NWV, CPW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_lmhead_f16<RG, NWV, CPW>), dim3(ceil_div(a.vocab, 16 * NWV * CPW)), dim3(64 * NWV), smem, s, a.head_w16, (const _Float16*)a.xh, a.N,
                     a.vocab, Vp, a.lm_stats, a.head_d, a.head_c, a.head_w16_unscale, a.logits,
                     a.lm_gmax, NGp);
  return hipGetLastError();
}

static hipError_t launch_lmhead_filtered(const DecoderArgs& a, int step, hipStream_t s) {
  const int Vp = round_up(a.vocab, 64);
  hipError_t e;
  const int rg = ceil_div(a.N, 16);
  if (PIO_LMF16_FUSED && rg <= 1) {
    const int NGp1 = round_up(ceil_div(a.vocab, 16), 64), NG1 = ceil_div(a.vocab, 16), nwg = ceil_div(a.vocab, 64);
    if (NG1 > 4096) return hipErrorInvalidValue;
    // a.lm_tail (PIO_LM_TAIL=1 when the engine is created): the filter as the tail of the head kernel, 21 launches per step instead of 22.
    // Built for VERDICT r4 and MEASURED SLOWER, so it is off: decode(16) 4.285 against 4.15 ms, decode(1) 3.90 against 3.79 (two alternations on one
    // box, ids identical) -- the written-through stores, their drain in every workgroup, the ticket and the agent-scope re-reads cost
    // ~9 us per step where the separate launch costs ~5.8 + its gap.
    if (a.lm_tail && a.splitk_cnt != nullptr && a.N <= nwg) {
      const LmTail tl{a.splitk_cnt + DEC_SPLITK_COUNTERS, a.head_w, a.E, step, a.steps, a.wte, a.wpe, a.ids, a.x, a.pos_base, NG1};
      hipLaunchKernelGGL(k_lmhead_f16_fused<true>, dim3(nwg), dim3(256), 0, s, a.head_w16, a.x, a.N, a.vocab, Vp, a.eps,
                         a.head_bound_coef, a.lm_stats, a.head_d, a.head_c, a.head_w16_unscale, a.logits, a.lm_gmax, NGp1, tl);
      return hipGetLastError();
    }
    hipLaunchKernelGGL(k_lmhead_f16_fused<false>, dim3(nwg), dim3(256), 0, s, a.head_w16, a.x, a.N, a.vocab, Vp, a.eps,
                       a.head_bound_coef, a.lm_stats, a.head_d, a.head_c, a.head_w16_unscale, a.logits, a.lm_gmax, NGp1, LmTail{});
    e = hipGetLastError();
  } else {
    hipLaunchKernelGGL(k_lm_prep, dim3(a.N), dim3(256), 0, s, a.x, a.E, a.eps, a.head_bound_coef, (_Float16*)a.xh, a.lm_stats);
  }
  if (PIO_LMF16_FUSED && rg <= 1) {}
  else if (rg <= 1) e = launch_lmhead_f16<1>(a, s);
  else if (rg <= 2) e = launch_lmhead_f16<2>(a, s);
  else if (rg <= 4) e = launch_lmhead_f16<4>(a, s);
  else if (rg <= 8) e = launch_lmhead_f16<8>(a, s);
  else e = launch_lmhead_f16<16>(a, s);             // 129 .. 256 prefixes: 32-KB X~ chunks, the weights still read once
  if (e != hipSuccess) return e;
  const int NG = ceil_div(a.vocab, 16), NGp = round_up(NG, 64);
  if (NG > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_dec_select_filter, dim3(a.N), dim3(256), 0, s, a.logits, a.lm_gmax, a.vocab, Vp, NG, NGp, a.lm_stats, a.x,
                     a.head_w, a.head_d, a.head_c, a.E, step, a.steps, a.wte, a.wpe, a.ids, a.x, a.pos_base);
  return hipGetLastError();
}

// LM head -> greedy partials: one (max, arg-max, sum-exp) per prefix and 16-column group.
// (A persistent variant that keeps x in registers and walks 5-7 column groups per workgroup measured
//  50 us against 41 us for this one at 16 prefixes: fewer bytes in flight per CU and a serial per-group
//  epilogue; dropped.)
hipError_t launch_lmhead(const float* W, const float* X, int N, int V, int E, const float* dvec, const float* cvec,
                         float eps, float* part, int* nblk, hipStream_t s) {
  *nblk = ceil_div(V, 16);
  if (PIO_LMHEAD_WIDE && E == 768 && N >= 1 && N <= 64) {
    const int rg = ceil_div(N, 16);
    if (rg <= 1) return launch_lmhead_wide<1>(W, X, N, V, dvec, cvec, eps, part, s);
    if (rg <= 2) return launch_lmhead_wide<2>(W, X, N, V, dvec, cvec, eps, part, s);
    return launch_lmhead_wide<4>(W, X, N, V, dvec, cvec, eps, part, s);
  }
  return dec_gemm<DE_ARGMAX, 1>(W, X, N, V, E, dvec, part, nullptr, cvec, eps, nullptr, nullptr, s);
}

// One new position `pos` of every prefix through the transformer layers (x in place; keys / values appended at `pos`).
static hipError_t dec_layers_step(const DecoderArgs& a, int pos, hipStream_t s) {
  const int N = a.N, E = a.E;
  for (int l = 0; l < a.layers; ++l) {
    const DecLayerW& w = a.layer[l];
    float* kc = a.kcache + (size_t)l * N * a.max_steps * E;
    float* vc = a.vcache + (size_t)l * N * a.max_steps * E;
    PIO_TRY((dec_gemm<DE_STORE, 1>(w.attn_w, a.x, N, 3 * E, E, w.attn_d, a.qkv, nullptr, w.attn_c, a.eps, nullptr, nullptr, s, w.attn_ws, w.attn_un)));
    hipLaunchKernelGGL(k_dec_attention, dim3(N * a.heads), dim3(256), 0, s, a.qkv, kc, vc, E, a.heads, pos, a.max_steps, a.att, 1);
    PIO_TRY((dec_gemm<DE_RESID, 0>(w.proj_w, a.att, N, E, E, w.proj_b, a.x, nullptr, nullptr, 0.f, nullptr, nullptr, s)));
    PIO_TRY((dec_gemm<DE_GELU, 1>(w.fc_w, a.x, N, 4 * E, E, w.fc_d, a.hid, nullptr, w.fc_c, a.eps, nullptr, nullptr, s, w.fc_ws, w.fc_un)));
    PIO_TRY((dec_gemm<DE_RESID, 0>(w.fc2_w, a.hid, N, E, 4 * E, w.fc2_b, a.x, nullptr, nullptr, 0.f, a.splitk_ws, a.splitk_cnt, s, w.fc2_ws, w.fc2_un)));
  }
  return hipSuccess;
}

// LM head on x -> ids[.][step] (+ log-prob) and the next position's input x = wte[id] + wpe[pos_base + step + 1]
static hipError_t dec_head_step(const DecoderArgs& a, int step, bool filtered, hipStream_t s) {
  if (filtered) return launch_lmhead_filtered(a, step, s);
  int nblk = 0;
  PIO_TRY(launch_lmhead(a.head_w, a.x, a.N, a.vocab, a.E, a.head_d, a.head_c, a.eps, a.logits, &nblk, s));
  hipLaunchKernelGGL(k_dec_select, dim3(a.N), dim3(256), 0, s, a.logits, nblk, a.N, a.E, step, a.steps, a.wte, a.wpe, a.ids,
                     a.logprob, a.x, a.pos_base);
  return hipGetLastError();
}

static bool dec_args_ok(const DecoderArgs& a, bool filtered, int positions) {
  return positions <= a.max_steps && positions <= 256 && a.steps <= 64 && a.E == 768 && (a.E / a.heads) % 32 == 0 &&
         (a.E / a.heads) <= 256 && a.N <= (filtered ? DEC_MAX_PREFIXES : 64) && ceil_div(a.vocab, 16) <= 4096;
}

hipError_t launch_decode_greedy(const DecoderArgs& a, hipStream_t s) {
  const int N = a.N, E = a.E;
  const bool filtered = PIO_LMHEAD_FILTER && a.logprob == nullptr && a.head_w16 != nullptr;
  // 65..128 prefixes only through the filtered (ids-only) head; the exact head is built for <= 64
  if (!dec_args_ok(a, filtered, a.steps) || a.pos_base != 0) return hipErrorInvalidValue;
  // step 0 input: clip_project(prefix) + wpe[0]   (decap.py:124; GPT-2 adds wpe to inputs_embeds)
  PIO_TRY((dec_gemm<DE_EMBED, 0>(a.clip_w, a.prefix, N, E, a.prefix_size, a.clip_b, a.x, a.wpe, nullptr, 0.f, nullptr, nullptr, s)));
  for (int step = 0; step < a.steps; ++step) {
    PIO_TRY(dec_layers_step(a, step, s));
    PIO_TRY(dec_head_step(a, step, filtered, s));
  }
  return hipGetLastError();
}

// x[n][:] = prompt[n][pos][:] + wpe[pos][:]
__global__ __launch_bounds__(256) void k_dec_prompt_x(const float* __restrict__ prompt, const float* __restrict__ wpe, int P, int pos, int E,
                                                      float* x) {
  const int n = blockIdx.x;
  for (int d = threadIdx.x; d < E; d += 256) x[(size_t)n * E + d] = prompt[((size_t)n * P + pos) * E + d] + wpe[(size_t)pos * E + d];
}

// greedy_search of the ViECap head (P/src/viecap/search.py:108-191) with a KV cache: the prompt embeddings [N][P][E]
// (soft + hard prompt, padded rows included: the reference uses no attention mask) occupy positions 0..P-1; the token
// chosen from the logits of position P-1+k is ids[.][k] and becomes position P+k.  The reference's 64 iterations run
// the prompt forward plus 64 single-token forwards and never use the logits of the last one: P + steps - 1 positions here.
// a.pos_base must be P - 1.
// ---- teacher-forced scoring: GPT2LMHeadModel(input_ids, labels = input_ids).loss as VieCap.compute_perplexity takes it
// (P/src/viecap/entrypoint.py:155-172): the tokens of a finished caption go through the KV-cached layers position by
// position; after position p the exact head gives log sum exp over the vocabulary (the partials of k_lmhead_wide) and one
// more dot product the logit of the NEXT token; nll[n] accumulates their difference for p + 1 < lens[n].
__global__ __launch_bounds__(256) void k_dec_token_x(const int32_t* __restrict__ tokens, int Lmax, int pos, const float* __restrict__ wte,
                                                     const float* __restrict__ wpe, int E, int V, float* x) {
  const int n = blockIdx.x;
  int t = tokens[(size_t)n * Lmax + pos];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  for (int d = threadIdx.x; d < E; d += 256) x[(size_t)n * E + d] = wte[(size_t)t * E + d] + wpe[(size_t)pos * E + d];
}

__global__ __launch_bounds__(256) void k_dec_score(const float* __restrict__ part, int nblk, int N, int E, int V,
                                                   const float* __restrict__ x, const float* __restrict__ head_w,
                                                   const float* __restrict__ head_c, const float* __restrict__ head_d, float eps,
                                                   const int32_t* __restrict__ tokens, const int32_t* __restrict__ lens, int Lmax, int pos,
                                                   float* nll) {
  __shared__ float s_a[4], s_b[4], s_c[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (pos + 1 >= lens[n]) return;                    // block-uniform
  int label = tokens[(size_t)n * Lmax + pos + 1];
  label = label < 0 ? 0 : (label >= V ? V - 1 : label);
  // log sum exp of the row's logits from the head's per-block (max, arg-max, sum of exp relative to the max) partials
  float bv = -INFINITY;
  for (int b = tid; b < nblk; b += 256) bv = fmaxf(bv, part[((size_t)b * N + n) * 4]);
  bv = wave_max(bv);
  if (lane == 0) s_a[wid] = bv;
  __syncthreads();
  bv = fmaxf(fmaxf(s_a[0], s_a[1]), fmaxf(s_a[2], s_a[3]));
  float se = 0.f;
  for (int b = tid; b < nblk; b += 256) {
    const float4 pr = *(const float4*)(part + ((size_t)b * N + n) * 4);
    se += pr.z * expf(pr.x - bv);
  }
  // the label's logit as the head computes it: rstd (W'_v . x - mu c_v) + d_v with the row's LayerNorm statistics
  const float* xr = x + (size_t)n * E;
  const float* wr = head_w + (size_t)label * E;
  float sx = 0.f, sq = 0.f, dot = 0.f;
  for (int d = tid; d < E; d += 256) {
    const float v = xr[d];
    sx += v; sq += v * v; dot += wr[d] * v;
  }
  se = wave_sum(se); sx = wave_sum(sx); sq = wave_sum(sq); dot = wave_sum(dot);
  __syncthreads();
  if (lane == 0) { s_a[wid] = se; s_b[wid] = sx; s_c[wid] = sq; }
  __syncthreads();
  se = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
  sx = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
  sq = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
  __syncthreads();
  if (lane == 0) s_a[wid] = dot;
  __syncthreads();
  if (tid == 0) {
    dot = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
    const float mu = sx / (float)E;
    const float var = fmaxf(sq / (float)E - mu * mu, 0.f);
    const float logit = rsqrtf(var + eps) * (dot - mu * head_c[label]) + head_d[label];
    nll[n] += (bv + logf(se)) - logit;
  }
}

hipError_t launch_lm_score(const DecoderArgs& a, const int32_t* tokens, const int32_t* lens, int Lmax, float* nll, hipStream_t s) {
  if (a.N < 1 || a.N > 64 || Lmax < 1 || Lmax > a.max_steps || !dec_args_ok(a, false, Lmax)) return hipErrorInvalidValue;
  PIO_TRY(hipMemsetAsync(nll, 0, (size_t)a.N * sizeof(float), s));
  for (int pos = 0; pos + 1 < Lmax; ++pos) {
    hipLaunchKernelGGL(k_dec_token_x, dim3(a.N), dim3(256), 0, s, tokens, Lmax, pos, a.wte, a.wpe, a.E, a.vocab, a.x);
    PIO_TRY(dec_layers_step(a, pos, s));
    int nblk = 0;
    PIO_TRY(launch_lmhead(a.head_w, a.x, a.N, a.vocab, a.E, a.head_d, a.head_c, a.eps, a.logits, &nblk, s));
    hipLaunchKernelGGL(k_dec_score, dim3(a.N), dim3(256), 0, s, a.logits, nblk, a.N, a.E, a.vocab, a.x, a.head_w, a.head_c, a.head_d,
                       a.eps, tokens, lens, Lmax, pos, nll);
  }
  return hipGetLastError();
}

// ---- batched prompt prefill (round 5).  The P prompt positions of a prefix do not depend on each other through anything but the key /
// value cache, so the layer GEMMs take all N * P rows at once (128 rows per launch on the exact fp32 kernels: no split-fp16 operands here)
// instead of P passes over N rows: per layer ceil(N P / 128) * 4 + 2 launches instead of 5 P (config 5: 16 prefixes x ~30 prompt
// positions x 12 layers = 1 800 launches of the 5 600 a ViECap caption batch took).  Rows are [n][p]; the attention of position p reads
// the cache, which k_kv_append fills for every position first.  The sums of a GEMM row are those of k_dec_gemm_b instead of k_dec_gemm
// (another order of the same exact fp32 products: rounding-level differences in the cache, ids held to the reference's fixtures by
// tests/test_gpu_viecap.py).
__global__ __launch_bounds__(256) void k_dec_prompt_x_all(const float* __restrict__ prompt, const float* __restrict__ wpe, int P, int E, float* x) {
  const int n = blockIdx.x, p = blockIdx.y;
  const size_t r = (size_t)n * P + p;
  for (int d = threadIdx.x; d < E; d += 256) x[r * E + d] = prompt[r * E + d] + wpe[(size_t)p * E + d];
}
__global__ __launch_bounds__(192) void k_kv_append(const float* __restrict__ qkv, float* kcache, float* vcache, int P, int max_steps, int E) {
  const int n = blockIdx.x, p = blockIdx.y;
  const float4* src = (const float4*)(qkv + ((size_t)n * P + p) * 3 * E);
  float4* kc = (float4*)(kcache + ((size_t)n * max_steps + p) * E);
  float4* vc = (float4*)(vcache + ((size_t)n * max_steps + p) * E);
  for (int d = threadIdx.x; d < E / 4; d += 192) { kc[d] = src[E / 4 + d]; vc[d] = src[E / 2 + d]; }
}
__global__ __launch_bounds__(256) void k_dec_last_rows(const float* __restrict__ xp, int P, int E, float* x) {
  const int n = blockIdx.x;
  for (int d = threadIdx.x; d < E; d += 256) x[(size_t)n * E + d] = xp[((size_t)n * P + P - 1) * E + d];
}
static hipError_t dec_prefill_layers(const DecoderArgs& a, int n0, int gN, int P, hipStream_t s) {
  const int E = a.E, R = gN * P;
  for (int l = 0; l < a.layers; ++l) {
    const DecLayerW& w = a.layer[l];
    float* kc = a.kcache + ((size_t)l * a.N + n0) * a.max_steps * E;
    float* vc = a.vcache + ((size_t)l * a.N + n0) * a.max_steps * E;
    for (int r0 = 0; r0 < R; r0 += 128) {
      const int rows = R - r0 < 128 ? R - r0 : 128;
      PIO_TRY((dec_gemm<DE_STORE, 1>(w.attn_w, a.pre_x + (size_t)r0 * E, rows, 3 * E, E, w.attn_d, a.pre_qkv + (size_t)r0 * 3 * E, nullptr, w.attn_c,
                                     a.eps, nullptr, nullptr, s)));
    }
    hipLaunchKernelGGL(k_kv_append, dim3(gN, P), dim3(192), 0, s, a.pre_qkv, kc, vc, P, a.max_steps, E);
    hipLaunchKernelGGL(k_dec_attention, dim3(gN * a.heads, P), dim3(256), 0, s, a.pre_qkv, kc, vc, E, a.heads, 0, a.max_steps, a.pre_att, P);
    for (int r0 = 0; r0 < R; r0 += 128) {
      const int rows = R - r0 < 128 ? R - r0 : 128;
      PIO_TRY((dec_gemm<DE_RESID, 0>(w.proj_w, a.pre_att + (size_t)r0 * E, rows, E, E, w.proj_b, a.pre_x + (size_t)r0 * E, nullptr, nullptr, 0.f,
                                     nullptr, nullptr, s)));
    }
    for (int r0 = 0; r0 < R; r0 += 128) {
      const int rows = R - r0 < 128 ? R - r0 : 128;
      PIO_TRY((dec_gemm<DE_GELU, 1>(w.fc_w, a.pre_x + (size_t)r0 * E, rows, 4 * E, E, w.fc_d, a.pre_hid + (size_t)r0 * 4 * E, nullptr, w.fc_c, a.eps,
                                    nullptr, nullptr, s)));
    }
    for (int r0 = 0; r0 < R; r0 += 128) {
      const int rows = R - r0 < 128 ? R - r0 : 128;
      PIO_TRY((dec_gemm<DE_RESID, 0>(w.fc2_w, a.pre_hid + (size_t)r0 * 4 * E, rows, E, 4 * E, w.fc2_b, a.pre_x + (size_t)r0 * E, nullptr, nullptr, 0.f,
                                     a.splitk_ws, a.splitk_cnt, s)));
    }
  }
  return hipGetLastError();
}

hipError_t launch_decode_prompted(const DecoderArgs& a, const float* prompt, int P, hipStream_t s) {
  const bool filtered = PIO_LMHEAD_FILTER && a.logprob == nullptr && a.head_w16 != nullptr;
  if (P < 1 || a.steps < 1 || a.pos_base != P - 1 || !dec_args_ok(a, filtered, P + a.steps - 1)) return hipErrorInvalidValue;
  if (a.pre_x != nullptr && P > 1 && P <= a.pre_rows) {
    const int gmax = a.pre_rows / P;                  // prefixes whose prompts fit the prefill workspace at once
    for (int n0 = 0; n0 < a.N; n0 += gmax) {
      const int gN = a.N - n0 < gmax ? a.N - n0 : gmax;
      hipLaunchKernelGGL(k_dec_prompt_x_all, dim3(gN, P), dim3(256), 0, s, prompt + (size_t)n0 * P * a.E, a.wpe, P, a.E, a.pre_x);
      PIO_TRY(dec_prefill_layers(a, n0, gN, P, s));
      hipLaunchKernelGGL(k_dec_last_rows, dim3(gN), dim3(256), 0, s, a.pre_x, P, a.E, a.x + (size_t)n0 * a.E);
    }
  } else {
    for (int pos = 0; pos < P; ++pos) {
      hipLaunchKernelGGL(k_dec_prompt_x, dim3(a.N), dim3(256), 0, s, prompt, a.wpe, P, pos, a.E, a.x);
      PIO_TRY(dec_layers_step(a, pos, s));
    }
  }
  for (int step = 0; step < a.steps; ++step) {
    PIO_TRY(dec_head_step(a, step, filtered, s));            // -> ids[.][step], x = wte[id] + wpe[P + step]
    if (step + 1 < a.steps) PIO_TRY(dec_layers_step(a, P + step, s));
  }
  return hipGetLastError();
}

// ---- beam search building blocks (ViECap: P/src/viecap/search.py:193-285; entrypoint.py:143-148 calls it per image with
// beam_width beams).  The reference re-runs the whole sequence of every beam each step; here the beams keep KV caches that are
// re-gathered by source beam after every selection.  None of this is on the benchmarked path: plain kernels.
// x[n][:] = wte[tokens[n]][:] + wpe[pos][:]
__global__ __launch_bounds__(256) void k_dec_token1_x(const int32_t* __restrict__ tokens, int pos, const float* __restrict__ wte,
                                                      const float* __restrict__ wpe, int E, int V, float* x) {
  const int n = blockIdx.x;
  int t = tokens[n];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  for (int d = threadIdx.x; d < E; d += 256) x[(size_t)n * E + d] = wte[(size_t)t * E + d] + wpe[(size_t)pos * E + d];
}

// dst[l][n][p][:] = src[l][rows ? rows[n] : n][p][:] for p < npos; grid (npos, N, layers)
__global__ __launch_bounds__(192) void k_kv_gather(const float* __restrict__ src, float* __restrict__ dst, const int32_t* __restrict__ rows,
                                                   int N, int max_steps, int E) {
  const int p = blockIdx.x, n = blockIdx.y, l = blockIdx.z;
  int r = rows ? rows[n] : n;
  r = r < 0 ? 0 : (r >= N ? N - 1 : r);
  const float4* a = (const float4*)(src + (((size_t)l * N + r) * max_steps + p) * E);
  float4* b = (float4*)(dst + (((size_t)l * N + n) * max_steps + p) * E);
  for (int d = threadIdx.x; d < E / 4; d += 192) b[d] = a[d];
}

// mean and 1 / sqrt(var + eps) of every row of x (the final LayerNorm the LM head folds in)
__global__ __launch_bounds__(256) void k_row_stats(const float* __restrict__ x, int E, float eps, float* stats) {
  __shared__ float s_a[4], s_b[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  float sx = 0.f, sq = 0.f;
  for (int d = tid; d < E; d += 256) { const float v = x[(size_t)n * E + d]; sx += v; sq += v * v; }
  sx = wave_sum(sx); sq = wave_sum(sq);
  if ((tid & 63) == 0) { s_a[tid >> 6] = sx; s_b[tid >> 6] = sq; }
  __syncthreads();
  if (tid == 0) {
    const float tx = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]), tq = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
    const float mu = tx / (float)E, var = fmaxf(tq / (float)E - mu * mu, 0.f);
    stats[2 * n] = mu;
    stats[2 * n + 1] = rsqrtf(var + eps);
  }
}

// every logit of every row: out[n][v] = rstd_n (<x_n, W'_v> - mu_n c_v) + d_v (the LayerNorm-folded tied head, fp32); one wave per column
__global__ __launch_bounds__(256) void k_lm_logits_full(const float* __restrict__ W, const float* __restrict__ X, int N, int V, int E,
                                                        const float* __restrict__ dvec, const float* __restrict__ cvec,
                                                        const float* __restrict__ stats, float* out) {
  const int lane = threadIdx.x & 63;
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const float* w = W + (size_t)v * E;
  for (int n = 0; n < N; ++n) {
    float acc = 0.f;
    for (int d = lane; d < E; d += 64) acc = fmaf(X[(size_t)n * E + d], w[d], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[(size_t)n * V + v] = stats[2 * n + 1] * (acc - stats[2 * n] * cvec[v]) + dvec[v];
  }
}

// in place: row -> log(softmax(row)) evaluated as the reference does (search.py:246: logits.softmax(-1).log())
__global__ __launch_bounds__(1024) void k_log_softmax_rows(float* x, int V) {
  __shared__ float red[16];
  float* r = x + (size_t)blockIdx.x * V;
  const int tid = threadIdx.x;
  float m = -INFINITY;
  for (int v = tid; v < V; v += 1024) m = fmaxf(m, r[v]);
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  float s = 0.f;
  for (int v = tid; v < V; v += 1024) s += expf(r[v] - m);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = 0.f;
  for (int i = 0; i < 16; ++i) s += red[i];
  for (int v = tid; v < V; v += 1024) r[v] = logf(expf(r[v] - m) / s);
}

// One selection of beam_search (search.py:247-266).  Candidate (b, v):
//   first step (scores == null): lp[0][v], row 0 only (scores, next_tokens = logits.topk(beam_width));
//   later: stopped[b] ? (v == 0 ? scores[b] / lens[b] : -inf) : (scores[b] + lp[b][v]) / (lens[b] + 1)
// (logits[is_stopped] = -inf, logits[is_stopped, 0] = 0; seq_lengths[~is_stopped] += 1; the sum divided by the length).  The W largest,
// descending, ties to the lower flat index b V + v -> out_val[W], out_idx[W].  One workgroup; W <= 8.
__global__ __launch_bounds__(1024) void k_beam_select(const float* __restrict__ lp, const float* __restrict__ scores,
                                                      const float* __restrict__ lens, const int32_t* __restrict__ stopped, int W, int V,
                                                      float* out_val, int64_t* out_idx) {
  __shared__ float s_v[16];
  __shared__ long long s_i[16];
  __shared__ int s_win;
  const int tid = threadIdx.x;
  float bv[8];
  long long bi[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { bv[k] = -INFINITY; bi[k] = -1; }
  const int rows = scores ? W : 1;
  const long long total = (long long)rows * V;
  for (long long c = tid; c < total; c += 1024) {
    const int b = (int)(c / V), v = (int)(c - (long long)b * V);
    float val;
    if (!scores) val = lp[v];
    else if (stopped[b]) val = v == 0 ? scores[b] / lens[b] : -INFINITY;
    else val = (scores[b] + lp[(size_t)b * V + v]) / (lens[b] + 1.0f);
    if (!(val > bv[7]) && bi[7] >= 0) continue;        // not better than this thread's worst kept candidate (NaN never enters)
    if (!(val == val)) continue;
    // insert (val, c) into the thread's descending list; equal values keep the earlier (lower) index first
    int k = 7;
    while (k > 0 && (bi[k - 1] < 0 || val > bv[k - 1])) { bv[k] = bv[k - 1]; bi[k] = bi[k - 1]; --k; }
    bv[k] = val; bi[k] = c;
  }
  int head = 0;
  for (int round = 0; round < W; ++round) {
    float v = head < 8 ? bv[head] : -INFINITY;
    long long i = head < 8 ? bi[head] : -1;
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o);
      const long long oi = __shfl_xor(i, o);
      if (oi >= 0 && (i < 0 || ov > v || (ov == v && oi < i))) { v = ov; i = oi; }
    }
    if ((tid & 63) == 0) { s_v[tid >> 6] = v; s_i[tid >> 6] = i; }
    __syncthreads();
    if (tid == 0) {
      float fv = s_v[0]; long long fi = s_i[0];
      for (int w = 1; w < 16; ++w)
        if (s_i[w] >= 0 && (fi < 0 || s_v[w] > fv || (s_v[w] == fv && s_i[w] < fi))) { fv = s_v[w]; fi = s_i[w]; }
      out_val[round] = fv;
      out_idx[round] = fi;
      s_win = fi >= 0 ? (int)(fi % 1024) : -1;       // the thread that owns candidate fi (c = tid + 1024 j)
    }
    __syncthreads();
    if (tid == s_win) ++head;
    __syncthreads();
  }
}

static hipError_t lm_logp_rows(const DecoderArgs& a, float* stats, float* logp, hipStream_t s) {
  hipLaunchKernelGGL(k_row_stats, dim3(a.N), dim3(256), 0, s, a.x, a.E, a.eps, stats);
  hipLaunchKernelGGL(k_lm_logits_full, dim3(ceil_div(a.vocab, 4)), dim3(256), 0, s, a.head_w, a.x, a.N, a.vocab, a.E, a.head_d, a.head_c,
                     stats, logp);
  hipLaunchKernelGGL(k_log_softmax_rows, dim3(a.N), dim3(1024), 0, s, logp, a.vocab);
  return hipGetLastError();
}

// embeds [N][P][E] at positions 0..P-1 -> logp [N][V] = log softmax of the logits of position P-1
hipError_t launch_lm_prefill(const DecoderArgs& a, const float* embeds, int P, float* stats, float* logp, hipStream_t s) {
  if (P < 1 || a.N < 1 || a.N > 16 || !dec_args_ok(a, false, P)) return hipErrorInvalidValue;
  for (int pos = 0; pos < P; ++pos) {
    hipLaunchKernelGGL(k_dec_prompt_x, dim3(a.N), dim3(256), 0, s, embeds, a.wpe, P, pos, a.E, a.x);
    PIO_TRY(dec_layers_step(a, pos, s));
  }
  return lm_logp_rows(a, stats, logp, s);
}

// beams re-ordered (KV rows of positions < pos gathered by src_rows through kv_scratch), then token[n] at position pos -> logp [N][V]
hipError_t launch_lm_advance(const DecoderArgs& a, const int32_t* tokens, const int32_t* src_rows, int pos, float* kscratch,
                             float* vscratch, float* stats, float* logp, hipStream_t s) {
  if (pos < 1 || a.N < 1 || a.N > 16 || !dec_args_ok(a, false, pos + 1)) return hipErrorInvalidValue;
  if (src_rows) {
    const dim3 g(pos, a.N, a.layers);
    hipLaunchKernelGGL(k_kv_gather, g, dim3(192), 0, s, a.kcache, kscratch, src_rows, a.N, a.max_steps, a.E);
    hipLaunchKernelGGL(k_kv_gather, g, dim3(192), 0, s, a.vcache, vscratch, src_rows, a.N, a.max_steps, a.E);
    hipLaunchKernelGGL(k_kv_gather, g, dim3(192), 0, s, kscratch, a.kcache, (const int32_t*)nullptr, a.N, a.max_steps, a.E);
    hipLaunchKernelGGL(k_kv_gather, g, dim3(192), 0, s, vscratch, a.vcache, (const int32_t*)nullptr, a.N, a.max_steps, a.E);
  }
  hipLaunchKernelGGL(k_dec_token1_x, dim3(a.N), dim3(256), 0, s, tokens, pos, a.wte, a.wpe, a.E, a.vocab, a.x);
  PIO_TRY(dec_layers_step(a, pos, s));
  return lm_logp_rows(a, stats, logp, s);
}

hipError_t launch_beam_select(const float* logp, const float* scores, const float* lens, const int32_t* stopped, int W, int V,
                              float* out_val, int64_t* out_idx, hipStream_t s) {
  if (W < 1 || W > 8 || V < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_beam_select, dim3(1), dim3(1024), 0, s, logp, scores, lens, stopped, W, V, out_val, out_idx);
  return hipGetLastError();
}

}  // namespace pio
