// ViT multi-head self-attention, flash-style, MFMA on gfx950.
//
// Replaces DINOv2's Attention.forward softmax(q k^T / sqrt(64)) v (reached from P/src/model.py:783)
// without materialising the [B,H,T,T] score tensor.  Inputs are the per-head buffers the QKV GEMM
// epilogue writes: q, k as [B][H][Tk][64] and v TRANSPOSED as [B][H][64][Tk] (Tk = T rounded up to 64,
// zero padded), operand type fp16/bf16.
//
// One wave owns 32 query rows of one (image, head).  K and V^T tiles of 64 keys live in double-buffered LDS
// (2 x 16 KiB) shared by the workgroup's waves.  Per wave and 64-key tile:
//   S^T[key][q]  = K[key][:] . Q[q][:]        8 x v_mfma_f32_32x32x16 (A = K rows from LDS, B = Q held in
//                                              registers for the whole kernel)
//   online softmax per query: the query is the MFMA column = the lane, so max / sum are 32 in-register
//   ops plus ONE exchange with lane^32 (the other 16 keys of the same query)
//   O^T[dv][q]  += V^T[dv][key] . P^T[key][q]  8 x MFMA; P^T is the S^T accumulator converted in place:
//                                              accumulator registers 8s..8s+7 are exactly the B-operand
//                                              fragment of k-step s (k order permuted; the V^T fragment
//                                              is read with the same permutation), so P never visits LDS.
// The accumulator O^T has the query on the lane as well, so rescaling by exp(m_old - m_new) is lane-local.
//
// Round 2.  What moved the kernel was its VALU count (~250 VALU instructions per 16 MFMAs in round 1).  k_vit_attention's
// softmax is now: padding keys masked only in a sequence's last tile, the scale folded into the exponent's FMA (maximum over
// raw scores), packed conversions, no rescale of O^T while no maximum moved; and a V^T row sits in LDS in the order the P^T
// fragment multiplies it (one ds_read_b128 per fragment, no regrouping v_movs).  64 images: 46.1 us per launch (290 TF)
// against round 1's 51.3; 16 images 18.4 against 20.5; 518^2 x 8: 86 us (537 TF) against 105.
// Round 3.  Round 2 had concluded that another wave's VALU never runs under an MFMA; its probe's fmaf loop had been packed by
// hipcc.  tools/microbench/mfma_valu_overlap2.hip (opcodes pinned): plain v_fma_f32 / v_add_f32 / v_cvt_pk DO overlap another
// wave's MFMAs, v_pk_fma_f32 / v_pk_add_f32 do NOT.  The FMAs and adds of the softmax are therefore plain instructions
// (PIO_ATTN_PLAIN_VALU): 43.0 us at 64 images against 44.8 packed, same bits.
// k_vit_attention2 (removed from the tree in round 4; git history: tools/microbench/attic/vit_attention2.hip) was a larger restructuring built before that was understood; correct (same tests), not
// faster, kept for its measurements:
//   * workgroup = ceil(nq / ceil(nq / 8)) waves (T = 261: 9 query tiles = two workgroups of 5 waves instead of three of 4
//     whose third is 6 % full); K / V^T by LDS-DMA through a ring of 2-4 tiles with counted vmcnt (no staging registers, no
//     ds_write); the same lean softmax, all-padding half tiles skipped.
//   * 64 images: 56 / 58 / 61 us with a ring of 2 / 3 / 4 tiles.  An XCD-aware block order (both kernels, attn_block below)
//     changed nothing.  PMC at 64 images (rocprofv3, tools/microbench/attn_run.py, profiles/r02_attention_pmc.json): MFMA
//     busy 11 % of the SIMD cycles, VALU ~35 %, LDS index-active 15 % of which 38 % bank conflicts, waves waiting (s_waitcnt /
//     barrier) 50 % of their resident cycles.  Every wave runs read K -> 8 MFMA -> max / exp chain (two lane exchanges) ->
//     read V -> 8 MFMA serially and meets 3-4 others at a barrier per 64 keys; more waves per SIMD do not help (128 VGPRs
//     spill: 72 us).
// A third form was built and removed again (round 2, same tests green): one PERSISTENT workgroup per CU, the whole K / V^T of
// a head resident in LDS (T <= 320: 80 KiB) and double-buffered across heads (2 x 80 KiB), one wave per query tile, ONE
// barrier per head, no load and no barrier in the key loop.  Its ablations at 64 images say where the time is: loading
// Q / K / V^T 16.5 us, the key loop 32 us, storing O 7-9 us; un-overlapped (one head per workgroup) they add up to 55 us,
// overlapped (persistent) to 48.1 us -- exactly this kernel's 47.9.  The key loop is ISSUE-bound: ~1 700 cycles per (32
// queries x 64 keys) = 512 of MFMA + ~200 plain VALU instructions (80 of them v_mov: hipcc pairs the two 8-byte V^T pieces of
// two rows in one ds_read2st64_b64 and then regroups them) that do not run under MFMAs; 120 MB of Q / K / V / O per launch
// put the memory floor at ~24 us.  What would move it: V^T stored in LDS in the order the P^T fragment needs (one b128 read,
// no moves), S^T of tile t+1 issued under the softmax of tile t, fewer VALU instructions still.
#include "common.h"
#include "kernels.h"

namespace pio {

#ifndef PIO_ATTN_PLAIN_VALU   // 1: the softmax's FMAs / adds as plain VALU instructions (see the kernel)
#define PIO_ATTN_PLAIN_VALU 1
#endif
#ifndef PIO_ATTN_VSWZ_PARITY  // 1: the V^T staging image's swizzle includes the row's parity: no LDS bank conflict is left in the kernel
#define PIO_ATTN_VSWZ_PARITY 0   // (rocprofv3: SQ_LDS_BANK_CONFLICT 0 against 0.178 of the LDS cycles) -- and it is 3-5 % SLOWER (42.4 against 41.1 us
#endif                           // at 64 images, 50.5 against 48.0 at 80, twice each on one box; profiles/r05_attention_pmc.json): the conflicts
                                 // were two-way ones on four 8-byte staging stores per thread and tile, which cost a store nothing until its
                                 // LDS-array cycles exceed its issue cycles (MI355X_MICROARCH.md).  Off.  Also measured in round 5 and not
                                 // kept: workgroups of FIVE waves for T = 261 (9 query tiles = 2 x 5 instead of 3 x 4 waves, a third fewer
                                 // workgroups staging K / V^T; same bits): 21.8 against 17.3 us at 16 images, 62.7 against 41.3 at 64.
#ifndef PIO_ATTN_OCC          // waves per SIMD k_vit_attention is compiled for: 2 (144 VGPRs, three workgroups per CU in practice);
#define PIO_ATTN_OCC 2         // 4 = 128 VGPRs with 56 B of scratch, measured: see the file header
#endif
static constexpr int KV_TILE = 64;
static constexpr int KV_TILE_BYTES = KV_TILE * 64 * 2;  // 8 KiB

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <typename T> struct Vec2;
template <> struct Vec2<f16> { typedef _Float16 type __attribute__((ext_vector_type(2))); };
template <> struct Vec2<bf16> { typedef __bf16 type __attribute__((ext_vector_type(2))); };
template <typename T> struct Vec4;
template <> struct Vec4<f16> { typedef _Float16 type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<bf16> { typedef __bf16 type __attribute__((ext_vector_type(4))); };

// XCD-aware block order.  The query blocks of one (image, head) all read that head's K and V^T (82 KB at T = 261), which
// the QKV GEMM of 64 images has just written as 94 MB of q / k / v -- far more than the 8 x 4 MB of L2, so they come from
// the Infinity Cache / HBM.  Workgroup ids are dealt round-robin over the 8 XCDs: with a (query block, head) grid the blocks
// of a head land on DIFFERENT XCDs and each fetches K / V^T again (round 1: 3 x 63 MB per launch, and 52 us at 64 images
// is exactly that traffic at 4.7 TB/s).  Here consecutive slots of ONE XCD take the query blocks of one head, so the second
// and third read hit that XCD's L2.  1-D grid of nblk * ceil(BH / 8) * 8 ids; ids whose head does not exist leave.
__device__ __forceinline__ bool attn_block(int nblk, int BH, int& qblk, int& bh) {
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  qblk = slot % nblk;
  bh = (slot / nblk) * 8 + xcd;
  return bh < BH;
}

template <typename T>
__global__ __launch_bounds__(256, PIO_ATTN_OCC) void k_vit_attention(const VitAttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // K0 V0 K1 V1
  typedef typename Vec8<T>::type frag_t;
  typedef typename Vec4<T>::type half4_t;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, r31 = lane & 31;
  int qblk, bh;                              // bh = b*H + head
  if (!attn_block((a.Tp + 127) / 128, a.B * a.H, qblk, bh)) return;
  const int b = bh / a.H, head = bh - b * a.H;
  const int nkeys = a.lens ? a.lens[b] : a.T;     // keys of this sequence (block-uniform)
  const int q0 = qblk * 128 + wid * 32;
  const bool active = q0 < a.Tp;             // wave-uniform
  const T* qb = (const T*)a.q + (size_t)bh * a.Tk * 64;
  const T* kb = (const T*)a.k + (size_t)bh * a.Tk * 64;
  const T* vb = (const T*)a.vT + (size_t)bh * 64 * a.Tk;

  // Q fragments (B operand of S^T): lane (q = r31, h) holds Q[q][16s + 8h + j]
  frag_t qf[4];
  {
    int qr = q0 + r31;
    qr = qr < a.Tk ? qr : a.Tk - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const frag_t*)(qb + (size_t)qr * 64 + 16 * s + 8 * h);
  }

  // staging assignment: 2 chunks of K and 2 of V^T per thread per tile
  const int kc = tid & 7, row0 = tid >> 3;   // rows row0, row0+32
  // K rows keep their 16-B chunks in place (chunk ^ swizzle(row)).  A V^T row is stored in the order the P^T fragment wants
  // it: lane-half h of k-step g (16 keys) multiplies keys {16g + 4h .. +3, 16g + 8 + 4h .. +3} -- the low (h = 0) or high 8
  // bytes of chunks 2g and 2g + 1 -- so slot 2g + h holds [that half of chunk 2g | that half of chunk 2g + 1] and the
  // fragment is ONE ds_read_b128 (it was two 8-byte pieces that hipcc fetched pairwise across rows and regrouped with
  // v_movs, which are MFMA time on gfx950).  Costs two ds_write_b64 instead of one b128 per staged chunk.
  int lds_off[2], lds_off_v[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = row0 + 32 * i, swz = (row >> 1) & 7;
    lds_off[i] = row * 128 + ((kc ^ swz) << 4);
    // (round 5, PIO_ATTN_VSWZ_PARITY) ... optionally XORed with the row's parity as well: a ds_write_b64 is served in groups of 16 lanes
    // = two neighbouring rows, which share (row >> 1) & 7 and -- 128 B apart -- the same banks: both write slots {j, 2 + j, 4 + j, 6 + j}
    // and every such store takes two passes (SQ_LDS_BANK_CONFLICT 0.18 of the LDS cycles since round 2: four of these stores per thread
    // and tile against 16 fragment reads per wave).  With the parity the odd row writes the other four slots and the counter reads 0; the
    // fragment reads stay conflict-free (a read group's even and odd rows sit in different bank halves, and inside each half the extra XOR
    // is the same for every lane).  It buys no time (see the switch above).
    const int swv = swz ^ (PIO_ATTN_VSWZ_PARITY ? (row & 1) : 0);
    lds_off_v[i][0] = row * 128 + ((((kc & ~1) + 0) ^ swv) << 4) + (kc & 1) * 8;
    lds_off_v[i][1] = row * 128 + ((((kc & ~1) + 1) ^ swv) << 4) + (kc & 1) * 8;
  }
  // staging registers: plain named values (arrays captured by lambdas end up in scratch)
  uint4 rk0, rk1, rv0, rv1;
  const T* k_src0 = kb + (size_t)row0 * 64 + kc * 8;
  const T* k_src1 = kb + (size_t)(row0 + 32) * 64 + kc * 8;
  const T* v_src0 = vb + (size_t)row0 * a.Tk + kc * 8;
  const T* v_src1 = vb + (size_t)(row0 + 32) * a.Tk + kc * 8;
#define PIO_LOAD_KV(kt)                                                 \
  do {                                                                  \
    rk0 = *(const uint4*)(k_src0 + (size_t)(kt) * KV_TILE * 64);        \
    rk1 = *(const uint4*)(k_src1 + (size_t)(kt) * KV_TILE * 64);        \
    rv0 = *(const uint4*)(v_src0 + (kt) * KV_TILE);                     \
    rv1 = *(const uint4*)(v_src1 + (kt) * KV_TILE);                     \
  } while (0)
#define PIO_STORE_KV(buf)                                               \
  do {                                                                  \
    char* _sk = smem + (buf) * 2 * KV_TILE_BYTES;                       \
    char* _sv = _sk + KV_TILE_BYTES;                                    \
    *(uint4*)(_sk + lds_off[0]) = rk0;                                  \
    *(uint4*)(_sk + lds_off[1]) = rk1;                                  \
    *(uint2*)(_sv + lds_off_v[0][0]) = make_uint2(rv0.x, rv0.y);        \
    *(uint2*)(_sv + lds_off_v[0][1]) = make_uint2(rv0.z, rv0.w);        \
    *(uint2*)(_sv + lds_off_v[1][0]) = make_uint2(rv1.x, rv1.y);        \
    *(uint2*)(_sv + lds_off_v[1][1]) = make_uint2(rv1.z, rv1.w);        \
  } while (0)

  const int sw7 = (lane >> 1) & 7;
  const int sw7v = sw7 ^ (PIO_ATTN_VSWZ_PARITY ? (lane & 1) : 0);              // V^T rows: the staging's swizzle includes the row's parity (lds_off_v)
  const float sl2 = a.scale * 1.44269504088896340736f;  // softmax in the log2 domain
  float m_run = -1e30f, l_run = 0.f;
  f32x16 ot[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;

  const int nkt = nkeys > 0 ? (nkeys + KV_TILE - 1) / KV_TILE : 1;
  PIO_LOAD_KV(0);
  PIO_STORE_KV(0);
  __syncthreads();
  typedef typename Vec2<T>::type half2_t;
  // One 64-key tile.  NKB = key blocks of 32 that hold keys of the sequence (2, or 1 in a last tile with <= 32 keys left), MASKED =
  // the tile holds padding (only a sequence's last tile does).  Round 4: the tile with padding is a code path of its own, entered by
  // a scalar branch (`left` is block-uniform), so full tiles carry no mask code and a short last tile -- T = 261 leaves 5 keys, 518^2
  // leaves 30 -- multiplies 32 keys instead of 64 for S^T and only the 16-key steps that hold keys for O^T.  The skipped products
  // are exact zeros of the long form (masked scores give p = 0; padded V^T rows are zero), so the bits do not change.  Same live
  // registers as before (round 3's skip inside the one loop body cost 4 VGPRs and with them a wave per SIMD).
#define PIO_ATTN_TILE(NKB, MASKED, NS2)                                                                                   \
  do {                                                                                                                \
    f32x16 st[NKB];                                                                                                   \
    _Pragma("unroll") for (int kbk = 0; kbk < NKB; ++kbk) {                                                           \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) st[kbk][r] = 0.f;                                                \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                 \
        const frag_t kf = *(const frag_t*)(sk + (kbk * 32 + r31) * 128 + (((2 * s + h) ^ sw7) << 4));                 \
        st[kbk] = mfma32(kf, qf[s], st[kbk]);                                                                         \
      }                                                                                                               \
    }                                                                                                                 \
    if (MASKED) {                                                                                                     \
      _Pragma("unroll") for (int kbk = 0; kbk < NKB; ++kbk)                                                           \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                \
          if (kbk * 32 + acc_row32(r, lane) >= left) st[kbk][r] = -1e30f;                                             \
    }                                                                                                                 \
    float mx = st[0][0];                                                                                              \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[0][r]);                                          \
    if (NKB == 2) { _Pragma("unroll") for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[NKB - 1][r]); }                  \
    mx = xor32_max(mx);                                /* v_permlane32_swap: no LDS crossbar round trip on the chain */ \
    const float m_new = fmaxf(m_run, mx * sl2);                                                                       \
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {                                                           \
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                                                      \
      l_run *= alpha;                                                                                                 \
      const f32x2 av = {alpha, alpha};                                                                                \
      _Pragma("unroll") for (int d = 0; d < 2; ++d)                                                                   \
        _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                                           \
          f32x2 o = {ot[d][r], ot[d][r + 1]};                                                                         \
          o *= av;                                                                                                    \
          ot[d][r] = o[0]; ot[d][r + 1] = o[1];                                                                       \
        }                                                                                                             \
      m_run = m_new;                                                                                                  \
    }                                                                                                                 \
    /* plain (un-packed) fp32 VALU: v_pk_fma_f32 / v_pk_add_f32 do not run beside another wave's MFMAs on gfx950, plain */ \
    /* v_fma_f32 / v_add_f32 do (tools/microbench/mfma_valu_overlap2.hip), and three waves share a SIMD here */       \
    float rs_a = 0.f, rs_b = 0.f;                                                                                     \
    half2_t ph[NKB][8];                                                                                               \
    const float nm = -m_run;                                                                                          \
    _Pragma("unroll") for (int kbk = 0; kbk < NKB; ++kbk)                                                             \
      _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                                             \
        float z0, z1;                                                                                                 \
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(z0) : "v"(st[kbk][r]), "v"(sl2), "v"(nm));                              \
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(z1) : "v"(st[kbk][r + 1]), "v"(sl2), "v"(nm));                          \
        f32x2 p;                                                                                                      \
        p[0] = __builtin_amdgcn_exp2f(z0);                                                                            \
        p[1] = __builtin_amdgcn_exp2f(z1);                                                                            \
        /* s_nop: the operand is a transcendental's result (v_exp_f32), which needs one wait state before a VALU reads it; hipcc */ \
        /* inserts it for its own instructions, not for inline asm (round 4: found as run-to-run differences once it scheduled */ \
        /* the add right behind the exponential) */                                                                     \
        asm("s_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(rs_a) : "v"(p[0]));                                               \
        asm("s_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(rs_b) : "v"(p[1]));                                               \
        ph[kbk][r >> 1] = __builtin_convertvector(p, half2_t);                                                        \
      }                                                                                                               \
    float rs = rs_a + rs_b;                                                                                           \
    rs = xor32_add(rs);                                                                                               \
    l_run += rs;                                                                                                      \
    /* O^T += V^T . P^T */                                                                                            \
    _Pragma("unroll") for (int kbk = 0; kbk < NKB; ++kbk) {                                                           \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                              \
        if (s2 < (NS2)) {                                                                                             \
          frag_t pf;                                                                                                  \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) { pf[2 * j] = ph[kbk][4 * s2 + j][0]; pf[2 * j + 1] = ph[kbk][4 * s2 + j][1]; } \
          _Pragma("unroll") for (int d = 0; d < 2; ++d) {                                                             \
            const frag_t vf = *(const frag_t*)(sv + (d * 32 + r31) * 128 + (((4 * kbk + 2 * s2 + h) ^ sw7v) << 4));   \
            ot[d] = mfma32(vf, pf, ot[d]);                                                                            \
          }                                                                                                           \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)
  // every tile but the last is full: no padding, no mask
  for (int kt = 0; kt + 1 < nkt; ++kt) {
    const int buf = kt & 1;
    constexpr int left = KV_TILE;
    PIO_LOAD_KV(kt + 1);
    if (active) {
      const char* sk = smem + buf * 2 * KV_TILE_BYTES;
      const char* sv = sk + KV_TILE_BYTES;
      PIO_ATTN_TILE(2, false, 2);
    }
    PIO_STORE_KV(buf ^ 1);
    __syncthreads();
  }
  if (active) {
    const int kt = nkt - 1;
    const char* sk = smem + (kt & 1) * 2 * KV_TILE_BYTES;
    const char* sv = sk + KV_TILE_BYTES;
    const int left = nkeys - kt * KV_TILE;              // keys of the sequence in the last tile (block-uniform; <= 0: an empty sequence)
    if (left >= KV_TILE) PIO_ATTN_TILE(2, false, 2);
    else if (left > 32) PIO_ATTN_TILE(2, true, 2);
    else {
      const int ns2 = left > 16 ? 2 : 1;                  // 16-key steps of O^T that hold keys
      PIO_ATTN_TILE(1, true, ns2);
    }
  }
#undef PIO_ATTN_TILE
#undef PIO_LOAD_KV
#undef PIO_STORE_KV

  if (active) {
    const int q = q0 + r31;
    if (q < a.Tp) {
      const float inv = 1.0f / l_run;
      T* orow = (T*)a.out + (size_t)(b * a.Tp + q) * a.D + head * 64;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          half4_t o4;
#pragma unroll
          for (int j = 0; j < 4; ++j) o4[j] = (T)(ot[d][4 * g4 + j] * inv);
          *(half4_t*)(orow + d * 32 + 8 * g4 + 4 * h) = o4;
        }
    }
  }
}

hipError_t launch_vit_attention(OperandType t, const VitAttnArgs& a, hipStream_t s) {
  if (a.D != a.H * 64 || a.Tk % KV_TILE != 0 || a.Tk < a.Tp || a.Tp < a.T) return hipErrorInvalidValue;
  dim3 grid(ceil_div(a.Tp, 128) * ceil_div(a.B * a.H, 8) * 8);
  if (t == OP_F16) hipLaunchKernelGGL((k_vit_attention<f16>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_vit_attention<bf16>), grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace pio
