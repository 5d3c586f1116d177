// ViT multi-head self-attention, flash-style, MFMA on gfx950.
//
// Replaces DINOv2's Attention.forward softmax(q k^T / sqrt(64)) v (reached from P/src/model.py:783)
// without materialising the [B,H,T,T] score tensor.  Inputs are the per-head buffers the QKV GEMM
// epilogue writes: q, k as [B][H][Tk][64] and v TRANSPOSED as [B][H][64][Tk] (Tk = T rounded up to 64,
// zero padded), operand type fp16/bf16.
//
// One workgroup = 4 waves = 128 query rows of one (image, head); each wave owns 32 query rows.  K and
// V^T tiles of 64 keys are register-staged into double-buffered LDS (2 x 16 KiB) shared by the 4 waves.
// Per wave and 64-key tile:
//   S^T[key][q]  = K[key][:] . Q[q][:]        8 x v_mfma_f32_32x32x16 (A = K rows from LDS, B = Q held in
//                                              registers for the whole kernel)
//   online softmax per query: the query is the MFMA column = the lane, so max / sum are 32 in-register
//   ops plus ONE exchange with lane^32 (the other 16 keys of the same query)
//   O^T[dv][q]  += V^T[dv][key] . P^T[key][q]  8 x MFMA; P^T is the S^T accumulator converted in place:
//                                              accumulator registers 8s..8s+7 are exactly the B-operand
//                                              fragment of k-step s (k order permuted; the V^T fragment
//                                              is read with the same permutation), so P never visits LDS.
// The accumulator O^T has the query on the lane as well, so rescaling by exp(m_old - m_new) is lane-local.
#include "common.h"
#include "kernels.h"

namespace pio {

static constexpr int KV_TILE = 64;
static constexpr int KV_TILE_BYTES = KV_TILE * 64 * 2;  // 8 KiB

template <typename T> struct Vec4;
template <> struct Vec4<f16> { typedef _Float16 type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<bf16> { typedef __bf16 type __attribute__((ext_vector_type(4))); };

template <typename T>
__global__ __launch_bounds__(256, 2) void k_vit_attention(const VitAttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // K0 V0 K1 V1
  typedef typename Vec8<T>::type frag_t;
  typedef typename Vec4<T>::type half4_t;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int h = lane >> 5, r31 = lane & 31;
  const int bh = blockIdx.y;                 // b*H + head
  const int b = bh / a.H, head = bh - b * a.H;
  const int nkeys = a.lens ? a.lens[b] : a.T;     // keys of this sequence (block-uniform)
  const int q0 = blockIdx.x * 128 + wid * 32;
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
  int lds_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = row0 + 32 * i;
    lds_off[i] = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
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
    *(uint4*)(_sv + lds_off[0]) = rv0;                                  \
    *(uint4*)(_sv + lds_off[1]) = rv1;                                  \
  } while (0)

  const int sw7 = (lane >> 1) & 7;
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
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    const int ktn = kt + 1 < nkt ? kt + 1 : kt;   // the last iteration reloads its own tile (never stored)
    PIO_LOAD_KV(ktn);
    if (active) {
      const char* sk = smem + buf * 2 * KV_TILE_BYTES;
      const char* sv = sk + KV_TILE_BYTES;
      f32x16 st[2];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kbk][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const frag_t kf = *(const frag_t*)(sk + (kbk * 32 + r31) * 128 + (((2 * s + h) ^ sw7) << 4));
          st[kbk] = mfma32(kf, qf[s], st[kbk]);
        }
      }
      // mask keys >= T, move to the log2 domain, tile max
      float mx = -1e30f;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KV_TILE + kbk * 32 + acc_row32(r, lane);
          const float z = key < nkeys ? st[kbk][r] * sl2 : -1e30f;
          st[kbk][r] = z;
          mx = fmaxf(mx, z);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[kbk][r] - m_new);
          st[kbk][r] = p;
          rs += p;
        }
      rs += __shfl_xor(rs, 32);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
      // O^T += V^T . P^T
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          frag_t pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = (T)st[kbk][8 * s2 + j];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const char* rowp = sv + (d * 32 + r31) * 128 + 8 * h;
            const half4_t lo = *(const half4_t*)(rowp + (((4 * kbk + 2 * s2) ^ sw7) << 4));
            const half4_t hi = *(const half4_t*)(rowp + (((4 * kbk + 2 * s2 + 1) ^ sw7) << 4));
            frag_t vf;
            vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
            vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
            ot[d] = mfma32(vf, pf, ot[d]);
          }
        }
    }
    PIO_STORE_KV(buf ^ 1);   // harmless on the last iteration: that buffer is not read again
    __syncthreads();
  }
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
  dim3 grid(ceil_div(a.Tp, 128), a.B * a.H);
  if (t == OP_F16) hipLaunchKernelGGL((k_vit_attention<f16>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_vit_attention<bf16>), grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace pio
