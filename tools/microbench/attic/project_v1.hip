// k_project: round 1's memory-projection kernel (3 barriers per tile, register-staged single buffer).  k_project2 (round 2) does
// every operation in the same order and was held bit-identical to it on the GPU for a round (5 bank shapes x 6 query counts)
// before this kernel left the product in round 3.  It compiled inside patchioner_amd/csrc/project.hip and was selected by
// PIO_PROJECT_V1=1.
// NQG = query groups of 16 per pass: 1 (256 threads, two workgroups per CU) or 2 (512 threads, one per CU: waves 0-3
// take queries q0..q0+15, waves 4-7 the next 16, both on the SAME bank tile in LDS, so a pass over the bank serves 32
// queries -- the pipeline projects two image batches per pass).
// RT = 16-row tiles per loop iteration: 2 for the 32-query form (one workgroup per CU has no second workgroup to fill its
// barriers, so it walks two tiles between them: GEMM1 of both, ONE reduction barrier, then soft-max + GEMM2 of each in
// row order -- the same operations in the same order per tile, bit-identical to RT = 1).
template <int D, int NQG, int RT>
__global__ __launch_bounds__(256 * NQG, NQG == 1 ? 2 : 1) void k_project(const float* __restrict__ bank, const float* __restrict__ inv_norm,
                                                    int64_t M, const float* __restrict__ q, int N, int q0,
                                                    float temperature, float* part_acc, float* part_ml, int parts) {
  constexpr int STRIDE = D + 4;                  // floats; +16 B skews rows across the 64 banks
  constexpr int DW = D / 4;                      // channels per wave
  constexpr int NT = 256 * NQG;                  // threads
  constexpr int ROWS = PR_ROWS * RT;             // bank rows staged per iteration
  constexpr int NV = D / 4 * ROWS / NT;          // float4 per thread per iteration
  constexpr int NQ = PR_Q * NQG;                 // queries per pass
  static_assert(D % 64 == 0 && (D / 4 * ROWS) % NT == 0, "D");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* s_bank = lds;                            // [ROWS][STRIDE]  (single buffer: two workgroups share a CU)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = (tid >> 6) & 3, grp = tid >> 8;  // channel slice of the wave, query group of the wave
  float* s_red = lds + ROWS * STRIDE + grp * RT * 4 * 256;   // [NQG][RT][4][256]
  const int li = lane & 15, kq = lane >> 4;

  // slab of rows for this workgroup (multiple of 16 rows)
  const int64_t tiles_total = (M + ROWS - 1) / ROWS;
  const int64_t tiles_per = (tiles_total + parts - 1) / parts;
  const int64_t t_begin = (int64_t)blockIdx.x * tiles_per;
  int64_t t_end = t_begin + tiles_per;
  if (t_end > tiles_total) t_end = tiles_total;

  // this wave's slice of the 16 queries stays in registers for the whole slab (zero rows for n >= N):
  // lane (li = query, kq) holds q[q0+li][wid*DW + 16c + 4kq .. +3]
  float4 qreg[DW / 16];
#pragma unroll
  for (int c = 0; c < DW / 16; ++c) {
    qreg[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + 16 * grp + li < N)
      qreg[c] = *(const float4*)(q + (size_t)(q0 + 16 * grp + li) * D + wid * DW + 16 * c + 4 * kq);
  }

  // Per-thread staging of the next tile: NV <= 12 float4 held in NAMED registers (hipcc keeps a staged
  // array that is written in one loop and read in another in scratch).
  static_assert(NV <= 12, "NV");
  float4 g0, g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11;
  // non-temporal: the bank is read once per call and is 7x the Infinity Cache; a default-policy stream evicts the
  // ViT / decoder weights that the next kernels want to find there
#define PIO_BANK_SRC(t, i) \
  ld_stream4(bank + (((t) * ROWS + (tid + NT * (i)) / (D / 4)) < M ? ((t) * ROWS + (tid + NT * (i)) / (D / 4)) : M - 1) * D + 4 * ((tid + NT * (i)) % (D / 4)))
#define PIO_BANK_DST(buf, i) \
  (*(float4*)(s_bank + ((tid + NT * (i)) / (D / 4)) * STRIDE + 4 * ((tid + NT * (i)) % (D / 4))))
#define PIO_LOAD_BANK(t)                                  \
  do {                                                    \
    if constexpr (NV > 0) g0 = PIO_BANK_SRC(t, 0);        \
    if constexpr (NV > 1) g1 = PIO_BANK_SRC(t, 1);        \
    if constexpr (NV > 2) g2 = PIO_BANK_SRC(t, 2);        \
    if constexpr (NV > 3) g3 = PIO_BANK_SRC(t, 3);        \
    if constexpr (NV > 4) g4 = PIO_BANK_SRC(t, 4);        \
    if constexpr (NV > 5) g5 = PIO_BANK_SRC(t, 5);        \
    if constexpr (NV > 6) g6 = PIO_BANK_SRC(t, 6);        \
    if constexpr (NV > 7) g7 = PIO_BANK_SRC(t, 7);        \
    if constexpr (NV > 8) g8 = PIO_BANK_SRC(t, 8);        \
    if constexpr (NV > 9) g9 = PIO_BANK_SRC(t, 9);        \
    if constexpr (NV > 10) g10 = PIO_BANK_SRC(t, 10);     \
    if constexpr (NV > 11) g11 = PIO_BANK_SRC(t, 11);     \
  } while (0);
#define PIO_STORE_BANK(buf)                               \
  do {                                                    \
    if constexpr (NV > 0) PIO_BANK_DST(buf, 0) = g0;      \
    if constexpr (NV > 1) PIO_BANK_DST(buf, 1) = g1;      \
    if constexpr (NV > 2) PIO_BANK_DST(buf, 2) = g2;      \
    if constexpr (NV > 3) PIO_BANK_DST(buf, 3) = g3;      \
    if constexpr (NV > 4) PIO_BANK_DST(buf, 4) = g4;      \
    if constexpr (NV > 5) PIO_BANK_DST(buf, 5) = g5;      \
    if constexpr (NV > 6) PIO_BANK_DST(buf, 6) = g6;      \
    if constexpr (NV > 7) PIO_BANK_DST(buf, 7) = g7;      \
    if constexpr (NV > 8) PIO_BANK_DST(buf, 8) = g8;      \
    if constexpr (NV > 9) PIO_BANK_DST(buf, 9) = g9;      \
    if constexpr (NV > 10) PIO_BANK_DST(buf, 10) = g10;   \
    if constexpr (NV > 11) PIO_BANK_DST(buf, 11) = g11;   \
  } while (0);

  f32x4 acc[DW / 16];
#pragma unroll
  for (int j = 0; j < DW / 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;          // for query n = li (replicated over kq and over waves)

  if (t_begin < t_end) {
    PIO_LOAD_BANK(t_begin)
    PIO_STORE_BANK(0)
  }
  __syncthreads();
  for (int64_t t = t_begin; t < t_end; ++t) {
    const int64_t tn = t + 1 < t_end ? t + 1 : t;   // last iteration: reloads its own tile (stored, never read)
    PIO_LOAD_BANK(tn)
    // ---- GEMM1: partial S[row][n] over this wave's channels, every 16-row tile of the iteration ----
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const float* sb = s_bank + r * PR_ROWS * STRIDE;
      f32x4 sp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < DW / 16; ++c) {
        const int d = wid * DW + 16 * c + 4 * kq;
        const float4 a = *(const float4*)(sb + li * STRIDE + d);
        const float4 b = qreg[c];
        sp = mfma16(a.x, b.x, sp);
        sp = mfma16(a.y, b.y, sp);
        sp = mfma16(a.z, b.z, sp);
        sp = mfma16(a.w, b.w, sp);
      }
      *(f32x4*)(s_red + (r * 4 + wid) * 256 + lane * 4) = sp;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const float* sb = s_bank + r * PR_ROWS * STRIDE;
      f32x4 sfull = *(const f32x4*)(s_red + (r * 4) * 256 + lane * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4 o = *(const f32x4*)(s_red + (r * 4 + w) * 256 + lane * 4);
        sfull += o;
      }
      // ---- online softmax for query n = li; this lane's rows are 4*kq + i ----
      float p[4];
      float tmax = -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = t * ROWS + r * PR_ROWS + 4 * kq + i;
        float z = -INFINITY;
        if (row < M) z = (sfull[i] * inv_norm[row]) / temperature;
        p[i] = z;
        tmax = fmaxf(tmax, z);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run, tmax);       // finite from the first tile on (a slab's first tile has a valid row)
      const float alpha = expf(m_run - m_new);      // exp(-inf) = 0 on the first tile
      float rs = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = expf(p[i] - m_new);
        rs += p[i];
      }
      rs += __shfl_xor(rs, 16);
      rs += __shfl_xor(rs, 32);
      l_run = l_run * alpha + rs;
      m_run = m_new;
      // Acc[n][d]: this lane's register i belongs to query n = 4*kq + i -> fetch that query's alpha
      float al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) al[i] = __shfl(alpha, 4 * kq + i);
      // ---- GEMM2: Acc[n][d] = Acc*alpha + P^T . bank ----
#pragma unroll
      for (int j = 0; j < DW / 16; ++j) {
        f32x4 a4 = acc[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] *= al[i];
        const float* bp = sb + (4 * kq) * STRIDE + wid * DW + 16 * j + li;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) a4 = mfma16(p[tt], bp[tt * STRIDE], a4);
        acc[j] = a4;
      }
    }
    __syncthreads();                 // every wave is done reading this tile
    PIO_STORE_BANK(0)
    __syncthreads();
  }
#undef PIO_LOAD_BANK
#undef PIO_STORE_BANK
#undef PIO_BANK_SRC
#undef PIO_BANK_DST

  // ---- partial results: part_acc[block][n][D], part_ml[block][n][2] ----
  float* pa = part_acc + ((size_t)blockIdx.x * NQ + 16 * grp) * D;
#pragma unroll
  for (int j = 0; j < DW / 16; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) pa[(size_t)(4 * kq + i) * D + wid * DW + 16 * j + li] = acc[j][i];
  if (wid == 0 && kq == 0) {
    part_ml[((size_t)blockIdx.x * NQ + 16 * grp + li) * 2 + 0] = m_run;
    part_ml[((size_t)blockIdx.x * NQ + 16 * grp + li) * 2 + 1] = l_run;
  }
}

