// k_vit_attention2: the LDS-DMA-ring / 5-wave restructuring of the ViT attention measured in round 2 (not faster: 48.1 vs 46.1 us
// at 64 images; profiles/r02_attention_pmc.json) and removed from the product in round 3.  Kept for reference: it compiled inside
// patchioner_amd/csrc/vit_attention.hip (namespace pio, after k_vit_attention) and was selected by PIO_ATTN_V2=1.



// blockDim.x = 64 * (waves per workgroup, 1..8); grid = (query blocks, B * H).
// K / V^T tiles stream through a ring of ATT_RING tiles (16 KiB each): the prologue puts ATT_RING tiles in flight, the
// iteration of tile kt first waits (counted vmcnt) for tile kt, meets the other waves at ONE barrier -- which also says
// that everyone has left tile kt-1 -- and refills that slot with tile kt-1+ATT_RING before it computes.  Round 1's two
// buffers had one tile of look-ahead, about 0.7 us of work per tile against a 1-2 us load: the loop ran at the latency.
// The 16 wave-operations of a tile are issued by the first 1, 2 or 4 waves (the largest of those <= the workgroup's
// waves), 16 / 8 / 4 each, so every issuing wave has the same number in flight per tile and one immediate serves it.
#ifndef PIO_ATT_RING
#define PIO_ATT_RING 3
#endif
static constexpr int ATT_RING = PIO_ATT_RING;
static constexpr int ATT_MAX_WAVES = 8;

template <int N> __device__ __forceinline__ void att_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most `tiles` (0..ATT_RING-2) tiles of `ops` (4, 8, 16) operations each may still be in flight
__device__ __forceinline__ void att_wait_tiles(int ops, int tiles) {
  const int n = ops * tiles;                     // wave-uniform
  if (n >= 32) att_wait_vm<32>();
  else if (n >= 16) att_wait_vm<16>();
  else if (n >= 8) att_wait_vm<8>();
  else if (n >= 4) att_wait_vm<4>();
  else att_wait_vm<0>();
}

template <typename T>
__global__ __launch_bounds__(512, 4) void k_vit_attention2(const VitAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];            // ATT_RING x (K tile, V^T tile)
  typedef typename Vec8<T>::type frag_t;
  typedef typename Vec4<T>::type half4_t;
  typedef typename Vec2<T>::type half2_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int h = lane >> 5, r31 = lane & 31;
  int qblk, bh;                              // bh = b*H + head
  if (!attn_block(((a.Tp + 31) / 32 + ATT_MAX_WAVES - 1) / ATT_MAX_WAVES, a.B * a.H, qblk, bh)) return;   // as the launcher counts them
  const int b = bh / a.H, head = bh - b * a.H;
  const int nkeys = a.lens ? a.lens[b] : a.T;     // keys of this sequence (block-uniform)
  const int q0 = (qblk * nw + wid) * 32;
  const bool active = q0 < a.Tp;             // wave-uniform
  const T* qb = (const T*)a.q + (size_t)bh * a.Tk * 64;
  const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)a.k + (size_t)bh * a.Tk * 64), 0, a.Tk * 64 * 2, 0x00020000);
  const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)a.vT + (size_t)bh * 64 * a.Tk), 0, a.Tk * 64 * 2, 0x00020000);

  // LDS-DMA staging: a tile pair is 16 wave-operations of 1 KiB (8 rows x 128 B, lane-linear in LDS): operations 0..7 the K
  // tile, 8..15 the V^T tile.  Lane l fills slot (l & 7) of row 8 j + (l >> 3) and so fetches source chunk
  // (l & 7) ^ ((row >> 1) & 7) -- the layout the fragment reads below expect.
  const int ni = nw >= 4 ? 4 : nw >= 2 ? 2 : 1;   // issuing waves
  const int ops = 16 / ni;                        // operations per issuing wave and tile
  const bool issuer = wid < ni;
  const int r8 = lane >> 3, slot = lane & 7;
#define PIO_STAGE_KV(kt)                                                                                              \
  do {                                                                                                                \
    if (issuer) {                                                                                                     \
      char* _base = smem + ((kt) % ATT_RING) * 2 * KV_TILE_BYTES;                                                     \
      for (int _op = wid; _op < 16; _op += ni) {                                                                      \
        const int _j = _op & 7, _row = 8 * _j + r8, _chunk = slot ^ ((_row >> 1) & 7);                                \
        char* _d = _base + _op * 1024;                                                                                \
        if (_op < 8)                                                                                                  \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr_t)_d, 16, (_row * 64 + _chunk * 8) * 2, (kt) * (KV_TILE * 128), 0, 0); \
        else                                                                                                          \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr_t)_d, 16, (_row * a.Tk + _chunk * 8) * 2, (kt) * (KV_TILE * 2), 0, 0); \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)

  const int sw7 = (lane >> 1) & 7;
  const float sl2 = a.scale * 1.44269504088896340736f;  // softmax in the log2 domain
  const f32x2 sl2v = {sl2, sl2};
  float m_run = -1e30f, l_run = 0.f;                    // m_run in the scaled log2 domain
  f32x16 ot[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;

  const int nkt = nkeys > 0 ? (nkeys + KV_TILE - 1) / KV_TILE : 1;
  for (int t = 0; t < ATT_RING - 1 && t < nkt; ++t) PIO_STAGE_KV(t);
  // Q fragments (B operand of S^T): lane (q = r31, h) holds Q[q][16s + 8h + j].  Loaded AFTER the ring's first tiles and
  // pinned here: hipcc otherwise waits for them at their first use INSIDE the loop with vmcnt(3..0) -- a drain of the ring
  // in every iteration.
  frag_t qf[4];
  {
    int qr = q0 + r31;
    qr = qr < a.Tk ? qr : a.Tk - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const frag_t*)(qb + (size_t)qr * 64 + 16 * s + 8 * h);
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
  }
  for (int kt = 0; kt < nkt; ++kt) {
    // tiles issued so far: 0 .. min(nkt - 1, kt + ATT_RING - 2); tile kt has to have landed
    const int ahead = nkt - 1 - kt < ATT_RING - 2 ? nkt - 1 - kt : ATT_RING - 2;
    if (issuer) att_wait_tiles(ops, ahead);
    // a raw barrier: __syncthreads() carries a release fence that hipcc turns into vmcnt(0) -- the whole ring drained per tile.
    // Nothing but LDS-DMA writes and ds_reads touches this LDS, and the counted wait above is what orders them.
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                        // tile kt is in LDS for everyone; everyone has left tile kt-1
    __builtin_amdgcn_sched_barrier(0);
    if (kt + ATT_RING - 1 < nkt) PIO_STAGE_KV(kt + ATT_RING - 1);   // into the slot of tile kt-1
    if (active) {
      const char* sk = smem + (kt % ATT_RING) * 2 * KV_TILE_BYTES;
      const char* sv = sk + KV_TILE_BYTES;
      const int left = nkeys - kt * KV_TILE;              // keys of the sequence in this tile and after (block-uniform)
      const bool two = left > 32;                         // the tile's upper 32 keys hold at least one real key
      f32x16 st[2];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
        const bool live = kbk == 0 || two;               // wave-uniform; a dead half scores -1e30: exp2 -> 0, no MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kbk][r] = live ? 0.f : -1e30f;
        if (live) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const frag_t kf = *(const frag_t*)(sk + (kbk * 32 + r31) * 128 + (((2 * s + h) ^ sw7) << 4));
            st[kbk] = mfma32(kf, qf[s], st[kbk]);
          }
        }
      }
      if (left < KV_TILE) {                               // the sequence ends inside this tile: mask the padding keys
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kbk * 32 + acc_row32(r, lane) >= left) st[kbk][r] = -1e30f;
      }
      float mx = st[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[1][r]);
      mx = xor32_max(mx);                                // v_permlane32_swap: no LDS crossbar round trip on the chain
      const float m_new = fmaxf(m_run, mx * sl2);
      if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {   // some query's running maximum moved: rescale
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
        const f32x2 av = {alpha, alpha};
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 o = {ot[d][r], ot[d][r + 1]};
            o *= av;
            ot[d][r] = o[0]; ot[d][r + 1] = o[1];
          }
        m_run = m_new;
      }
      const f32x2 mv = {m_run, m_run};
      f32x2 rs2 = {0.f, 0.f};
      half2_t ph[2][8];
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 sv2 = {st[kbk][r], st[kbk][r + 1]};
          const f32x2 z = __builtin_elementwise_fma(sv2, sl2v, -mv);
          f32x2 p;
          p[0] = __builtin_amdgcn_exp2f(z[0]);
          p[1] = __builtin_amdgcn_exp2f(z[1]);
          rs2 += p;
          ph[kbk][r >> 1] = __builtin_convertvector(p, half2_t);
        }
      float rs = rs2[0] + rs2[1];
      rs = xor32_add(rs);
      l_run += rs;
      // O^T += V^T . P^T
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
        if (kbk == 1 && !two) break;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          frag_t pf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { pf[2 * j] = ph[kbk][4 * s2 + j][0]; pf[2 * j + 1] = ph[kbk][4 * s2 + j][1]; }
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
    }
  }
#undef PIO_STAGE_KV

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
