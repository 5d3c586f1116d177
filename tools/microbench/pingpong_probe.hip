// Feasibility probe for a ping-pong ViT GEMM (round 3): ONE wave per SIMD multiplies a 128 x 256 x 64 K-tile per slot from a
// 3-stage LDS ring (48 KiB per stage) while the other four waves of the workgroup only issue the LDS-DMA that refills it.
// Question: how many cycles does a K-tile take (1024 = the matrix pipe's own time for 32 MFMAs of 32x32x16)?  No epilogue, no
// output (the accumulators are kept alive by a dummy store).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include
//   -I patchioner_amd/csrc tools/microbench/pingpong_probe.hip -o tools/microbench/bin/pingpong_probe
#include "../../patchioner_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstring>
using namespace pio;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static constexpr int TK = 64, STAGE = 48 * 1024, NST = 3;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef _Float16 frag_t __attribute__((ext_vector_type(8)));

#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// stage: A [128 rows][128 B] at 0, B [256 rows][128 B] at 16 KiB; 16-B chunk c of row r sits at chunk c ^ ((r >> 1) & 7)
template <int MODE>   // 0: waves 4-7 issue all DMA; 1: every wave issues half of it (waves 0-3 between their MFMAs)
__global__ __launch_bounds__(512, 2) void k_probe(const _Float16* A, const _Float16* W, int M, int N, int K, float* sink, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, w4 = wid & 3, h = lane >> 5, r31 = lane & 31;
  const int ntn = N / 256, ntiles = ((M + 127) / 128) * ntn, nk = K / TK;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((size_t)M * K * 2), 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((size_t)N * K * 2), 0x00020000);
  const int sw7 = (lane >> 1) & 7;
  int co[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) co[s] = ((2 * s + h) ^ sw7) << 4;
  // DMA piece p (1 KiB = 8 rows): lane -> row 8 p + (lane >> 3), chunk (lane & 7) ^ swz(row)
  const int prow = lane >> 3;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  long nkt = 0;
  // the stream of (tile, K-tile) pairs this workgroup walks
  int stream_len = 0;
  for (int id = xcd_remap(blockIdx.x, gridDim.x); id < ntiles; id += gridDim.x) stream_len += nk;
  auto tile_of = [&](int q) { return xcd_remap(blockIdx.x, gridDim.x) + (q / nk) * (int)gridDim.x; };
#define ISSUE(q)                                                                                           \
  do {                                                                                                     \
    const int _id = tile_of(q), _kt = (q) % nk, _tm = _id / ntn, _tn = _id - _tm * ntn;                    \
    char* const _st = smem + ((q) % NST) * STAGE;                                                          \
    const int _np = MODE == 0 ? 12 : 6, _p0 = MODE == 0 ? w4 * 12 : wid * 6;                               \
    _Pragma("unroll") for (int _i = 0; _i < 12; ++_i) {                                                    \
      if (_i >= _np) break;                                                                                \
      const int _p = _p0 + _i;              /* 0..15: A pieces, 16..47: B pieces */                        \
      if (_p < 16) {                                                                                       \
        int _r = _tm * 128 + 8 * _p + prow; _r = _r < M ? _r : M - 1;                                      \
        const uint32_t _off = (uint32_t)_r * (uint32_t)(K * 2) + (uint32_t)((((lane & 7) ^ (((8 * _p + prow) >> 1) & 7))) << 4); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(_st + _p * 1024), 16, _off, _kt * 128, 0, 0); \
      } else {                                                                                             \
        const int _pb = _p - 16;                                                                           \
        const int _r = _tn * 256 + 8 * _pb + prow;                                                         \
        const uint32_t _off = (uint32_t)_r * (uint32_t)(K * 2) + (uint32_t)((((lane & 7) ^ (((8 * _pb + prow) >> 1) & 7))) << 4); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(_st + 16384 + _pb * 1024), 16, _off, _kt * 128, 0, 0); \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)
  if (MODE == 1 || grp == 1) { ISSUE(0); if (stream_len > 1) ISSUE(1); }
  if (MODE == 1 || grp == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MODE == 0 ? 12 : 6) : "memory");
  BAR();
  for (int q = 0; q < stream_len; ++q) {
    const char* st = smem + (q % NST) * STAGE;
    if (grp == 0) {
      frag_t fa[2][4], fb[2][2];
      const int a_rd = r31 * 128, b_rd = 16384 + (w4 * 64 + r31) * 128;
#define READ(buf, s)                                                                                       \
      do {                                                                                                 \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) fa[buf][rt] = *(const frag_t*)(st + a_rd + rt * 4096 + co[s]); \
        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) fb[buf][ct] = *(const frag_t*)(st + b_rd + ct * 4096 + co[s]); \
      } while (0)
#define MMA(buf)                                                                                           \
      do {                                                                                                 \
        _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)  \
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[buf][ct], fa[buf][rt], acc[rt][ct], 0, 0, 0); \
      } while (0)
      READ(0, 0);
      READ(1, 1);
      MMA(0);
      if (MODE == 1 && q + 2 < stream_len) ISSUE(q + 2);
      READ(0, 2);
      MMA(1);
      READ(1, 3);
      MMA(0);
      MMA(1);
      if (MODE == 1) { if (q + 2 < stream_len) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#undef READ
#undef MMA
    } else {
      if (q + 2 < stream_len) { ISSUE(q + 2); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MODE == 0 ? 12 : 6) : "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BAR();
    ++nkt;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0 && stamps) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = (unsigned long long)nkt; }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
  if (s == 12345.678f) sink[tid] = s;
}

static uint16_t f2h(float f) { _Float16 x = (_Float16)f; uint16_t u; memcpy(&u, &x, 2); return u; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 80;
  const int M = B * 264, K = 768;
  for (int N : {2304, 3072, 768}) {
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : ha) v = f2h(nd(rng));
    for (auto& v : hw) v = f2h(nd(rng) * 0.03f);
    void *A, *W; float* sink; unsigned long long* st;
    CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&W, hw.size() * 2)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&st, 256 * 16));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; ++mode) {
      auto kern = mode == 0 ? k_probe<0> : k_probe<1>;
      CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), NST * STAGE, 0, (const _Float16*)A, (const _Float16*)W, M, N, K, sink, st);
      CK(hipEventRecord(e0, 0));
      const int it = 10;
      for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), NST * STAGE, 0, (const _Float16*)A, (const _Float16*)W, M, N, K, sink, st);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> hs(512);
      CK(hipMemcpy(hs.data(), st, 512 * 8, hipMemcpyDeviceToHost));
      double cyc = 0, kts = 0;
      for (int w = 0; w < 256; ++w) { cyc += (double)hs[2 * w]; kts += (double)hs[2 * w + 1]; }
      const double us = ms * 1e3 / it, fl = 2.0 * M * (double)N * K;
      printf("B=%d N=%4d mode %d: %.1f us (%.0f TF, main loop only), %.0f cycles per K-tile (128x256x64 per CU; 1024 = MFMA-bound)\n", B, N, mode, us, fl / us / 1e6, cyc / kts);
    }
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(sink)); CK(hipFree(st));
  }
  return 0;
}
