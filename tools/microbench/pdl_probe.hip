// Can the decoder's chain of small dependent kernels overlap a kernel's dispatch, start-up and weight loads with its predecessor's
// execution (what CUDA calls programmatic dependent launch)?  Diagnostic, never shipped.
//   A "stage" has the data dependence of one decoder GEMM at <= 16 prefixes: G workgroups each stream their own 48 KB of weights,
//   read ALL of the 48-KB activation vector the previous stage produced (pieces written by all of its workgroups, i.e. by all 8 XCDs)
//   and write their piece of the next vector.  Integer arithmetic chosen so that ONE stale read anywhere changes the final vector
//   (checked against a host model).  Forms of a chain of `n` stages:
//     chain    : one stream, one hipGraph: every stage waits for its predecessor at the kernel boundary (what the decoder does)
//     pdl-sc1  : TWO linear graphs on two streams, even stages in one, odd stages in the other -- stage k+1 is dispatched while stage k
//                runs, issues its weight loads, then waits for stage k's arrival counter (relaxed agent-scope atomics, one poller per
//                workgroup, bounded) and reads the vector with sc1 loads; stage k publishes with sc1 stores + counter
//     pdl-fence: the same with plain loads / stores and a release fence before the arrival, an acquire fence after the wait
//   In flight at any time: stage k (runnable) and stage k+1 (waiting); k+2 is behind k in the same queue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/pdl_probe.hip -o tools/microbench/bin/pdl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static constexpr int LIMIT = 20000, NV = 12288;     // NV = 16 x 768 values
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0 chain, 1 pdl-sc1, 2 pdl-fence
template <int MODE>
__global__ __launch_bounds__(256) void k_stage(const u32x4* __restrict__ W, const u64* src, u64* dst, const unsigned* flag_in, unsigned need,
                                               unsigned* flag_out, unsigned* fail) {
  __shared__ unsigned s_red[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int G = gridDim.x;
  // the weight stream: 12 x 16 B per lane, all in flight, independent of the predecessor
  u32x4 w[12];
  const u32x4* wp = W + (size_t)blockIdx.x * 3072 + tid;
#pragma unroll
  for (int c = 0; c < 12; ++c) w[c] = __builtin_nontemporal_load(wp + 256 * c);
  __builtin_amdgcn_sched_barrier(0);
  if (MODE != 0) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && ++spins < LIMIT) __builtin_amdgcn_s_sleep(1);
      if (spins >= LIMIT) atomicAdd(fail, 1u);
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  unsigned sum = 0;
  u64 v[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    if (MODE == 1) v[i] = __hip_atomic_load(src + tid + 256 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else v[i] = src[tid + 256 * i];
  }
#pragma unroll
  for (int i = 0; i < 24; ++i) sum += (unsigned)v[i] + (unsigned)(v[i] >> 32);
  unsigned ws = 0;
#pragma unroll
  for (int c = 0; c < 12; ++c) ws += (w[c].x ^ w[c].y) + (w[c].z ^ w[c].w);
  sum += ws;                       // weights are all-zero words: the sum is unchanged, but the loads are needed
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) s_red[wid] = sum;
  __syncthreads();
  sum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  // this workgroup's NV / G values of the next vector: dst[j] = sum + j
  const int per = NV / G, words = per / 2;
  if (tid < words) {
    const unsigned j = (unsigned)per * blockIdx.x + 2u * tid;
    const u64 o = (u64)(sum + j) | ((u64)(sum + j + 1u) << 32);
    if (MODE == 1) __hip_atomic_store(dst + (size_t)words * blockIdx.x + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else dst[(size_t)words * blockIdx.x + tid] = o;
  }
  if (MODE != 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(flag_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int MODE>
static hipError_t launch(hipStream_t s, int G, const u32x4* W, const u64* src, u64* dst, const unsigned* fin, unsigned need, unsigned* fout, unsigned* fail) {
  hipLaunchKernelGGL(k_stage<MODE>, dim3(G), dim3(256), 0, s, W, src, dst, fin, need, fout, fail);
  return hipGetLastError();
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 600;       // stages per chain
  const int grids[3] = {48, 96, 192};
  const int NW = 8;                                    // distinct weight sets, rotated (8 x 192 x 48 KB = 75 MB: beyond L2)
  u32x4* W; u64 *A, *B; unsigned *flags, *fail;
  CK(hipMalloc(&W, (size_t)NW * 192 * 49152)); CK(hipMemset(W, 0, (size_t)NW * 192 * 49152));
  CK(hipMalloc(&A, NV * 4)); CK(hipMalloc(&B, NV * 4)); CK(hipMalloc(&flags, (n + 1) * 64 * 4)); CK(hipMalloc(&fail, 4));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  std::vector<unsigned> init(NV), got(NV);
  for (int i = 0; i < NV; ++i) init[i] = (unsigned)i * 2654435761u;
  std::vector<unsigned> cur = init;
  for (int it = 0; it < n; ++it) {
    unsigned s = 0;
    for (int i = 0; i < NV; ++i) s += cur[i];
    for (int i = 0; i < NV; ++i) cur[i] = s + (unsigned)i;
  }
  for (int gi = 0; gi < 3; ++gi) {
    const int G = grids[gi];
    for (int mode = 0; mode < 3; ++mode) {
      // build the graphs: flag k (64 words apart) counts the arrivals of stage k; stage k waits for flag k-1 == G (stage 0: flag n, pre-set)
      hipGraph_t g[2] = {nullptr, nullptr}; hipGraphExec_t ge[2] = {nullptr, nullptr};
      const int nchains = mode == 0 ? 1 : 2;
      for (int c = 0; c < nchains; ++c) {
        hipStream_t s = c == 0 ? s1 : s2;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = c; k < n; k += nchains) {
          const u64* src = (k & 1) ? B : A; u64* dst = (k & 1) ? A : B;
          const u32x4* w = W + (size_t)(k % NW) * 192 * 3072;
          const unsigned* fin = flags + 64 * (k == 0 ? n : k - 1); unsigned* fout = flags + 64 * k;
          if (mode == 0) CK(launch<0>(s, G, w, src, dst, fin, (unsigned)G, fout, fail));
          else if (mode == 1) CK(launch<1>(s, G, w, src, dst, fin, (unsigned)G, fout, fail));
          else CK(launch<2>(s, G, w, src, dst, fin, (unsigned)G, fout, fail));
        }
        CK(hipStreamEndCapture(s, &g[c]));
        CK(hipGraphInstantiate(&ge[c], g[c], nullptr, nullptr, 0));
      }
      float best = 1e30f; unsigned f = 0; int bad = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpyAsync(A, init.data(), NV * 4, hipMemcpyHostToDevice, s1)); CK(hipMemsetAsync(B, 0, NV * 4, s1));
        CK(hipMemsetAsync(flags, 0, (n + 1) * 64 * 4, s1)); CK(hipMemsetAsync(fail, 0, 4, s1));
        const unsigned gg = (unsigned)G;
        CK(hipMemcpyAsync(flags + 64 * n, &gg, 4, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        CK(hipEventRecord(e0, s1));
        if (nchains == 2) { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); }
        CK(hipGraphLaunch(ge[0], s1));
        if (nchains == 2) { CK(hipGraphLaunch(ge[1], s2)); CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); }
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), (n & 1) ? B : A, NV * 4, hipMemcpyDeviceToHost));
        bad = 0;
        for (int i = 0; i < NV; ++i) bad += got[i] != cur[i];
        if (f || bad) break;
      }
      printf("G=%3d %-9s: %6.2f us per stage (%d stages; each workgroup streams 48 KB of weights, reads the 48-KB vector written by all)   "
             "spin-limit hits %u   wrong values %d of %d\n", G, mode == 0 ? "chain" : mode == 1 ? "pdl-sc1" : "pdl-fence", best * 1e3f / n, n, f, bad, NV);
      for (int c = 0; c < nchains; ++c) { CK(hipGraphExecDestroy(ge[c])); CK(hipGraphDestroy(g[c])); }
    }
  }
  return 0;
}
