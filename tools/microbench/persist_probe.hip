// What a phase boundary costs INSIDE a persistent kernel on MI355X (diagnostic, never shipped): G = 256 resident workgroups run
// a chain of dependent "phases"; in each, every workgroup reads ALL of a 48-KB vector the previous phase produced (written
// piecewise by all workgroups, i.e. by all 8 XCDs), reduces it and writes its own 48 elements of the next vector -- the data
// dependence of the decoder's skinny GEMMs, without their weights.  The values are integers chosen so that ONE stale read anywhere
// changes the final vector (checked on the host).  Variants of the exchange:
//   atomics : data through 64-bit relaxed agent-scope atomic loads / stores (sc1 accesses, coherent per location), barrier = relaxed
//             atomics on 16 counters (tools/microbench/l2_prefetch_probe.hip: 1.2-1.4 us at any grid size), no fence anywhere
//   fence1  : plain loads / stores, one agent-scope release fence by thread 0 before its arrival, one acquire fence after the wait
//   fenceall: the same fences executed by every wave
// Every spin loop is bounded.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/persist_probe.hip -o tools/microbench/bin/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static constexpr int LIMIT = 50000, NV = 12288;     // NV = 16 x 768 values

// lean grid barrier: workgroup b arrives on counter b & 15 (64 words apart), wave 0 polls all 16 with one load per lane
__device__ __forceinline__ bool grid_sync(unsigned* counters, int G, unsigned gen, int mode, int* s_dead) {
  const int lane = threadIdx.x & 63;
  __syncthreads();
  if (threadIdx.x < 64) {
    if (mode == 1 && lane == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_fetch_add(counters + 64 * (blockIdx.x & 15), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned per = (unsigned)((G >> 4) + ((lane & 15) < (G & 15) ? 1 : 0));
    const unsigned want = gen * per;
    int spins = 0;
    for (;;) {
      const unsigned v = __hip_atomic_load(counters + 64 * (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_ballot_w64(v < want) == 0ull) break;
      if (++spins >= LIMIT) { if (lane == 0) *s_dead = 1; break; }
    }
    if (mode == 1 && lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return *s_dead == 0;
}

template <int MODE>   // 0 atomics, 1 fence by thread 0, 2 fence by every wave
__global__ __launch_bounds__(256) void k_phases(unsigned long long* bufA, unsigned long long* bufB, unsigned* counters, int G, int iters, unsigned* fail) {
  __shared__ int s_dead;
  __shared__ unsigned s_red[4];
  if (threadIdx.x == 0) s_dead = 0;
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long* src = (it & 1) ? bufB : bufA;
    unsigned long long* dst = (it & 1) ? bufA : bufB;
    // every workgroup sums ALL NV values: NV / 2 = 6144 64-bit words, 24 per thread
    unsigned sum = 0;
    unsigned long long v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      if (MODE == 0) v[i] = __hip_atomic_load(src + tid + 256 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v[i] = src[tid + 256 * i];
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) sum += (unsigned)v[i] + (unsigned)(v[i] >> 32);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) s_red[wid] = sum;
    __syncthreads();
    sum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    // this workgroup's 48 values of the next vector: dst[48 b + t] = sum + 48 b + t  (24 words, threads 0..23)
    if (tid < 24) {
      const unsigned j = 48u * blockIdx.x + 2u * tid;
      const unsigned long long w = (unsigned long long)(sum + j) | ((unsigned long long)(sum + j + 1u) << 32);
      if (MODE == 0) __hip_atomic_store(dst + 24 * blockIdx.x + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else dst[24 * blockIdx.x + tid] = w;
    }
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (!grid_sync(counters, G, (unsigned)(it + 1), MODE == 1 ? 1 : 0, &s_dead)) { if (tid == 0) atomicAdd(fail, 1u); return; }
    if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

int main() {
  const int G = 256, iters = 300;
  unsigned long long *A, *B; unsigned *cnt, *fail;
  CK(hipMalloc(&A, NV * 4)); CK(hipMalloc(&B, NV * 4)); CK(hipMalloc(&cnt, 65536)); CK(hipMalloc(&fail, 4));
  std::vector<unsigned> init(NV), exp_v(NV), got(NV);
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < NV; ++i) init[i] = (unsigned)i * 2654435761u;
    // host model
    std::vector<unsigned> cur = init;
    for (int it = 0; it < iters; ++it) {
      unsigned s = 0;
      for (int i = 0; i < NV; ++i) s += cur[i];
      for (int i = 0; i < NV; ++i) cur[i] = s + (unsigned)i;
    }
    CK(hipMemcpy(A, init.data(), NV * 4, hipMemcpyHostToDevice)); CK(hipMemset(B, 0, NV * 4));
    CK(hipMemset(cnt, 0, 65536)); CK(hipMemset(fail, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    if (mode == 0) hipLaunchKernelGGL(k_phases<0>, dim3(G), dim3(256), 0, 0, A, B, cnt, G, iters, fail);
    else if (mode == 1) hipLaunchKernelGGL(k_phases<1>, dim3(G), dim3(256), 0, 0, A, B, cnt, G, iters, fail);
    else hipLaunchKernelGGL(k_phases<2>, dim3(G), dim3(256), 0, 0, A, B, cnt, G, iters, fail);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned f = 0; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(got.data(), (iters & 1) ? B : A, NV * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < NV; ++i) bad += got[i] != cur[i];
    printf("%-8s: %6.2f us per phase (256 workgroups, each reads 48 KB written by all, writes 192 B)   spin-limit hits %u   wrong values %d of %d\n",
           mode == 0 ? "atomics" : mode == 1 ? "fence1" : "fenceall", ms * 1e3f / iters, f, bad, NV);
  }
  return 0;
}
