// Two questions behind the round-4 decoder work (diagnostic, never shipped):
//  (1) does a line fetched into an XCD's L2 by kernel A survive the kernel boundary, i.e. can kernel A warm L2 with kernel B's
//      weights?  Chain  [sweep 48 MB (evicts the 8 x 4 MB of L2, not the 256 MB Infinity Cache)] [P: prefetch] [B: consumer]
//      with P touching one dword per 128-B line of exactly the 48-KB chunk the SAME-numbered workgroup of B reads (same XCD by the
//      round-robin dispatch), or the chunk of workgroup + 1 (another XCD), or nothing.
//  (2) what an in-kernel grid barrier costs when it is written without acquire / release fences in the poll loop (relaxed agent-scope
//      atomics only; grid_barrier.hip's numbers, 2.4 .. 11 us, were measured with an L2 invalidate per poll).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/l2_prefetch_probe.hip -o tools/microbench/bin/l2_prefetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static constexpr int LIMIT = 50000;      // polls; a workgroup that hits it leaves the kernel (so does every other one, at its own limit)

__global__ __launch_bounds__(256) void k_sweep(const float4* __restrict__ p, size_t n4, float* sink) {
  float a = 0.f;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
  if (a == 123.456f) *sink = a;
}

// consumer: workgroup b streams its own 48-KB chunk (16 rows x 768 fp32), 12 x 16 B per lane all in flight, like k_dec_gemm
__global__ __launch_bounds__(256) void k_consume(const float* __restrict__ W, float* out) {
  const float* wp = W + (size_t)blockIdx.x * 12288 + threadIdx.x * 4;
  float4 w[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) w[c] = *(const float4*)(wp + 1024 * c);
  float a = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c) a += (w[c].x + w[c].y) + (w[c].z + w[c].w);
  out[blockIdx.x * 256 + threadIdx.x] = a;
}

// prefetch: workgroup b touches one dword per 128-B line of the chunk of consumer workgroup (b + shift) % G
__global__ __launch_bounds__(256) void k_prefetch(const float* __restrict__ W, int G, int shift, float* sink) {
  const int c = (blockIdx.x + shift) % G;
  const float* wp = W + (size_t)c * 12288;
  unsigned acc = 0;
  for (int line = threadIdx.x; line < 384; line += 256) acc |= __float_as_uint(wp[line * 32]);
  if (acc == 0x7fc12345u) *sink = 1.f;
}

// the same prefetch issued from INSIDE a consumer-like kernel that also does its own work (own chunk from W1, prefetch from W2)
__global__ __launch_bounds__(256) void k_consume_pf(const float* __restrict__ W1, const float* __restrict__ W2, float* out) {
  const float* wp = W1 + (size_t)blockIdx.x * 12288 + threadIdx.x * 4;
  float4 w[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) w[c] = *(const float4*)(wp + 1024 * c);
  __builtin_amdgcn_sched_barrier(0);
  const float* pp = W2 + (size_t)blockIdx.x * 12288;
  unsigned p0 = __float_as_uint(pp[threadIdx.x * 32]);
  unsigned p1 = threadIdx.x < 128 ? __float_as_uint(pp[(256 + threadIdx.x) * 32]) : 0u;
  __builtin_amdgcn_sched_barrier(0);
  float a = 0.f;
#pragma unroll
  for (int c = 0; c < 12; ++c) a += (w[c].x + w[c].y) + (w[c].z + w[c].w);
  out[blockIdx.x * 256 + threadIdx.x] = a;
  if ((p0 | p1) == 0x7fc12345u) out[0] = 1.f;
}

// ---- lean barriers: relaxed agent-scope atomics, no fence in the loop ----
__global__ __launch_bounds__(256) void k_bar_flat(unsigned* counter, int G, int iters, unsigned* fail) {
  __shared__ int s_dead;
  if (threadIdx.x == 0) s_dead = 0;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(it + 1) * (unsigned)G;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < LIMIT) {}
      if (spins >= LIMIT) { atomicAdd(fail, 1u); s_dead = 1; }
    }
    __syncthreads();
    if (s_dead) return;
  }
}
// 16 counters on separate 256-B lines (workgroup b arrives on counter b & 15); wave 0 polls all 16 with one load
__global__ __launch_bounds__(256) void k_bar_16(unsigned* counters, int G, int iters, unsigned* fail) {
  const int lane = threadIdx.x & 63;
  __shared__ int s_dead;
  if (threadIdx.x == 0) s_dead = 0;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x < 64) {
      if (lane == 0) __hip_atomic_fetch_add(counters + 64 * (blockIdx.x & 15), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned per = (unsigned)((G >> 4) + ((lane & 15) < (G & 15) ? 1 : 0));
      const unsigned want = (unsigned)(it + 1) * per;
      int spins = 0;
      for (;;) {
        const unsigned v = __hip_atomic_load(counters + 64 * (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64(v < want) == 0ull) break;
        if (++spins >= LIMIT) { if (lane == 0) { atomicAdd(fail, 1u); s_dead = 1; } break; }
      }
    }
    __syncthreads();
    if (s_dead) return;
  }
}
// flag array: workgroup b stores the generation into its own word (plain agent-scope store), wave 0 reads all G words
__global__ __launch_bounds__(256) void k_bar_flags(unsigned* flags, int G, int iters, unsigned* fail) {
  const int lane = threadIdx.x & 63;
  __shared__ int s_dead;
  if (threadIdx.x == 0) s_dead = 0;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x < 64) {
      if (lane == 0) __hip_atomic_store(flags + blockIdx.x, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      for (;;) {
        bool ok = true;
        for (int i = lane; i < G; i += 64) ok = ok && (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(it + 1));
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        if (++spins >= LIMIT) { if (lane == 0) { atomicAdd(fail, 1u); s_dead = 1; } break; }
      }
    }
    __syncthreads();
    if (s_dead) return;
  }
}

template <typename F>
static float chain_us(F f, int iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iter; ++i) f();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e3f / iter;
}

int main() {
  const int G = 192;
  const size_t wbytes = (size_t)G * 12288 * 4, sweep_bytes = (size_t)48 << 20, big_bytes = (size_t)768 << 20;
  float *W, *W2, *S, *out, *sink;
  CK(hipMalloc(&W, wbytes)); CK(hipMalloc(&W2, wbytes)); CK(hipMalloc(&S, big_bytes)); CK(hipMalloc(&out, (size_t)G * 256 * 4 + 64)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(W, 0, wbytes)); CK(hipMemset(W2, 0, wbytes)); CK(hipMemset(S, 0, big_bytes));
  auto sweep = [&](size_t bytes) { hipLaunchKernelGGL(k_sweep, dim3(1024), dim3(256), 0, 0, (const float4*)S, bytes / 16, sink); };
  auto consume = [&] { hipLaunchKernelGGL(k_consume, dim3(G), dim3(256), 0, 0, W, out); };
  const int IT = 200;
  for (int rep = 0; rep < 2; ++rep) {
    const float t_sweep = chain_us([&] { sweep(sweep_bytes); }, IT);
    const float t_cold = chain_us([&] { sweep(sweep_bytes); consume(); }, IT);
    const float t_pf = chain_us([&] { sweep(sweep_bytes); hipLaunchKernelGGL(k_prefetch, dim3(G), dim3(256), 0, 0, W, G, 0, sink); }, IT);
    const float t_pf_b = chain_us([&] { sweep(sweep_bytes); hipLaunchKernelGGL(k_prefetch, dim3(G), dim3(256), 0, 0, W, G, 0, sink); consume(); }, IT);
    const float t_pfx_b = chain_us([&] { sweep(sweep_bytes); hipLaunchKernelGGL(k_prefetch, dim3(G), dim3(256), 0, 0, W, G, 1, sink); consume(); }, IT);
    const float t_bb = chain_us([&] { consume(); }, 1000);
    const float t_big = chain_us([&] { sweep(big_bytes); }, 10);
    const float t_big_b = chain_us([&] { sweep(big_bytes); consume(); }, 10);
    // prefetch from inside a working kernel: A reads W2 (its own chunk) and prefetches W's chunk; then B = consume(W)
    const float t_a = chain_us([&] { sweep(sweep_bytes); hipLaunchKernelGGL(k_consume, dim3(G), dim3(256), 0, 0, W2, out); }, IT);
    const float t_a_b = chain_us([&] { sweep(sweep_bytes); hipLaunchKernelGGL(k_consume, dim3(G), dim3(256), 0, 0, W2, out); consume(); }, IT);
    const float t_apf = chain_us([&] { sweep(sweep_bytes); hipLaunchKernelGGL(k_consume_pf, dim3(G), dim3(256), 0, 0, W2, W, out); }, IT);
    const float t_apf_b = chain_us([&] { sweep(sweep_bytes); hipLaunchKernelGGL(k_consume_pf, dim3(G), dim3(256), 0, 0, W2, W, out); consume(); }, IT);
    printf("rep %d: consumer B (192 wg x 48 KB = 9.4 MB):\n", rep);
    printf("  back to back (everything warm)              %6.2f us\n", t_bb);
    printf("  after a 48-MB sweep (L2 cold, MALL warm)     %6.2f us   (sweep alone %.2f)\n", t_cold - t_sweep, t_sweep);
    printf("  after a 768-MB sweep (MALL cold too)         %6.2f us   (sweep alone %.2f)\n", t_big_b - t_big, t_big);
    printf("  sweep, prefetch kernel (same wg -> same XCD) %6.2f us   (prefetch kernel itself %.2f)\n", t_pf_b - t_pf, t_pf - t_sweep);
    printf("  sweep, prefetch kernel (wg + 1: other XCD)   %6.2f us\n", t_pfx_b - t_pf);
    printf("  sweep, A(own 9.4 MB), B                      %6.2f us   (A %.2f)\n", t_a_b - t_a, t_a - t_sweep);
    printf("  sweep, A(own 9.4 MB + prefetch of B's), B    %6.2f us   (A %.2f)\n", t_apf_b - t_apf, t_apf - t_sweep);
  }
  unsigned *cnt, *fail;
  CK(hipMalloc(&cnt, 65536)); CK(hipMalloc(&fail, 4));
  for (int g : {64, 128, 192, 256}) {
    const int iters = 200;
    float us[3];
    for (int which = 0; which < 3; ++which) {
      CK(hipMemset(cnt, 0, 65536)); CK(hipMemset(fail, 0, 4));
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      if (which == 0) hipLaunchKernelGGL(k_bar_flat, dim3(g), dim3(256), 0, 0, cnt, g, iters, fail);
      else if (which == 1) hipLaunchKernelGGL(k_bar_16, dim3(g), dim3(256), 0, 0, cnt, g, iters, fail);
      else hipLaunchKernelGGL(k_bar_flags, dim3(g), dim3(256), 0, 0, cnt, g, iters, fail);
      hipEventRecord(e1, 0);
      CK(hipEventSynchronize(e1));
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      unsigned f = 0; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
      us[which] = f ? -1.f : ms * 1e3f / iters;
    }
    printf("lean barrier, %3d workgroups: one counter %6.2f us   16 counters %6.2f us   flag array %6.2f us   (-1 = a spin limit hit)\n", g, us[0], us[1], us[2]);
  }
  return 0;
}
