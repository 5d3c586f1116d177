// Cost of an in-kernel grid barrier on MI355X (diagnostic, never shipped): G resident workgroups (one per CU) cross ITERS barriers;
// the time per barrier is what a persistent decoder-layer kernel would pay per phase instead of a kernel boundary (1.6 us of graph
// dispatch + ~2 us to the first weight byte).  Every spin loop has an exit (LIMIT polls), so a lost arrival ends the kernel, not the GPU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/grid_barrier.hip -o tools/microbench/bin/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static constexpr int LIMIT = 2000000;

// flat: one counter, every workgroup's wave 0 lane 0 arrives and polls it
__global__ __launch_bounds__(256) void k_flat(unsigned* counter, int G, int iters, unsigned* fail) {
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(it + 1) * (unsigned)G;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < LIMIT) __builtin_amdgcn_s_sleep(1);
      if (spins >= LIMIT) atomicAdd(fail, 1u);
    }
    __syncthreads();
  }
}

// two levels: the workgroups of an XCD (HW_REG_XCC_ID) meet on their own counter; the last arrival of each XCD arrives on the global
// one and, when all 8 are there, bumps a generation word that everybody polls
__global__ __launch_bounds__(256) void k_hier(unsigned* xcd_cnt /*[8][16]*/, unsigned* glob, unsigned* gen, const int* per_xcd, int iters, unsigned* fail) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned n = (unsigned)per_xcd[xcc];
      const unsigned a = __hip_atomic_fetch_add(xcd_cnt + 16 * xcc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (a == (unsigned)(it + 1) * n - 1u) {                   // last of this XCD
        const unsigned b = __hip_atomic_fetch_add(glob, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (b == (unsigned)(it + 1) * 8u - 1u) __hip_atomic_store(gen, (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      int spins = 0;
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1) && ++spins < LIMIT) __builtin_amdgcn_s_sleep(1);
      if (spins >= LIMIT) atomicAdd(fail, 1u);
    }
    __syncthreads();
  }
}

__global__ void k_count_xcd(int* per_xcd) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) atomicAdd(per_xcd + (xcc & 7u), 1);
}

int main(int argc, char** argv) {
  const int iters = 200;
  unsigned *cnt, *fail; int* per;
  CK(hipMalloc(&cnt, 4096)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&per, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {64, 128, 192, 256}) {
    for (int variant = 0; variant < 2; ++variant) {
      float best = 1e9f; unsigned hf = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(cnt, 0, 4096)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(per, 0, 64));
        if (variant == 1) {
          // the workgroup -> XCD placement of THIS launch is what the kernel itself reads; the per-XCD counts must match it, so
          // they are taken from a launch of the same grid (round-robin placement: G / 8 each when 8 | G) and checked by `fail`
          hipLaunchKernelGGL(k_count_xcd, dim3(G), dim3(256), 0, 0, per);
          CK(hipDeviceSynchronize());
        }
        CK(hipEventRecord(e0, 0));
        if (variant == 0) hipLaunchKernelGGL(k_flat, dim3(G), dim3(256), 0, 0, cnt, G, iters, fail);
        else hipLaunchKernelGGL(k_hier, dim3(G), dim3(256), 0, 0, cnt, cnt + 512, cnt + 768, per, iters, fail);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        hf += f;
        if (ms < best) best = ms;
      }
      printf("G = %3d workgroups, %s barrier: %.2f us per barrier%s\n", G, variant ? "two-level (per XCD, then global)" : "flat", best * 1e3f / iters,
             hf ? "  [a spin loop hit its limit: placement differed, figure invalid]" : "");
    }
  }
  return 0;
}
