// Do dependent kernel chains on different streams overlap?  (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/queue_probe.hip -o queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(float* p, int iters) {          // ~short latency-bound kernel: dependent global loads
  float v = p[threadIdx.x + blockIdx.x * 256];
  for (int i = 0; i < iters; ++i) v = p[((int)v & 1023) + threadIdx.x] + 1.0f;
  p[threadIdx.x + blockIdx.x * 256] = v;
}
int main() {
  const int NS = 8, NODES = 300;
  float* buf[NS]; hipStream_t st[NS]; hipGraphExec_t ex[NS];
  for (int s = 0; s < NS; ++s) {
    hipMalloc(&buf[s], 1 << 22); hipMemset(buf[s], 0, 1 << 22);
    hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
    hipGraph_t g;
    hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < NODES; ++i) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, st[s], buf[s], 4);
    hipStreamEndCapture(st[s], &g);
    hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0);
  }
  for (int P : {1, 2, 4, 8}) {
    for (int s = 0; s < P; ++s) hipGraphLaunch(ex[s], st[s]);
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    const int reps = 10;
    for (int r = 0; r < reps; ++r) for (int s = 0; s < P; ++s) hipGraphLaunch(ex[s], st[s]);
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
    printf("graphs x%d concurrently (%d nodes each): %.1f us per round = %.2f us per node per chain, %.2f us per node overall\n",
           P, NODES, us, us / NODES, us / NODES / P);
  }
  // same with plain launches from one host thread (round-robin over streams)
  for (int P : {1, 2, 4}) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < NODES; ++i) for (int s = 0; s < P; ++s) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, st[s], buf[s], 4);
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("eager  x%d streams: %.1f us = %.2f us per node overall\n", P, us, us / NODES / P);
  }
  return 0;
}
