// Where a tile's cycles go in k_project2 (diagnostic build, never shipped): per-phase s_memtime sums of wave 0 of every workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I patchioner_amd/csrc tools/microbench/project_stamps.hip -o tools/microbench/bin/project_stamps
//   project_stamps [rows]          32 queries, D = 768; split-fp16 GEMM2, then the exact fp32 form
#define PIO_PROJ_STAMPS 1
#include "../../patchioner_amd/csrc/project.hip"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace pio;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static void* split_ptr = nullptr;
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 591753;
  const int D = 768, N = 32, parts = 512;
  float *bank, *inv, *q, *out, *pacc, *pml;
  CK(hipMalloc(&bank, (size_t)M * D * 4)); CK(hipMalloc(&inv, (size_t)M * 4)); CK(hipMalloc(&q, N * D * 4)); CK(hipMalloc(&out, N * D * 4));
  CK(hipMalloc(&pacc, (size_t)parts * 16 * D * 4)); CK(hipMalloc(&pml, (size_t)parts * 16 * 2 * 4));
  {
    std::vector<float> h((size_t)M * D);
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    for (size_t i = 0; i < (size_t)4096 * D; ++i) h[i] = nd(rng);
    for (size_t i = (size_t)4096 * D; i < h.size(); ++i) h[i] = h[i - (size_t)4096 * D] * 1.0001f;   // cheap to generate, not constant
    CK(hipMemcpy(bank, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hq(N * D); for (auto& v : hq) v = nd(rng);
    CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  }
  CK(launch_row_inv_norm(bank, M, D, inv, 0));
  CK(hipMalloc(&split_ptr, (size_t)M * D * 4));
  CK(launch_split_bank(bank, M, D, 4096.0f, split_ptr, 0));
  unsigned long long* dbuf; CK(hipMalloc(&dbuf, 1024 * 8 * 8));
  std::vector<unsigned long long> h(1024 * 8);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int split = 1; split >= 0; --split) {
    ProjectArgs a; a.bank = bank; a.inv_norm = inv; a.M = M; a.D = D; a.q = q; a.N = N; a.temperature = 0.01f; a.normalize = 1; a.out = out;
    a.n_best = 0; a.best_sims = nullptr; a.part_acc = pacc; a.part_ml = pml; a.part_best = nullptr; a.parts = parts; a.n_best_cap = 16;
    a.bank_scale = split ? 4096.0f : 0.f; a.bank_split = split ? ::split_ptr : nullptr;
    for (int i = 0; i < 3; ++i) CK(launch_mem_project(a, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) CK(launch_mem_project(a, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemset(dbuf, 0, 1024 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(proj_stamps), &dbuf, sizeof(dbuf)));
    CK(launch_mem_project(a, 0)); CK(hipDeviceSynchronize());
    unsigned long long* nul = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(proj_stamps), &nul, sizeof(nul)));
    CK(hipMemcpy(h.data(), dbuf, 1024 * 8 * 8, hipMemcpyDeviceToHost));
    double ph[7] = {0}; double tiles = 0; int nwg = 0;
    for (int w = 0; w < 1024; ++w) if (h[8 * w + 7]) { ++nwg; tiles += (double)h[8 * w + 7]; for (int i = 0; i < 7; ++i) ph[i] += (double)h[8 * w + i]; }
    printf("%s: %.1f us per call (with the stamps compiled in); %d workgroups, %.1f tiles each; cycles per tile (wave 0):\n", split ? "split fp16 operands" : "exact fp32",
           ms * 100.f, nwg, tiles / nwg);
    const char* names[7] = {"softmax-end .. GEMM2 + rowsum issued", "wait own DMA", "barrier A", "DMA issue + GEMM1", "-", "barrier B", "softmax"};
    double tot = 0; for (int i = 0; i < 7; ++i) tot += ph[i];
    for (int i = 0; i < 7; ++i) printf("   %-40s %8.0f\n", names[i], ph[i] / tiles);
    printf("   %-40s %8.0f\n", "sum", tot / tiles);
  }
  return 0;
}
