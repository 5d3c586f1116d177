// Stand-alone timing of the decoder kernels (diagnostic, never shipped):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I patchioner_amd/csrc tools/microbench/dec_bench.hip -o dec_bench
// Each kernel is launched in a dependent chain of ITER launches on one stream (like the decode graph) and the
// per-launch time is the elapsed time / ITER.
#include "../../patchioner_amd/csrc/decoder.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace pio;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr) *p = 0.f; }
__global__ void k_touch(float* p) { p[blockIdx.x * 256 + threadIdx.x] += 1.0f; }
__global__ __launch_bounds__(256) void k_touch_read(const float4* __restrict__ p, size_t n4, float* sink) {
  float a = 0.f;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
  if (a == 123.456f) *sink = a;
}

template <typename F>
static float time_chain(F f, int iter = 200) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iter; ++i) f();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iter;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16;
  const int E = 768, V = 50257, S = 32, heads = 4;
  float *w_qkv, *w_proj, *w_fc, *w_fc2, *w_head, *x, *qkv, *att, *hid, *part, *bias, *cvec, *kc, *vc, *wte, *wpe;
  int32_t* ids; float* lp; float* ws; unsigned* cnt;
  hipMalloc(&ws, (size_t)256 * 16 * 8 * 256 * 4); hipMalloc(&cnt, 4096); hipMemset(cnt, 0, 4096);
  CK(hipMalloc(&w_qkv, (size_t)3 * E * E * 4)); CK(hipMalloc(&w_proj, (size_t)E * E * 4));
  CK(hipMalloc(&w_fc, (size_t)4 * E * E * 4)); CK(hipMalloc(&w_fc2, (size_t)4 * E * E * 4));
  CK(hipMalloc(&w_head, (size_t)V * E * 4)); CK(hipMalloc(&wte, (size_t)V * E * 4)); CK(hipMalloc(&wpe, (size_t)1024 * E * 4));
  CK(hipMalloc(&x, (size_t)N * E * 4)); CK(hipMalloc(&qkv, (size_t)N * 3 * E * 4)); CK(hipMalloc(&att, (size_t)N * E * 4));
  CK(hipMalloc(&hid, (size_t)N * 4 * E * 4)); CK(hipMalloc(&part, (size_t)4096 * N * 4 * 4));
  CK(hipMalloc(&bias, (size_t)V * 4)); CK(hipMalloc(&cvec, (size_t)V * 4));
  CK(hipMalloc(&kc, (size_t)N * S * E * 4)); CK(hipMalloc(&vc, (size_t)N * S * E * 4));
  CK(hipMalloc(&ids, (size_t)N * S * 4)); CK(hipMalloc(&lp, (size_t)N * S * 4));
  std::vector<float> h((size_t)V * E);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((int)((i * 2654435761u) >> 20) % 2001 - 1000) * 2e-5f;
  CK(hipMemcpy(w_head, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wte, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w_qkv, h.data(), (size_t)3 * E * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w_proj, h.data(), (size_t)E * E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w_fc, h.data(), (size_t)4 * E * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w_fc2, h.data(), (size_t)4 * E * E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wpe, h.data(), (size_t)1024 * E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, h.data(), (size_t)N * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(hid, h.data(), (size_t)N * 4 * E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(att, h.data(), (size_t)N * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, h.data(), (size_t)V * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(cvec, h.data(), (size_t)V * 4, hipMemcpyHostToDevice));
  CK(hipMemset(kc, 0, (size_t)N * S * E * 4)); CK(hipMemset(vc, 0, (size_t)N * S * E * 4)); CK(hipMemset(qkv, 0, (size_t)N * 3 * E * 4));
  CK(decoder_init());
  hipStream_t s = 0;
  int nblk = 0;
  printf("N=%d\n", N);
#ifdef PIO_DEC_STAMPS
  {  // where a layer GEMM's time goes, in situ: 30 "steps" of 4 layers (own weights each) + a 77-MB sweep standing in for the head
    float *wq[4], *wp[4], *wf[4], *wg[4], *sweep;
    for (int l = 0; l < 4; ++l) {
      CK(hipMalloc(&wq[l], (size_t)3 * E * E * 4)); CK(hipMalloc(&wp[l], (size_t)E * E * 4)); CK(hipMalloc(&wf[l], (size_t)4 * E * E * 4)); CK(hipMalloc(&wg[l], (size_t)4 * E * E * 4));
      CK(hipMemcpy(wq[l], h.data(), (size_t)3 * E * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wp[l], h.data(), (size_t)E * E * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(wf[l], h.data(), (size_t)4 * E * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wg[l], h.data(), (size_t)4 * E * E * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&sweep, (size_t)77 << 20)); CK(hipMemset(sweep, 0, (size_t)77 << 20));
    auto step = [&] {
      for (int l = 0; l < 4; ++l) {
        dec_gemm<DE_STORE, 1>(wq[l], x, N, 3 * E, E, bias, qkv, nullptr, cvec, 1e-5f, nullptr, nullptr, s);
        hipLaunchKernelGGL(k_dec_attention, dim3(N * heads), dim3(256), 0, s, qkv, kc, vc, E, heads, 15, S, att);
        dec_gemm<DE_RESID, 0>(wp[l], att, N, E, E, bias, x, nullptr, nullptr, 0.f, nullptr, nullptr, s);
        dec_gemm<DE_GELU, 1>(wf[l], x, N, 4 * E, E, bias, hid, nullptr, cvec, 1e-5f, nullptr, nullptr, s);
        dec_gemm<DE_RESID, 0>(wg[l], hid, N, E, 4 * E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s);
      }
      hipLaunchKernelGGL(k_touch_read, dim3(1024), dim3(256), 0, s, (const float4*)sweep, ((size_t)77 << 20) / 16, x);
    };
    hipStream_t cs; hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    const float us = time_chain(step, 30);
    printf("emulated step (20 layer kernels + 77-MB sweep): %.1f us\n", us);
    static unsigned long long st[8][512][8];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_dec_stamps), sizeof(st)));
    const char* names[8] = {"qkv (store, LN)", "proj (resid)", "fc (gelu, LN)", "embed", "-", "fc2 (resid, split-K)", "-", "-"};
    const int nwg[8] = {3 * E / 16, E / 16, 4 * E / 16, 0, 0, 4 * (E / 16), 0, 0};
    for (int k = 0; k < 8; ++k) {
      if (!nwg[k]) continue;
      double d[7] = {0, 0, 0, 0, 0, 0, 0}; unsigned long long t0min = ~0ull, t6max = 0; int cntd = 0;
      for (int b = 0; b < nwg[k] && b < 512; ++b) {
        const unsigned long long* t = st[k][b];
        if (t[6] < t[0] || t[6] == 0) continue;      // a workgroup that left early (not the last split-K arrival)
        for (int i = 1; i < 7; ++i) if (t[i] >= t[i - 1] || i == 5) d[i] += (double)(t[i] > t[i - 1] ? t[i] - t[i - 1] : 0);
        t0min = t[0] < t0min ? t[0] : t0min; t6max = t[6] > t6max ? t[6] : t6max; ++cntd;
      }
      if (!cntd) continue;
      printf("  %-22s stamps (100-MHz ticks x10 ns or shader cycles, mean over %3d wg): issue->X staged %6.0f | barrier %6.0f | reads+MFMA %6.0f | reduce barrier %6.0f | ticket %6.0f | epilogue+stores landed %6.0f | first entry -> last exit %6.0f\n",
             names[k], cntd, d[1] / cntd, d[2] / cntd, d[3] / cntd, d[4] / cntd, d[5] / cntd, d[6] / cntd, (double)(t6max - t0min));
    }
  }
#endif
  printf("empty kernel       %7.2f us\n", time_chain([&] { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s, x); }, 1000));
  printf("touch kernel (RMW) %7.2f us\n", time_chain([&] { hipLaunchKernelGGL(k_touch, dim3(48), dim3(256), 0, s, x); }, 1000));
  {  // the same two kernels replayed from a graph of 200 nodes (what the decode loop uses)
    hipStream_t cs; hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    for (int which = 0; which < 2; ++which) {
      hipGraph_t gr; hipGraphExec_t ex;
      hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
      for (int i = 0; i < 200; ++i) { if (which == 0) hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, cs, x); else hipLaunchKernelGGL(k_touch, dim3(48), dim3(256), 0, cs, x); }
      hipStreamEndCapture(cs, &gr); hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
      printf("graph x200 %s  %7.2f us per node\n", which == 0 ? "empty" : "touch", time_chain([&] { hipGraphLaunch(ex, s); }, 20) / 200.f);
    }
  }
  printf("qkv  (LN, store)   %7.2f us   old %7.2f us\n", time_chain([&] { dec_gemm<DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, qkv, nullptr, cvec, 1e-5f, nullptr, nullptr, s); }), time_chain([&] { dec_gemm_rg<12, 1, DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, qkv, nullptr, cvec, 1e-5f, nullptr, nullptr, s); }));
  {  // round 4: what a fused c_attn + (attn.c_proj folded into V) GEMM would cost: 768 + 768 + 4 x 768 = 4608 output columns
    float *w2, *o2; CK(hipMalloc(&w2, (size_t)6 * E * E * 4)); CK(hipMalloc(&o2, (size_t)N * 6 * E * 4));
    CK(hipMemcpy(w2, h.data(), (size_t)6 * E * E * 4, hipMemcpyHostToDevice));
    printf("qkv' 4608 cols     %7.2f us\n", time_chain([&] { dec_gemm<DE_STORE, 1>(w2, x, N, 6 * E, E, bias, o2, nullptr, cvec, 1e-5f, nullptr, nullptr, s); }));
    // ... and in a chain with another kernel between (so that its X is not L2-hot from the previous launch of itself)
    printf("qkv' + proj chain  %7.2f us per pair   qkv + proj chain %7.2f us per pair\n",
           time_chain([&] { dec_gemm<DE_STORE, 1>(w2, x, N, 6 * E, E, bias, o2, nullptr, cvec, 1e-5f, nullptr, nullptr, s); dec_gemm<DE_RESID, 0>(w_proj, att, N, E, E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }),
           time_chain([&] { dec_gemm<DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, qkv, nullptr, cvec, 1e-5f, nullptr, nullptr, s); dec_gemm<DE_RESID, 0>(w_proj, att, N, E, E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }));
  }
  printf("attention pos=15   %7.2f us\n", time_chain([&] { hipLaunchKernelGGL(k_dec_attention, dim3(N * heads), dim3(256), 0, s, qkv, kc, vc, E, heads, 15, S, att); }));
  printf("attention pos=29   %7.2f us\n", time_chain([&] { hipLaunchKernelGGL(k_dec_attention, dim3(N * heads), dim3(256), 0, s, qkv, kc, vc, E, heads, 29, S, att); }));
  printf("proj (resid)       %7.2f us\n", time_chain([&] { dec_gemm<DE_RESID, 0>(w_proj, att, N, E, E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }));
  if (N > 16) {   // split-K sensitivity of the non-LN GEMMs (X traffic per workgroup / workgroup count)
    printf("proj KS2 CPW6      %7.2f us\n", time_chain([&] { dec_gemm_rg<6, 2, DE_RESID, 0>(w_proj, att, N, E, E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }));
    printf("proj KS3 CPW4      %7.2f us\n", time_chain([&] { dec_gemm_rg<4, 3, DE_RESID, 0>(w_proj, att, N, E, E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }));
    printf("proj KS6 CPW2      %7.2f us\n", time_chain([&] { dec_gemm_rg<2, 6, DE_RESID, 0>(w_proj, att, N, E, E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }));
    printf("fc2 KS8 CPW6       %7.2f us\n", time_chain([&] { dec_gemm_rg<6, 8, DE_RESID, 0>(w_fc2, hid, N, E, 4 * E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }));
    printf("fc2 KS12 CPW4      %7.2f us\n", time_chain([&] { dec_gemm_rg<4, 12, DE_RESID, 0>(w_fc2, hid, N, E, 4 * E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }));
  }
  if (N > 16) {   // X tiles through LDS (k_dec_gemm_b): shape sweep
    float* ref; float* got; CK(hipMalloc(&ref, (size_t)N * 4 * E * 4)); CK(hipMalloc(&got, (size_t)N * 4 * E * 4));
    std::vector<float> hr((size_t)N * 4 * E), hg((size_t)N * 4 * E);
    auto cmp = [&](const char* what, int cols) {
      hipDeviceSynchronize();
      hipMemcpy(hr.data(), ref, (size_t)N * cols * 4, hipMemcpyDeviceToHost); hipMemcpy(hg.data(), got, (size_t)N * cols * 4, hipMemcpyDeviceToHost);
      double md = 0, mr = 0; for (size_t i = 0; i < (size_t)N * cols; ++i) { md = fmax(md, fabs((double)hr[i] - hg[i])); mr = fmax(mr, fabs((double)hr[i])); }
      printf("   check %-10s max|diff| %.3g  (max|ref| %.3g)\n", what, md, mr);
    };
    dec_gemm_rg<12, 1, DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, ref, nullptr, cvec, 1e-5f, nullptr, nullptr, s);
    dec_gemm_rg<12, 1, DE_GELU, 1>(w_fc, x, N, 4 * E, E, bias, ref, nullptr, cvec, 1e-5f, nullptr, nullptr, s);
    dec_gemm_rg<12, 1, DE_STORE, 0>(w_proj, att, N, E, E, bias, ref, nullptr, nullptr, 0.f, nullptr, nullptr, s);
    hipMemset(ref, 0, (size_t)N * E * 4); hipMemset(got, 0, (size_t)N * E * 4);
    dec_gemm_rg<12, 4, DE_RESID, 0>(w_fc2, hid, N, E, 4 * E, bias, ref, nullptr, nullptr, 0.f, ws, cnt, s);
    dec_gemm_rg<12, 1, DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, ref, nullptr, cvec, 1e-5f, nullptr, nullptr, s);
    CK((dec_gemm_b_launch<4, 2, 1, DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, got, cvec, 1e-5f, ws, cnt, s))); cmp("b qkv 4,2", 3 * E);
    CK((dec_gemm_b_launch<2, 3, 1, DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, got, cvec, 1e-5f, ws, cnt, s))); cmp("b qkv 2,3", 3 * E);
    CK((dec_gemm_b_launch<1, 2, 1, DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, got, cvec, 1e-5f, ws, cnt, s))); cmp("b qkv 1,2", 3 * E);
    dec_gemm_rg<12, 1, DE_STORE, 0>(w_proj, att, N, E, E, bias, ref, nullptr, nullptr, 0.f, nullptr, nullptr, s);
    CK((dec_gemm_b_launch<2, 1, 1, DE_STORE, 0>(w_proj, att, N, E, E, bias, got, nullptr, 0.f, ws, cnt, s))); cmp("b proj 2,1", E);
    hipMemset(ref, 0, (size_t)N * E * 4); hipMemset(got, 0, (size_t)N * E * 4);
    dec_gemm_rg<12, 4, DE_RESID, 0>(w_fc2, hid, N, E, 4 * E, bias, ref, nullptr, nullptr, 0.f, ws, cnt, s);
    CK((dec_gemm_b_launch<2, 3, 4, DE_RESID, 0>(w_fc2, hid, N, E, 4 * E, bias, got, nullptr, 0.f, ws, cnt, s))); cmp("b fc2 2,3x4", E);
#define TB(label, R, C, KSV, EPIV, LNV, WW, XX, NO, KK, OUT, CV) \
    printf("b %-8s R%d C%d KS%d %7.2f us  (%d wg)\n", label, R, C, KSV, time_chain([&] { dec_gemm_b_launch<R, C, KSV, EPIV, LNV>(WW, XX, N, NO, KK, bias, OUT, CV, 1e-5f, ws, cnt, s); }), (NO / (16 * C)) * KSV * ((N + 16 * R - 1) / (16 * R)))
    TB("qkv", 4, 2, 1, DE_STORE, 1, w_qkv, x, 3 * E, E, qkv, cvec);
    TB("qkv", 2, 3, 1, DE_STORE, 1, w_qkv, x, 3 * E, E, qkv, cvec);
    TB("qkv", 2, 2, 1, DE_STORE, 1, w_qkv, x, 3 * E, E, qkv, cvec);
    TB("qkv", 2, 1, 1, DE_STORE, 1, w_qkv, x, 3 * E, E, qkv, cvec);
    TB("qkv", 1, 3, 1, DE_STORE, 1, w_qkv, x, 3 * E, E, qkv, cvec);
    TB("qkv", 1, 2, 1, DE_STORE, 1, w_qkv, x, 3 * E, E, qkv, cvec);
    TB("proj", 4, 1, 1, DE_RESID, 0, w_proj, att, E, E, x, nullptr);
    TB("proj", 2, 1, 1, DE_RESID, 0, w_proj, att, E, E, x, nullptr);
    TB("proj", 1, 1, 1, DE_RESID, 0, w_proj, att, E, E, x, nullptr);
    TB("fc", 4, 2, 1, DE_GELU, 1, w_fc, x, 4 * E, E, hid, cvec);
    TB("fc", 2, 3, 1, DE_GELU, 1, w_fc, x, 4 * E, E, hid, cvec);
    TB("fc", 2, 2, 1, DE_GELU, 1, w_fc, x, 4 * E, E, hid, cvec);
    TB("fc", 1, 3, 1, DE_GELU, 1, w_fc, x, 4 * E, E, hid, cvec);
    TB("fc", 1, 2, 1, DE_GELU, 1, w_fc, x, 4 * E, E, hid, cvec);
    TB("fc2", 4, 2, 4, DE_RESID, 0, w_fc2, hid, E, 4 * E, x, nullptr);
    TB("fc2", 2, 3, 4, DE_RESID, 0, w_fc2, hid, E, 4 * E, x, nullptr);
    TB("fc2", 2, 2, 4, DE_RESID, 0, w_fc2, hid, E, 4 * E, x, nullptr);
    TB("fc2", 2, 1, 4, DE_RESID, 0, w_fc2, hid, E, 4 * E, x, nullptr);
    TB("fc2", 1, 2, 4, DE_RESID, 0, w_fc2, hid, E, 4 * E, x, nullptr);
    TB("fc2", 1, 1, 4, DE_RESID, 0, w_fc2, hid, E, 4 * E, x, nullptr);
    {  // split-fp16 form (k_dec_gemm_s): weights split once, S = the power of two that puts max |w| into [2^13, 2^14)
      void *s_qkv, *s_proj, *s_fc, *s_fc2;
      CK(hipMalloc(&s_qkv, (size_t)3 * E * E * 4)); CK(hipMalloc(&s_proj, (size_t)E * E * 4)); CK(hipMalloc(&s_fc, (size_t)4 * E * E * 4)); CK(hipMalloc(&s_fc2, (size_t)4 * E * E * 4));
      const float S = 524288.f, un = 1.0f / S;
      CK(launch_dec_split_weights(w_qkv, (size_t)3 * E * E, S, s_qkv, s)); CK(launch_dec_split_weights(w_proj, (size_t)E * E, S, s_proj, s));
      CK(launch_dec_split_weights(w_fc, (size_t)4 * E * E, S, s_fc, s)); CK(launch_dec_split_weights(w_fc2, (size_t)4 * E * E, S, s_fc2, s));
      CK(hipMemcpy(x, h.data(), (size_t)N * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(hid, h.data(), (size_t)N * 4 * E * 4, hipMemcpyHostToDevice));
      dec_gemm_rg<12, 1, DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, ref, nullptr, cvec, 1e-5f, nullptr, nullptr, s);
      CK((dec_gemm_s_launch<3, 1, DE_STORE, 1>(s_qkv, un, x, N, 3 * E, E, bias, got, cvec, 1e-5f, ws, cnt, s))); cmp("s qkv C3", 3 * E);
      CK((dec_gemm_s_launch<2, 1, DE_STORE, 1>(s_qkv, un, x, N, 3 * E, E, bias, got, cvec, 1e-5f, ws, cnt, s))); cmp("s qkv C2", 3 * E);
      dec_gemm_rg<12, 1, DE_STORE, 0>(w_proj, att, N, E, E, bias, ref, nullptr, nullptr, 0.f, nullptr, nullptr, s);
      CK((dec_gemm_s_launch<3, 1, DE_STORE, 0>(s_proj, un, att, N, E, E, bias, got, nullptr, 0.f, ws, cnt, s))); cmp("s proj C3", E);
      dec_gemm_rg<12, 1, DE_GELU, 1>(w_fc, x, N, 4 * E, E, bias, ref, nullptr, cvec, 1e-5f, nullptr, nullptr, s);
      CK((dec_gemm_s_launch<3, 1, DE_GELU, 1>(s_fc, un, x, N, 4 * E, E, bias, got, cvec, 1e-5f, ws, cnt, s))); cmp("s fc C3", 4 * E);
      hipMemset(ref, 0, (size_t)N * E * 4); hipMemset(got, 0, (size_t)N * E * 4);
      dec_gemm_rg<12, 4, DE_RESID, 0>(w_fc2, hid, N, E, 4 * E, bias, ref, nullptr, nullptr, 0.f, ws, cnt, s);
      CK((dec_gemm_s_launch<3, 4, DE_RESID, 0>(s_fc2, un, hid, N, E, 4 * E, bias, got, nullptr, 0.f, ws, cnt, s))); cmp("s fc2 C3x4", E);
#define TS(label, C, KSV, EPIV, LNV, WW, XX, NO, KK, OUT, CV) \
      printf("s %-8s C%d KS%d %7.2f us  (%d wg)\n", label, C, KSV, time_chain([&] { dec_gemm_s_launch<C, KSV, EPIV, LNV>(WW, un, XX, N, NO, KK, bias, OUT, CV, 1e-5f, ws, cnt, s); }), (NO / (16 * C)) * KSV * ((N + 31) / 32))
      TS("qkv", 3, 1, DE_STORE, 1, s_qkv, x, 3 * E, E, qkv, cvec);
      TS("qkv", 2, 1, DE_STORE, 1, s_qkv, x, 3 * E, E, qkv, cvec);
      TS("proj", 2, 1, DE_RESID, 0, s_proj, att, E, E, x, nullptr);
      TS("fc", 3, 1, DE_GELU, 1, s_fc, x, 4 * E, E, hid, cvec);
      TS("fc", 2, 1, DE_GELU, 1, s_fc, x, 4 * E, E, hid, cvec);
      TS("fc2", 2, 4, DE_RESID, 0, s_fc2, hid, E, 4 * E, x, nullptr);
      TS("fc2", 3, 4, DE_RESID, 0, s_fc2, hid, E, 4 * E, x, nullptr);
#define TS1(label, C, KSV, EPIV, LNV, WW, XX, NO, KK, OUT, CV) \
      printf("s16 %-6s C%d KS%d %7.2f us  (%d wg)\n", label, C, KSV, time_chain([&] { dec_gemm_s_launch<C, KSV, EPIV, LNV, 1>(WW, un, XX, N, NO, KK, bias, OUT, CV, 1e-5f, ws, cnt, s); }), (NO / (16 * C)) * KSV * ((N + 15) / 16))
      if (N <= 128) {
        dec_gemm_rg<12, 1, DE_STORE, 1>(w_qkv, x, N, 3 * E, E, bias, ref, nullptr, cvec, 1e-5f, nullptr, nullptr, s);
        CK((dec_gemm_s_launch<3, 1, DE_STORE, 1, 1>(s_qkv, un, x, N, 3 * E, E, bias, got, cvec, 1e-5f, ws, cnt, s))); cmp("s16 qkv C3", 3 * E);
        TS1("qkv", 3, 1, DE_STORE, 1, s_qkv, x, 3 * E, E, qkv, cvec);
        TS1("qkv", 2, 1, DE_STORE, 1, s_qkv, x, 3 * E, E, qkv, cvec);
        TS1("fc", 3, 1, DE_GELU, 1, s_fc, x, 4 * E, E, hid, cvec);
        TS1("fc", 2, 1, DE_GELU, 1, s_fc, x, 4 * E, E, hid, cvec);
        TS1("fc2", 3, 4, DE_RESID, 0, s_fc2, hid, E, 4 * E, x, nullptr);
        TS1("proj", 1, 1, DE_RESID, 0, s_proj, att, E, E, x, nullptr);
      }
      TS("qkv", 4, 1, DE_STORE, 1, s_qkv, x, 3 * E, E, qkv, cvec);
      TS("fc", 4, 1, DE_GELU, 1, s_fc, x, 4 * E, E, hid, cvec);
      TS("fc2", 4, 4, DE_RESID, 0, s_fc2, hid, E, 4 * E, x, nullptr);
      TS("proj", 3, 1, DE_RESID, 0, s_proj, att, E, E, x, nullptr);
    }
    CK(hipMemcpy(x, h.data(), (size_t)N * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(hid, h.data(), (size_t)N * 4 * E * 4, hipMemcpyHostToDevice));
  }
  printf("fc   (LN, gelu)    %7.2f us\n", time_chain([&] { dec_gemm<DE_GELU, 1>(w_fc, x, N, 4 * E, E, bias, hid, nullptr, cvec, 1e-5f, nullptr, nullptr, s); }));
  printf("fc2  (resid)       %7.2f us\n", time_chain([&] { dec_gemm<DE_RESID, 0>(w_fc2, hid, N, E, 4 * E, bias, x, nullptr, nullptr, 0.f, ws, cnt, s); }));
  printf("lm head (argmax)   %7.2f us\n", time_chain([&] { launch_lmhead(w_head, x, N, V, E, bias, cvec, 1e-5f, part, &nblk, s); }, 50));
  printf("lm head generic    %7.2f us\n", time_chain([&] { dec_gemm<DE_ARGMAX, 1>(w_head, x, N, V, E, bias, part, nullptr, cvec, 1e-5f, nullptr, nullptr, s); }, 50));
  printf("select             %7.2f us\n", time_chain([&] { hipLaunchKernelGGL(k_dec_select, dim3(N), dim3(256), 0, s, part, nblk, N, E, 3, S, wte, wpe, ids, lp, x, 0); }));
  CK(hipDeviceSynchronize());
  return 0;
}
