// A host-only stand-in for the HIP runtime and for the kernel launchers, so that api.cpp's HOST side (contexts, weight
// packing, workspaces, clones, graph caches, teardown orders) can run under AddressSanitizer / LeakSanitizer on a box
// without a GPU (tests/test_asan_host.py; GPU AddressSanitizer is not available on the pool).  "Device" memory is plain
// malloc, so every hipMemcpy / hipMemset size is checked by ASan on both ends; streams, events and graphs are heap
// objects, so a missing destroy shows up as a leak and a double destroy as a double free.  No kernel runs.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "kernels.h"

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) { memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip stub error"; }
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t) new int(1); return hipSuccess; }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = (hipStream_t) new int(2); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete (int*)s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { if (s) { volatile int v = *(int*)s; (void)v; } return hipSuccess; }   // touches it: use-after-destroy is caught
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t) new int(3); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t) new int(3); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete (int*)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { volatile int v = *(int*)e; (void)v; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { volatile int v = *(int*)e; (void)v; return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = (hipGraph_t) new int(4); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = (hipGraphExec_t) new int(5); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete (int*)g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete (int*)e; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { volatile int v = *(int*)e; (void)v; return hipSuccess; }
}

namespace pio {
hipError_t launch_vit_gemm(OperandType, GemmEpilogue, const GemmArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_vit_attention(OperandType, const VitAttnArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_layernorm(OperandType, const float*, const float*, const float*, float, int, int, void*, float*, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_im2col(OperandType, const float*, int, int, int, int, int, void*, hipStream_t) { return hipSuccess; }
hipError_t launch_im2col_f32(const float*, int, int, int, int, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_embed_scatter_f32(const float*, const float*, int, int, int, int, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_attention_f32(const float*, int, int, int, int, int, float, const int32_t*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_resid_ls_f32(float*, const float*, const float*, size_t, int, hipStream_t) { return hipSuccess; }
hipError_t launch_gelu_f32(float*, size_t, int, hipStream_t) { return hipSuccess; }
hipError_t launch_layernorm_f32(const float*, const float*, const float*, float, int, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_box_sequences(const float*, const int32_t*, int, int, int, int, int, int, int, int, int, float*, int32_t*, hipStream_t) { return hipSuccess; }
hipError_t launch_box_seq_reduce(const float*, const int32_t*, int, int, int, int, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_token_init(float*, const float*, const float*, const float*, int, int, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_cls_logits(const float*, int, int, int, int, int, float, float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_softmax_rows(const float*, float*, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_trace_grids(const double*, const int32_t*, int, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_bbox_weights(const int32_t*, int, int, int, int, float, const int32_t*, float*, float*, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_region_reduce(const float*, int, int, int, int, const float*, const int32_t*, int, float, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_ctx_clean(const float*, const float*, int, int, int, int, float, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_gaussian_map(int, float, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_mem_project(const ProjectArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_mem_topk(const float*, const float*, int64_t, int, float*, int, int, float*, float*, int64_t*, hipStream_t) { return hipSuccess; }
hipError_t launch_split_bank(const float*, int64_t, int, float, void*, hipStream_t) { return hipSuccess; }
hipError_t launch_abs_max(const float* x, int64_t n, uint32_t* out, hipStream_t) {
  uint32_t m = 0;
  for (int64_t i = 0; i < n; ++i) { uint32_t b; memcpy(&b, x + i, 4); b &= 0x7FFFFFFFu; m = b > m ? b : m; }
  *out = m;
  return hipSuccess;
}
hipError_t launch_row_inv_norm(const float*, int64_t M, int, float* inv, hipStream_t) { for (int64_t i = 0; i < M; ++i) inv[i] = 1.f; return hipSuccess; }
hipError_t launch_l2norm_rows(float*, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_activation_f32(float*, size_t, int, hipStream_t) { return hipSuccess; }
hipError_t launch_revert(const float*, const float*, const float*, int, int, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_decode_greedy(const DecoderArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_decode_prompted(const DecoderArgs&, const float*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_lm_score(const DecoderArgs&, const int32_t*, const int32_t*, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_lm_prefill(const DecoderArgs&, const float*, int, float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_lm_advance(const DecoderArgs&, const int32_t*, const int32_t*, int, float*, float*, float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_beam_select(const float*, const float*, const float*, const int32_t*, int, int, float*, int64_t*, hipStream_t) { return hipSuccess; }
hipError_t launch_viecap_mapping(const ViecapMapArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_sgemm_tn(const float*, int, const float*, int, const float*, float, float*, int, int, int, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_build_prompt(const float*, const int32_t*, const float*, int, int, int, int, int, int, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_preprocess(const uint8_t*, const PrepImage*, const int32_t*, uint8_t*, const float*, int, int, int, float*, hipStream_t) { return hipSuccess; }
hipError_t decoder_init() { return hipSuccess; }
hipError_t dec_split_weights(const float*, size_t, void*, float* unscale, hipStream_t) { *unscale = 1.f; return hipSuccess; }
}  // namespace pio
