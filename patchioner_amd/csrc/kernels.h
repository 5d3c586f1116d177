// Internal launch interface between api.cpp and the .hip kernel files.  Every launcher only enqueues
// work on `stream` (no allocation, no synchronisation) and returns the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pio {

enum OperandType { OP_F16 = 0, OP_BF16 = 1 };

// ---------------------------------------------------------------------------------------------
// ViT encoder
// ---------------------------------------------------------------------------------------------
enum GemmEpilogue {
  EPI_PATCH_EMBED = 0,  // x[b*Tp + G + p][n]  = acc + bias[n] + pos[(1+p)][n]          (fp32)
  EPI_QKV = 1,          // q/k [b][h][Tk][64], vT [b][h][64][Tk] (operand type) (+ fp32 qkv_last [B][T][3D])
  EPI_RESIDUAL = 2,     // x[m][n] += acc + bias[n], LayerScale folded into W and bias at load (fp32; the order of the sum: resid_join_ktile)
  EPI_GELU = 3          // out16[m][n] = gelu_erf(acc + bias[n])                         (operand type)
};

#ifndef PIO_GEMM_WARM_NEXT      // extra workgroups that touch the next GEMM's weights (GemmArgs::pf); 0: never launched
#define PIO_GEMM_WARM_NEXT 1
#endif
struct GemmArgs {
  const void* A;   // [M][lda] operand type, K contiguous
  const void* W;   // [N][K]   operand type, K contiguous (torch Linear layout)
  const float* bias;
  int M, N, K, lda;
  // epilogue operands
  float* x;            // residual stream [B*Tp][D] fp32 (PATCH_EMBED, RESIDUAL)
  const float* pos;    // interpolated position table [1+n2][D] (PATCH_EMBED)
  void* out16;         // GELU output [M][N]
  void* q; void* k; void* vT;   // QKV outputs
  float* qkv_last;     // optional fp32 capture [B][T][3D]
  int T, Tp, Tk, G, n2, D, H;  // tokens / padded rows per image / padded key count / global tokens / patches
  int act;             // EPI_GELU: 0 = exact-erf GELU (DINOv2), 1 = QuickGELU x * sigmoid(1.702 x) (OpenAI-CLIP towers)
  // optional: weights of the NEXT GEMM of the block, touched (one dword per 128-B line) by a few extra workgroups on compute units the
  // tiles leave idle, so that they wait in the Infinity Cache instead of HBM (round 4, vit_gemm.hip: "cold weights")
  const void* pf = nullptr;
  int pf_bytes = 0;
};
// the extra workgroups' body: workgroup e of ne, nthr threads each
__device__ __forceinline__ void gemm_warm_next(const GemmArgs& g, int e, int ne, int tid, int nthr) {
  const int lines = g.pf_bytes >> 7;
  const char* p = (const char*)g.pf;
  unsigned acc = 0;
  for (int l0 = e * nthr + tid; l0 < lines; l0 += 8 * ne * nthr) {       // eight loads in flight per lane
    unsigned r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int l = l0 + i * ne * nthr; r[i] = l < lines ? *(const unsigned*)(p + (size_t)l * 128) : 0u; }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc |= r[i];
  }
  asm volatile("" :: "v"(acc));
}

// EPI_RESIDUAL, the arithmetic EVERY GEMM kernel follows (one definition, so that an element of x never depends on the kernel, the
// launch size or the image's place in the launch that produced it).  LayerScale is folded into the operands at load (api.cpp:
// W' = op(ls[n] W[n][k]), b' = ls[n] b[n]), and
//     x_new[m][n] = ( sum over k ascending in MFMA steps of 16, from 0, with the OLD x[m][n] joining the sum after K-tile J(m, n) ) + b'[n]
// where a K-tile is 64 consecutive k and J(m, n) = min(e + 1, K / 64 - 1) with the class e = (m mod 4) + 4 ((n mod 256) / 128): the row's
// index mod 4 is the same wherever its image sits in a launch (images start at multiples of 8 rows: Tp % 8 == 0), the column half of the
// 256-grid always is -- a first version keyed the class to the row's place in the 256-row tile and an image's tokens changed bits
// with its position in the batch.  Why x joins in the MIDDLE of the sum: the persistent kernel (vit_gemm_roll.hip) fetches x one class
// per K-tile through 32 KiB of LDS while the multiplies run, so that the fp32 read of the residual stream costs no time of its own; at
// the end (or the start) of the sum all of a tile's x would be needed at once.  With the token on the accumulator REGISTER (32 x 32
// MFMA: register r of lane-half h holds row 8 (r >> 2) + 4 h + (r & 3)) a class is the registers r = c (mod 4) of EVERY lane, in the
// blocks of one column half.
__host__ __device__ constexpr int resid_class(int m, int n) { return (m & 3) + 4 * ((n >> 7) & 1); }
__host__ __device__ constexpr int resid_join_ktile(int e, int nk) { return e + 1 < nk - 1 ? e + 1 : nk - 1; }

hipError_t launch_vit_gemm(OperandType t, GemmEpilogue epi, const GemmArgs& a, hipStream_t s);
// 256 x 256 tiles, 8 waves (vit_gemm256.hip): same arithmetic per output element; launch_vit_gemm dispatches to it
bool vit_gemm256_fits(GemmEpilogue epi, const GemmArgs& a);
hipError_t launch_vit_gemm256(OperandType t, GemmEpilogue epi, const GemmArgs& a, hipStream_t s);
// persistent 256 x 256 workgroups with a rolling epilogue (vit_gemm_roll.hip): qkv without capture and fc1 at many tiles per CU, proj / fc2
// (EPI_RESIDUAL) from one tile per CU on
bool vit_gemm_roll_fits(GemmEpilogue epi, const GemmArgs& a);
hipError_t launch_vit_gemm_roll(OperandType t, GemmEpilogue epi, const GemmArgs& a, hipStream_t s);

struct VitAttnArgs {
  const void* q; const void* k; const void* vT;  // as written by EPI_QKV
  void* out;       // [B*Tp][D] operand type
  int B, H, T, Tp, Tk, D;
  float scale;     // head_dim^-0.5
  const int32_t* lens = nullptr;   // optional [B]: tokens of each sequence (keys >= lens[b] are masked); else T for all
};
hipError_t launch_vit_attention(OperandType t, const VitAttnArgs& a, hipStream_t s);

// LayerNorm rows of x [M][D] fp32 -> operand type (out16) or fp32 (out32; compacts Tp -> T rows per image)
hipError_t launch_layernorm(OperandType t, const float* x, const float* w, const float* b, float eps, int M,
                            int D, void* out16, float* out32, int T, int Tp, hipStream_t s);
// ---- vit_fp32.hip: the exact-fp32 parity mode of the backbone (vit_operand_type = 2); plain kernels
hipError_t launch_im2col_f32(const float* imgs, int B, int S, int p, int n, int Kpad, float* out, hipStream_t s);
hipError_t launch_embed_scatter_f32(const float* emb, const float* pos, int B, int n2, int Tp, int G, int D, float* x, hipStream_t s);
hipError_t launch_attention_f32(const float* qkv, int B, int H, int T, int Tp, int D, float scale, const int32_t* lens, float* out, hipStream_t s);
hipError_t launch_resid_ls_f32(float* x, const float* y, const float* ls, size_t total, int D, hipStream_t s);
hipError_t launch_gelu_f32(float* y, size_t n, int act, hipStream_t s);
hipError_t launch_layernorm_f32(const float* x, const float* w, const float* b, float eps, int M, int D, float* y, hipStream_t s);

// imgs [B][3][S][S] fp32 -> patch rows [B*n2][Kpad] (k = c*p*p + py*p + px; columns >= 3*p*p stay zero)
hipError_t launch_im2col(OperandType t, const float* imgs, int B, int S, int p, int n, int Kpad, void* out,
                         hipStream_t s);
// x[b][0] = cls + pos[0]; x[b][1..R] = registers; x[b][T..Tp-1] = 0
// double-DINO boxes (P/src/bbox_utils.py:300-403): sequence s = (image, box) -> x[s] = [cls | registers | region patches]
// (final tokens; `slices` [Ns][4] = python-normalised ys, ye, xs, xe), zero rows up to Tp; lens[s] = its token count
hipError_t launch_box_sequences(const float* tokens, const int32_t* slices, int s_base, int Ns, int NB, int T, int Tp, int G,
                                int n, int D, int use_global, float* x, int32_t* lens, hipStream_t s);
// after the block: mode 0 = row 0 (cls), 1 = mean of rows [G_s, lens) (NaN for an empty region, like torch.mean)
hipError_t launch_box_seq_reduce(const float* x, const int32_t* lens, int Ns, int Tp, int D, int Gs, int mode, float* out,
                                 hipStream_t s);
hipError_t launch_token_init(float* x, const float* cls, const float* pos0, const float* reg, int B, int R, int T,
                             int Tp, int D, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// attention read-out and region weighting (all fp32)
// ---------------------------------------------------------------------------------------------
// mean_logits [B][n2] = head-mean of (q_cls*scale).k_patch ; head_logits [B][Hr][n2] (may be null)
hipError_t launch_cls_logits(const float* qkv_last, int B, int T, int G, int D, int Hr, float scale,
                             float* mean_logits, float* head_logits, hipStream_t s);
hipError_t launch_softmax_rows(const float* in, float* out, int rows, int n, hipStream_t s);
hipError_t launch_trace_grids(const double* xy, const int32_t* offsets, int B, int n, float* grids, hipStream_t s);
hipError_t launch_bbox_weights(const int32_t* boxes, int B, int NB, int n, int mode, float variance,
                               const int32_t* center_choice, float* attn, float* weights, int single_map,
                               float* single, hipStream_t s);
hipError_t launch_region_reduce(const float* tokens, int T, int G, int D, int n2, const float* weights,
                                const int32_t* img_index, int R, float scale, float* out, hipStream_t s);
hipError_t launch_ctx_clean(const float* dirty, const float* ctx, int R, int D, int rows_per_ctx, int mode, float alpha,
                            int normalize_inputs, float* out, hipStream_t s);
hipError_t launch_gaussian_map(int n, float variance, float* map, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// memory projection
// ---------------------------------------------------------------------------------------------
struct ProjectArgs {
  const float* bank;      // [M][D] raw rows
  const float* inv_norm;  // [M]
  int64_t M; int D;
  float* q;               // [N][D], normalised in place
  int N;
  float temperature;
  int normalize;
  float* out;             // [N][D]
  int n_best; float* best_sims;
  // workspaces (owned by the context)
  float* part_acc;        // [parts][N][D]
  float* part_ml;         // [parts][N][2]  (running max, running sum)
  float* part_best;       // [parts][N][n_best_cap]
  int parts; int n_best_cap;
  int part_rows = 0;      // rows of part_acc / part_ml the context allocated (16 parts; 24 parts allows the 48-query pass)
  float bank_scale;       // > 0 with bank_split: both GEMMs on split fp16 operands (bank * bank_scale = hi + lo; a power of two with
  const void* bank_split; // max|bank| * scale <= 2^15; [M][2][D] fp16, launch_split_bank); 0 / null: the exact fp32 form
};
hipError_t launch_mem_project(const ProjectArgs& a, hipStream_t s);
// q normalised in place, then top-k cosine similarities and their rows per query (sims_scratch: [16][M] floats)
hipError_t launch_mem_topk(const float* bank, const float* inv_norm, int64_t M, int D, float* q, int N, int k, float* sims_scratch,
                           float* best_sims, int64_t* best_rows, hipStream_t s);
hipError_t launch_row_inv_norm(const float* bank, int64_t M, int D, float* inv_norm, hipStream_t s);
// out[0] = max |x[i]| as the bit pattern of a non-negative float (zeroed by the launcher); NaN / inf give >= 0x7F800000
hipError_t launch_abs_max(const float* x, int64_t n, uint32_t* out, hipStream_t s);
// out[row] = [fp16 hi of bank[row] * scale (D values) | fp16 lo (D values)]: the operand image of the split-fp16 projection
hipError_t launch_split_bank(const float* bank, int64_t M, int D, float scale, void* out, hipStream_t s);
hipError_t launch_revert(const float* x, const float* b, const float* A_pinv, int N, int D, int P, float* out,
                         hipStream_t s);

// ---------------------------------------------------------------------------------------------
// DeCap decoder (fp32, KV-cached greedy)
// ---------------------------------------------------------------------------------------------
// LayerNorm-folded linear layers (decoder.hip header): w = W * ln_w per input channel, c_j = sum_k w_jk,
// d_j = sum_k ln_b_k W_jk + b_j.
struct DecLayerW {
  const float *attn_w /*[3E][E] folded*/, *attn_c, *attn_d, *proj_w /*[E][E]*/, *proj_b;
  const float *fc_w /*[4E][E] folded*/, *fc_c, *fc_d, *fc2_w /*[E][4E]*/, *fc2_b;
  // the same matrices as split-fp16 operands (decoder.hip, k_dec_gemm_s; null: fp32 kernels only) and 1 / (their power-of-two scale x the
  // activations' scale)
  const void *attn_ws = nullptr, *fc_ws = nullptr, *fc2_ws = nullptr;
  float attn_un = 0.f, fc_un = 0.f, fc2_un = 0.f;
};
// W [n] fp32 -> split-fp16 layout ([row][8-k group][hi x 8 | lo' x 8]), same bytes; *unscale = 1 / (S x activation scale), S = the power of
// two that puts max |W| into [2^13, 2^14).  Synchronises (load time only).
hipError_t dec_split_weights(const float* W, size_t n, void* out, float* unscale, hipStream_t s);
static constexpr int DEC_MAX_PREFIXES = 256;       // rows of one greedy decode (ids only; log-probabilities: 64 per call)
static constexpr int DEC_SPLITK_COUNTERS = 128;    // arrival tickets / slab groups of the in-launch split-K (api.cpp allocates them)
static constexpr int DEC_TICKET_WORDS = DEC_SPLITK_COUNTERS + 2;   // ... + the LM head's arrival ticket and its tail's done-count (decoder.hip: k_lmhead_f16_fused<true>)
struct DecoderArgs {
  int N, steps, E, heads, layers, vocab, prefix_size;
  float eps;
  const float* prefix;        // [N][prefix_size]
  const float *clip_w /*[E][prefix]*/, *clip_b, *wte /*[V][E]*/, *wpe /*[P][E]*/;
  const float *head_w /*[V][E] = wte * ln_f.weight*/, *head_c, *head_d;
  const DecLayerW* layer;     // host array [layers]
  // workspaces
  float* x;      // [N][E] residual
  float* qkv;    // [N][3E]
  float* att;    // [N][E]
  float* hid;    // [N][4E]
  float* kcache; // [layers][N][max_steps][E]
  float* vcache;
  int max_steps;
  float* splitk_ws;       // [64 column groups][4 k-slices][4 row groups][256] partial tiles
  unsigned* splitk_cnt;   // [DEC_TICKET_WORDS] arrival tickets (zero between launches)
  int lm_tail;            // <= 16 prefixes: k_dec_select_filter as the ticketed tail of the LM head kernel (off by default: measured slower)
  float* logits; // [ceil(V/16)][N][4] per-workgroup (max, arg-max, sum-exp) partials of the LM head, or the
                 // approximate logits [N][round_up(V, 64)] of the fp16 filter
  // fp16 filter of the LM head (null head_w16: exact head only)
  const uint16_t* head_w16 = nullptr;   // [V][E] fp16 bits of head_w * 2^s
  float head_w16_unscale = 1.f;         // 2^-s
  float head_bound_coef = 0.f;          // 1.25e-3 * max_v |head_w[v]|_2
  void* xh = nullptr;                   // [N][E] fp16 copy of x (scaled per row)
  float* lm_stats = nullptr;            // [N][4] mu, rstd, 2^e, error bound
  float* lm_gmax = nullptr;             // [N][round_up(ceil(V/16), 64)] maxima of the approximate logits per 16 columns
  int32_t* ids;      // [N][steps]
  float* logprob;    // [N][steps] or null
  int pos_base = 0;  // position of the input that produces ids[.][0] (0: the DeCap prefix; P - 1 after a P-position prompt)
  // optional workspace of the batched prompt prefill (launch_decode_prompted): pre_rows rows of x / qkv / att / hid; null: position by position
  float *pre_x = nullptr, *pre_qkv = nullptr, *pre_att = nullptr, *pre_hid = nullptr;
  int pre_rows = 0;
};
hipError_t launch_decode_greedy(const DecoderArgs& a, hipStream_t s);
// ViECap greedy search: prompt [N][P][E] at positions 0..P-1, then a.steps greedy tokens (a.pos_base = P - 1)
hipError_t launch_decode_prompted(const DecoderArgs& a, const float* prompt, int P, hipStream_t s);
// teacher-forced pass over given tokens [N][Lmax] (rows of lens[n] tokens): nll[n] = sum over p + 1 < lens[n] of -log p(token p+1 | tokens <= p); N <= 64
hipError_t launch_lm_score(const DecoderArgs& a, const int32_t* tokens, const int32_t* lens, int Lmax, float* nll, hipStream_t s);
// beam search building blocks (N <= 16 beams per call; stats: [N][2] scratch; logp [N][V] = log softmax of the next-token logits)
hipError_t launch_lm_prefill(const DecoderArgs& a, const float* embeds, int P, float* stats, float* logp, hipStream_t s);
hipError_t launch_lm_advance(const DecoderArgs& a, const int32_t* tokens, const int32_t* src_rows, int pos, float* kscratch,
                             float* vscratch, float* stats, float* logp, hipStream_t s);
hipError_t launch_beam_select(const float* logp, const float* scores, const float* lens, const int32_t* stopped, int W, int V,
                              float* out_val, int64_t* out_idx, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// ViECap head (viecap.hip): mapping network, entity logits, prompt assembly -- all fp32
// ---------------------------------------------------------------------------------------------
struct ViecapMapLayerW {
  const float *n1w, *n1b, *q_w /*[E][E]*/, *kv_w /*[2E][E]*/, *proj_w, *proj_b, *n2w, *n2b, *fc1_w /*[H][E]*/, *fc1_b, *fc2_w /*[E][H]*/, *fc2_b;
};
struct ViecapMapArgs {
  int N, C, E, Lp, Lc, heads, layers, hidden;
  float eps;
  const float* feats;        // [N][C], already L2-normalised
  const float *lin_w /*[Lp*E][C]*/, *lin_b, *prefix_const /*[Lc][E]*/;
  const ViecapMapLayerW* layer;
  float *lin, *x, *ln, *q, *kv, *att, *hid;   // workspaces: [N][Lp*E], [M][E] x 4, [M][2E], [M][H] with M = N (Lp + Lc)
  float* out;                // [N][Lc][E]
};
hipError_t launch_viecap_mapping(const ViecapMapArgs& a, hipStream_t s);
// C = alpha * A W^T (+ bias) (ReLU) (+= C): exact fp32, any M, N; K % 32 == 0
hipError_t launch_sgemm_tn(const float* A, int lda, const float* W, int ldw, const float* bias, float alpha, float* C, int ldc,
                           int M, int N, int K, int relu, int resid, hipStream_t s);
hipError_t launch_activation_f32(float* y, size_t n, int act, hipStream_t s);     // 2 tanh, 3 sigmoid, in place
hipError_t launch_build_prompt(const float* cont, const int32_t* tokens, const float* wte, int N, int Lc, int Lt, int E, int V,
                               int soft_first, float* prompt, hipStream_t s);
hipError_t launch_l2norm_rows(float* x, int N, int D, hipStream_t s);
// ---- preprocess.hip: image_transforms on the device (P/src/model.py:347-357; Pillow's 8-bit two-pass resampler) ----
struct PrepImage {
  int64_t src;                              // byte offset of the image's first pixel in `pixels` (RGB, HWC, row-major)
  int64_t tmp;                              // byte offset of its intermediate image [nr][nx][3] in `tmp`
  int32_t W, H;                             // source size
  int32_t x0, y0, nx, ny;                   // output window [y0, y0+ny) x [x0, x0+nx) that holds image content
  int32_t kh, kv;                           // taps per output column / row (strides of the coefficient tables)
  int32_t r0, nr;                           // first source row the vertical pass needs, number of such rows
  int32_t coef_h, coef_v, bnd_h, bnd_v;     // int32 offsets into `tables`: coefficients [n][k], bounds [n][2] = (min, count)
};
hipError_t launch_preprocess(const uint8_t* pixels, const PrepImage* imgs, const int32_t* tables, uint8_t* tmp,
                             const float* lut, int B, int S, int max_tmp_elems, float* out, hipStream_t s);

hipError_t decoder_init();   // one-time function attributes (call outside stream capture)

}  // namespace pio
