/*
 * patchioner_hip.h -- C ABI of libpatchioner_hip.so, the MI355X (gfx950) implementation of the
 * Patch-ioner captioning hot path.
 *
 * The reference (Ruggero1912/Patch-ioner) is pure Python on PyTorch: it has no FFI.  Each entry point
 * below therefore replaces a *Python* function (or an inline block of Patchioner.forward) and cites
 * it as P/<file>:<lines> with P = /root/reference/Patch-ioner.  The reference-side binding a
 * maintainer would add is the ctypes stub shown in INTEGRATION.md (and shipped as
 * patchioner_amd/_lib.py).
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch / HIP types (streams travel as void*).
 *   - every function returns 0 on success or a negative pio_status; pio_last_error() gives the
 *     message of the last failure on the calling thread.
 *   - "dev" pointers are HIP device pointers owned by the caller; the library never frees them.
 *     Weights, the memory bank copy and all workspaces are owned by the handle (allocated in
 *     pio_create / pio_finalize_weights / pio_set_memory_bank; no allocation on the forward path).
 *   - one handle per (process, GPU); calls on one handle are serialised by the caller; all work of a
 *     call is enqueued on the given stream and nothing synchronises the device unless stated.
 *   - all floating-point tensors crossing the ABI are fp32, row-major, contiguous.
 */
#ifndef PATCHIONER_HIP_H
#define PATCHIONER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pio_context* pio_handle;
typedef void* pio_stream; /* hipStream_t */

typedef enum {
  PIO_OK = 0,
  PIO_ERR_INVALID_ARG = -1,
  PIO_ERR_HIP = -2,           /* a HIP runtime call failed (message has the HIP error string) */
  PIO_ERR_NOT_READY = -3,     /* weights / bank not loaded yet */
  PIO_ERR_UNKNOWN_WEIGHT = -4,
  PIO_ERR_SHAPE = -5,
  PIO_ERR_CAPACITY = -6       /* batch / prefix count above what pio_create sized */
} pio_status;

/* Static model description; mirrors what Patchioner.__init__ derives from the YAML config
 * (P/src/model.py:196-198, 285-286, 323-337) and decoder_config.pkl (P/src/decap/decap.py:66-71). */
typedef struct {
  /* ViT backbone (DINOv2 with registers) */
  int32_t embed_dim;          /* 384 / 768 / 1024 */
  int32_t depth;              /* 12 / 24 */
  int32_t num_heads;          /* backbone heads (12 for ViT-B) */
  int32_t patch_size;         /* 14 */
  int32_t num_registers;      /* 4 for *_reg models, else 0 */
  int32_t pretrain_grid;      /* 37: side of the learned position grid */
  int32_t crop_dim;           /* input side in pixels; must be a multiple of patch_size */
  float   vit_ln_eps;         /* 1e-6 */
  /* CLS-attention read-out quirk: 16 heads and scale 0.125 whatever the backbone (model.py:336-337) */
  int32_t readout_heads;
  float   readout_scale;
  /* DeCap decoder (GPT-2) */
  int32_t dec_layers;         /* 4 */
  int32_t dec_heads;          /* 4 */
  int32_t dec_embd;           /* 768 */
  int32_t dec_vocab;          /* 50257 */
  int32_t dec_positions;      /* 1024 */
  int32_t prefix_size;        /* clip_project input width (768, or 512 with the Talk2DINO inversion) */
  float   dec_ln_eps;         /* 1e-5 */
  /* capacities */
  int32_t max_batch;          /* images per pio_vit_forward call */
  int32_t max_prefixes;       /* prefixes per pio_decode_greedy / pio_mem_project call, <= 256 (log-probabilities: 64 rows per call) */
  int32_t max_steps;          /* decode steps (reference: 30) */
  /* numerics of the ViT MFMA path: 0 = fp16 operands, 1 = bf16 operands (fp32 accumulate either way); 2 = exact fp32
   * operands on the fp32 MFMA, a PARITY mode (plain kernels, ~20x slower) in which the whole path is held to the fp32
   * reference with no near-tie clause (csrc/vit_fp32.hip) */
  int32_t vit_operand_type;
  int32_t device;             /* HIP device ordinal */
  /* backbone family.  0: DINOv2 (torch.hub, P/src/model.py:342-343: LayerScale, exact-erf GELU, registers, interpolated
   * position grid).  1: the OpenAI-CLIP ViT the reference loads through timm with QuickGELU for its "DeCap original"
   * configurations (P/src/model.py:358-392, 786-796; configs/decap_B16*.k.yaml, decap_B32.k.yaml): patch 16 / 32, no
   * registers, norm_pre after the position add, no LayerScale, x * sigmoid(1.702 x), final norm on every token and the
   * bias-free head (embed_dim -> vit_out_dim) applied to every token; no qkv hook (has_attention = False, :864-865). */
  int32_t vit_arch;
  int32_t vit_out_dim;        /* width of the tokens pio_vit_forward returns: 0 = embed_dim; 512 for the CLIP ViT-B head */
} pio_config;

const char* pio_last_error(void);
const char* pio_version(void);

/* -- stream plumbing for pipeline.py (no reference counterpart: the reference runs on torch's current stream).
 *    A HIP stream restricted to the first `n_cus` compute units of `device` (hipExtStreamCreateWithCUMask), so that
 *    a latency-bound stage (the greedy decode) keeps compute units of its own beside the ViT GEMMs of the next
 *    batches.  n_cus <= 0 or >= the device's CU count: an ordinary stream.  `skip_cus` shifts the window. -- */
int pio_stream_create(int32_t device, int32_t skip_cus, int32_t n_cus, void** stream);
int pio_stream_destroy(void* stream);

/* -- a1: construction (replaces Patchioner.__init__ / from_config, P/src/model.py:98-662, 666-715) -- */
int pio_create(const pio_config* cfg, pio_handle* out);
int pio_destroy(pio_handle h);
/* A second decoder on the SAME weights (no reference counterpart; the reference decodes one batch at a time,
 * P/src/decap/decap.py:116-183): a handle that borrows `src`'s finalized decoder weights (read-only, not copied) and
 * owns its own activations, KV caches, scratch and decode graphs, so that pio_decode_greedy calls on `src` and on its
 * clones may run concurrently on different streams (pipeline.py decodes consecutive groups of prefixes that way).
 * Only pio_decode_greedy (and pio_destroy) are meaningful on a clone; `src` cannot be destroyed before its clones. */
int pio_clone_decoder(pio_handle src, pio_handle* out);

/* Upload one fp32 host tensor under its checkpoint key.  Keys are the reference checkpoints' own:
 *   backbone: "cls_token", "pos_embed", "register_tokens", "patch_embed.proj.weight", ...,
 *             "blocks.<i>.attn.qkv.weight", ..., "norm.bias"   (torch.hub dinov2 state_dict)
 *   decoder : "clip_project.model.0.weight|bias", "decoder.transformer.wte.weight", ...
 *             (DeCap checkpoint, P/src/decap/decap.py:61-79, 188-222)
 *   inversion: "talk2dino.A_pinv" [prefix_size, embed_dim], "talk2dino.b" [embed_dim]
 *             (P/src/model.py:618-627)
 * Unknown keys return PIO_ERR_UNKNOWN_WEIGHT (callers doing strict=False loading ignore it). */
int pio_load_weight(pio_handle h, const char* key, const float* host_data, const int64_t* shape, int32_t ndim);

/* Pack everything for the device: operand-precision copies, [out,in] transposes of the GPT-2 Conv1D
 * weights, bicubic-antialias interpolation of the position grid to crop_dim (DINOv2
 * interpolate_pos_encoding; cached per resolution).  Must be called once after the last pio_load_weight. */
int pio_finalize_weights(pio_handle h);

/* -- a9 state: the text memory bank (Im2TxtProjector.embs_dataset,
 *    P/src/decap/im2txtprojection/im2txtprojection.py:341-349).  Rows with zero norm are dropped as
 *    the reference does; inverse norms are precomputed once.  `rows_kept` may be NULL.
 *    Device memory: the fp32 copy (rows * dim * 4 B) and, unless vit_operand_type = 2 (the exact mode), a second image of the
 *    same size -- each row as the fp16 hi / lo halves of row * 2^k -- which is what pio_mem_project streams. */
int pio_set_memory_bank(pio_handle h, const float* host_bank, int64_t rows, int32_t dim, int64_t* rows_kept);

/* Same, from a device-resident fp32 bank (copied; no zero-row filtering -- the caller guarantees none). */
int pio_set_memory_bank_device(pio_handle h, const float* dev_bank, int64_t rows, int32_t dim);

/* -- a2 + a3: backbone forward with the last block's fused-QKV output captured
 *    (self.dino(imgs, is_training=True), P/src/model.py:782-783; hook P/src/dino_extraction.py:7-9).
 *    imgs_dev       [B,3,crop,crop]
 *    tokens_dev     [B,T,D]   final-LayerNorm'd tokens (cls | registers | patches), T = 1+R+n*n
 *    qkv_last_dev   [B,T,3D]  may be NULL */
int pio_vit_forward(pio_handle h, const float* imgs_dev, int32_t B, float* tokens_dev, float* qkv_last_dev,
                    pio_stream stream);

/* -- a4 + a5: CLS-row attention read-out and the two attention-weighted means
 *    (process_self_attention, P/src/dino_extraction.py:24-34; P/src/model.py:867-872).
 *    self_attn_dev [B,n*n]; head_maps_dev [B,H,n*n] (pre-softmax logits, may be NULL);
 *    avg_token_dev [B,D] (may be NULL); disentangled_dev [B,H,D] (may be NULL). */
int pio_cls_attention(pio_handle h, const float* qkv_last_dev, const float* tokens_dev, int32_t B,
                      float* self_attn_dev, float* head_maps_dev, float* avg_token_dev,
                      float* disentangled_dev, pio_stream stream);

/* -- a6: traces -> count grids (map_traces_to_grid, P/src/bbox_utils.py:158-168).
 *    xy_dev [P,2] float64 (x,y) of all points; offsets_dev [B+1] int32 CSR offsets; grids_dev [B,n,n]. */
int pio_trace_grids(pio_handle h, const double* xy_dev, const int32_t* offsets_dev, int32_t B,
                    int32_t total_points, float* grids_dev, pio_stream stream);

/* -- a7 weights: per-box weight maps (extract_bboxes_feats, P/src/bbox_utils.py:8-109).
 *    boxes_dev [B,NB,4] xywh already floor-divided by patch_size and truncated to int32;
 *    mode 0 uniform, 1 gaussian (variance > 0), 2 centre one-hot (variance == 0; `center_choice_dev`
 *    [B,NB,2] int32 gives the (y,x) picks for even spans, drawn by the host RNG like the reference's
 *    random.choice), 3 attention-map slice (attn_dev [B,n*n] is renormalised IN PLACE per box, in box
 *    order, as the reference does).
 *    weights_dev [B,NB,n*n]: zero outside the box; empty boxes give NaN rows like the reference's mean
 *    of an empty slice.  If single_map != 0 also produces the per-image normalised sum of maps
 *    (get_single_embedding_per_image) in single_dev [B,n*n], skipping boxes whose components sum < 0. */
int pio_bbox_weights(pio_handle h, const int32_t* boxes_dev, int32_t B, int32_t NB, int32_t mode, float variance,
                     const int32_t* center_choice_dev, float* attn_dev, float* weights_dev,
                     int32_t single_map, float* single_dev, pio_stream stream);

/* -- a6/a7/a8 reduction: out[r,:] = scale * sum_p weights[r,p] * patch_tokens[img[r], p, :]
 *    (P/src/model.py:1054; bbox_utils.py:53,79,109; model.py:90-92).
 *    tokens_dev [B,T,D] as written by pio_vit_forward (patch rows start at 1+R);
 *    weights_dev [R,n*n]; img_index_dev [R] int32 (NULL = row r uses image r); out_dev [R,D]. */
int pio_region_reduce(pio_handle h, const float* tokens_dev, int32_t B, const float* weights_dev,
                      const int32_t* img_index_dev, int32_t R, float scale, float* out_dev, pio_stream stream);

/* -- a8: fixed whole-image weight map (compute_region_means, P/src/model.py:45-94), variance > 0.
 *    map_dev [n*n]. variance >= 100 -> uniform. */
int pio_gaussian_map(pio_handle h, float variance, float* map_dev, pio_stream stream);

/* -- a9: memory projection (Im2TxtProjector.project, im2txtprojection.py:353-385), one pass over the bank
 *    with an online softmax.  q_dev [N,D] is L2-normalised IN PLACE (reference line 368).
 *    Arithmetic: operands as fp16 hi + lo pairs (22 significant bits), products on the fp16 matrix pipe, fp32
 *    accumulation, fp32 softmax: as close to an fp64 evaluation as an fp32 evaluation is; with
 *    vit_operand_type = 2 every product is an fp32 FMA in the reference's order.
 *    out_dev [N,D]; normalize != 0 L2-normalises the result; best_sims_dev [N,n_best] (may be NULL)
 *    receives the n_best largest cosine similarities in descending order. */
int pio_mem_project(pio_handle h, float* q_dev, int32_t N, float temperature, int32_t normalize,
                    float* out_dev, int32_t n_best, float* best_sims_dev, pio_stream stream);

/* Im2TxtProjector.project(..., return_argmax_text=True, return_n_best_sims=k) (P/src/decap/im2txtprojection/
 * im2txtprojection.py:367-375): q [N, D] is L2-normalised IN PLACE (line 368), then for every query the k largest cosine
 * similarities against the bank rows, descending (best_sims [N, k]), and the rows they belong to (best_rows [N, k], int64;
 * ties: the lower row, like torch.argmax; row indices count the rows KEPT at load, as the reference's do).  1 <= k <= 16. */
int pio_mem_topk(pio_handle h, float* q, int32_t N, int32_t k, float* best_sims, int64_t* best_rows, pio_stream stream);

/* -- f4 (bank build): ProjectionLayer.project_clip_txt (P/src/talk2dino/talk2dino.py:73-83), the step between CLIP's text
 *    features and the rows of the memory bank in Im2TxtProjector._build_support_memory (im2txtprojection.py:520-523):
 *    out = hidden_layer(act(linear_layer(x))) -- or linear_layer(x) alone when w2 is NULL (no hidden layer: the
 *    activation is then never applied, as in the reference's loop).  Exact fp32.  x [N, in_dim], w1 [out_dim, in_dim],
 *    w2 [out_dim, out_dim] (torch Linear layout), hidden = [N, out_dim] scratch (needed with w2), out [N, out_dim];
 *    act: 0 none, 1 relu, 2 tanh, 3 sigmoid (from_config's choices, :44-53); widths are multiples of 32. */
int pio_text_project(pio_handle h, const float* x, int32_t N, int32_t in_dim, const float* w1, const float* b1, int32_t out_dim,
                     const float* w2, const float* b2, int32_t act, float* hidden, float* out, pio_stream stream);

/* -- a10: (x - b) @ A_pinv^T (revert_transformation, P/src/embedding_utils.py:17-25). x_dev [N,D] ->
 *    out_dev [N,prefix_size]. */
int pio_revert_transformation(pio_handle h, const float* x_dev, int32_t N, float* out_dev, pio_stream stream);

/* -- a11 + a12: greedy decode (decoding_batched, P/src/decap/decap.py:116-155), KV-cached.
 *    prefix_dev [N,prefix_size]; ids_dev [N,steps] int32; logprob_dev [N,steps] per-token
 *    log-probabilities (may be NULL; the reference's score is exp of their row sum). */
int pio_decode_greedy(pio_handle h, const float* prefix_dev, int32_t N, int32_t steps, int32_t* ids_dev,
                      float* logprob_dev, pio_stream stream);

/* ---- ViECap head (P/src/model.py:1394-1398 -> P/src/viecap/entrypoint.py:98-153) -----------------------
 * Weights arrive through pio_load_weight under the checkpoint's own names: `mapping_network.*` (ClipCap.py:122-153) and
 * `gpt.transformer.*` (GPT2LMHeadModel; stored like DeCap's `decoder.transformer.*`).  pio_create: dec_layers = 12,
 * dec_heads = 12, max_steps >= continuous prompt + hard prompt + 63 (and >= the longest caption pio_lm_score is to score; <= 256). */
/* Entity vocabulary embeddings [K, C] (host; retrieval_categories.py:61-95 normalises them per call: done once here). */
int pio_viecap_set_entities(pio_handle h, const float* host_embeddings, int32_t K, int32_t C);
/* continuous_prompt_length of the loaded mapping network (rows of prefix_const), 0 without one */
int pio_viecap_prompt_length(pio_handle h);
/* VieCap.forward, first half (entrypoint.py:108-110): feats [N, C] L2-normalised IN PLACE, then
 * MappingNetwork.forward -> out [N, continuous_prompt_length, 768]. */
int pio_viecap_mapping(pio_handle h, float* feats, int32_t N, float* out, pio_stream stream);
/* image_text_simiarlity (retrieval_categories.py:61-95) on the already normalised feats: out [N, K] =
 * softmax(feats . entities^T / temperature). */
int pio_viecap_entity_logits(pio_handle h, const float* feats, int32_t N, float temperature, float* out, pio_stream stream);
/* VieCap.compute_perplexity (P/src/viecap/entrypoint.py:155-172): GPT2LMHeadModel(input_ids, labels = input_ids).loss per
 * caption, as a sum.  tokens [N, Lmax] int32 (row n holds lens[n] token ids, the rest is ignored), lens [N] int32, all on
 * the device; nll [N] receives sum_{p + 1 < lens[n]} -log p(token_{p+1} | token_0..p) (exact fp32 head), so that the
 * reference's loss is nll / (lens - 1) and its perplexity exp of that (lens = 1: 0 / 0, NaN, as in the reference).
 * 1 <= N <= min(max_prefixes, 64), Lmax <= max_steps. */
int pio_lm_score(pio_handle h, const int32_t* tokens, const int32_t* lens, int32_t N, int32_t Lmax, float* nll, pio_stream stream);

/* word_embed + torch.cat + greedy_search (entrypoint.py:126-150, search.py:108-191): cont [N, Lc, 768] soft prompt,
 * tokens [N, Lt] int32 hard-prompt ids already padded to one length (pad_sequence), soft_first as in the config; `steps`
 * greedy tokens (64 in the reference) with a KV cache, no attention mask, no early stop -> ids [N, steps] int32.
 * cont == NULL: only_hard_prompt (entrypoint.py:130-131) -- the prompt is the word embeddings of `tokens` alone (Lt >= 1). */
int pio_viecap_decode(pio_handle h, const float* cont, const int32_t* tokens, int32_t N, int32_t Lt, int32_t soft_first,
                      int32_t steps, int32_t* ids, pio_stream stream);

/* beam_search (P/src/viecap/search.py:193-285; VieCap.forward calls it per image when using_greedy_search is false,
 * entrypoint.py:143-148) as three device steps; the bookkeeping between them (token lists, lengths, stop flags) is the caller's.
 * The W beams of an image are the N rows of these calls and keep KV caches on the handle (the reference re-runs every beam's whole
 * sequence each step).  N <= min(16, max_prefixes); logp_dev [N, vocab] receives log(softmax(logits)) of the next token, evaluated
 * as the reference does (softmax, then log).
 *   pio_lm_prefill: embeds_dev [N, P, 768] (the prompt, repeated per beam) at positions 0..P-1.
 *   pio_lm_advance: beams re-ordered first -- row n continues the sequence of row src_rows_dev[n] (NULL: no re-ordering) -- then
 *     tokens_dev[n] (int32) is appended at position `pos` (= P + tokens generated so far - 1 ... the position of the new token).
 *   pio_beam_select: one selection.  scores_dev == NULL: the first one, the W largest of row 0.  Otherwise candidate (b, v) =
 *     stopped[b] ? (v == 0 ? scores[b] / lens[b] : -inf) : (scores[b] + logp[b][v]) / (lens[b] + 1); the W largest in descending
 *     order (ties: lower b * vocab + v first) -> out_val_dev [W] fp32, out_idx_dev [W] int64 flat indices. */
/* word_embed + torch.cat alone (entrypoint.py:126-135): prompt_dev [N, Lc + Lt, 768] as pio_viecap_decode assembles it
 * (cont_dev NULL: only_hard_prompt, Lc = 0); what beam search starts from. */
int pio_viecap_build_prompt(pio_handle h, const float* cont_dev, const int32_t* tokens_dev, int32_t N, int32_t Lt, int32_t soft_first,
                            float* prompt_dev, pio_stream stream);
int pio_lm_prefill(pio_handle h, const float* embeds_dev, int32_t N, int32_t P, float* logp_dev, pio_stream stream);
int pio_lm_advance(pio_handle h, const int32_t* tokens_dev, const int32_t* src_rows_dev, int32_t N, int32_t pos, float* logp_dev,
                   pio_stream stream);
int pio_beam_select(pio_handle h, const float* logp_dev, const float* scores_dev, const float* lens_dev, const int32_t* stopped_dev,
                    int32_t W, float* out_val_dev, int64_t* out_idx_dev, pio_stream stream);

/* -- measurement: live HIP-event timing of the launches a call makes, on the stream they are launched on.
 *    While enabled every bracketed launch records a (start, stop) event pair; pio_profile_read waits for the
 *    recorded events of one class and returns the summed device time, the launch count and the ALGORITHMIC
 *    flops / bytes of those launches (no padding, no recompute). */
typedef enum {
  PIO_PROF_VIT_GEMM = 0,     /* k_vit_gemm: patch-embed, qkv, proj, fc1, fc2 (flops) */
  PIO_PROF_VIT_ATTN = 1,     /* k_vit_attention (flops) */
  PIO_PROF_VIT_LN = 2,       /* k_layernorm (bytes) */
  PIO_PROF_MEM_PROJECT = 3,  /* k_project + combine (bytes: one pass over the bank per 16 queries) */
  PIO_PROF_DECODE = 4        /* the whole greedy-decode graph (bytes: all fp32 weights once per step) */
} pio_profile_class;
int pio_profile_enable(pio_handle h, int32_t on);   /* also clears previous records */
int pio_profile_read(pio_handle h, int32_t cls, double* total_ms, int64_t* launches, double* flops, double* bytes);

/* -- f.2: extract_bboxes_feats_double_dino (P/src/bbox_utils.py:300-403, called at model.py:983-992): for every
 *    (image, box) re-run the LAST ViT block on [cls | registers | the box's region patches] of the final tokens
 *    (cls / registers only with use_cls) and return row 0 (return_type 0, "cls") or the mean of the region rows
 *    (1, "avg"; NaN for an empty region).  "gaussian_avg" of the reference weights the INPUT patches and does not
 *    need the block: the host mirror builds those weights and calls pio_region_reduce.
 *    tokens [B][T][D] fp32 (pio_vit_forward output); slices DEVICE int32 [B*NB][4] = (y_start, y_end, x_start, x_end)
 *    patch-grid slices after Python slice normalisation; out [B*NB][D].  Sequences run in chunks of max_batch. */
int pio_bbox_double_dino(pio_handle h, const float* tokens, const int32_t* slices, int32_t B, int32_t NB, int32_t use_cls,
                         int32_t return_type, float* out, pio_stream stream);

/* -- f.4: Patchioner.ctx_cleaner (P/src/model.py:1425-1436, used by forward(cleaning_type=...) at model.py:879-922):
 *    dirty [R][D], row r cleaned with context row r / rows_per_ctx of ctx; cleaning_type 0 = "orthogonal_projection"
 *    (dirty - alpha * (dirty . ctx / |ctx|^2) * ctx), 1 = "contrastive_mask" (dirty * (1 - ctx / (|ctx| + 1e-6))).
 *    normalize_inputs: both are L2-normalised first (the clean_after_projection=False branch, model.py:907-913).
 *    out may alias dirty.  D <= 1024. -- */
int pio_ctx_clean(pio_handle h, const float* dirty, const float* ctx, int32_t R, int32_t D, int32_t rows_per_ctx,
                  int32_t cleaning_type, float alpha, int32_t normalize_inputs, float* out, pio_stream stream);

/* -- a0: model.image_transforms / image_transforms_no_crop on the device (P/src/model.py:347-357):
 *    T.Resize(resize_dim, BICUBIC) -> T.CenterCrop(crop_dim) -> T.ToTensor() -> T.Normalize(ImageNet mean / std)
 *    (mode 0), or T.Resize((resize_dim, resize_dim), BICUBIC) -> ToTensor -> Normalize (mode 1; crop_dim ignored),
 *    for B RGB uint8 images of different sizes.  Bit-exact to the host pipeline: torchvision's size rules and
 *    Pillow's two-pass 22-bit fixed-point 8-bit resampler (coefficient tables built on the host per image).
 *    pixels      : DEVICE, the images packed back to back, each H x W x 3 row-major
 *    offsets     : HOST, B byte offsets into `pixels`;  wh : HOST, B x (width, height)
 *    out         : DEVICE fp32 [B][3][S][S], S = crop_dim (mode 0) or resize_dim (mode 1) -- what pio_vit_forward takes */
int pio_preprocess(pio_handle h, const void* pixels, const int64_t* offsets, const int32_t* wh, int32_t B,
                   int32_t resize_dim, int32_t crop_dim, int32_t mode, float* out, void* stream);

/* Introspection used by the host mirror and the tests. */
int pio_num_tokens(pio_handle h);     /* T */
int pio_grid_side(pio_handle h);      /* n */
int64_t pio_bank_rows(pio_handle h);

/* Host-only (no GPU needed): the load-time position-grid interpolation of pio_finalize_weights
 * (DINOv2 interpolate_pos_encoding: bicubic, antialias, offset 0).  pos [1+grid*grid, dim] -> out [1+n*n, dim]. */
int pio_host_interpolate_pos_embed(const float* pos, int32_t grid, int32_t dim, int32_t n, float* out);
/* the same for the hub models WITHOUT registers (num_registers = 0): bicubic without antialias on the scale factor
 * (n + offset) / grid, offset 0.1 (facebookresearch/dinov2 vision_transformer.py interpolate_pos_encoding, reached from
 * P/src/model.py:342-343, 783). */
int pio_host_interpolate_pos_embed_plain(const float* pos, int32_t grid, int32_t dim, int32_t n, double offset, float* out);

/* Host-only: the Pillow resampling table pio_preprocess builds for output samples [first, first + count) of an axis
 * resized from in_size to out_size (bicubic; Resample.c precompute_coeffs + normalize_coeffs_8bpc).  kk must hold
 * count * pio_host_pil_ksize(in_size, out_size) int32, bounds 2 * count (min, taps). */
int pio_host_pil_ksize(int32_t in_size, int32_t out_size);
int pio_host_pil_table(int32_t in_size, int32_t out_size, int32_t first, int32_t count, int32_t* kk, int32_t* bounds);

#ifdef __cplusplus
}
#endif
#endif /* PATCHIONER_HIP_H */
