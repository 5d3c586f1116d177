// C ABI of libpatchioner_hip.so (see include/patchioner_hip.h): context, weight packing, launch plans.
#include "../../include/patchioner_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "host_prep.h"
#include "kernels.h"

using namespace pio;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return fail(PIO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                \
  } while (0)

// Bracket `launch` with HIP events on its stream when profiling is on (class ids: pio_profile_class).
#define PROF(c, cls_id, fl, by, stream, launch)                                                   \
  do {                                                                                            \
    hipEvent_t _a = nullptr, _b = nullptr;                                                        \
    if ((c)->prof_on) { _a = (c)->next_event(); _b = (c)->next_event(); }                         \
    if (_a && _b) HIP_OK(hipEventRecord(_a, (stream)));                                           \
    HIP_OK(launch);                                                                               \
    if (_a && _b) {                                                                               \
      HIP_OK(hipEventRecord(_b, (stream)));                                                       \
      (c)->prof.push_back({(cls_id), _a, _b, (double)(fl), (double)(by)});                        \
    }                                                                                             \
  } while (0)

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct VitLayerDev {
  float *n1w, *n1b, *qkvb, *projb, *ls1, *n2w, *n2b, *fc1b, *fc2b, *ls2;
  float *projb_ls, *fc2b_ls;     // ls1 * proj.bias, ls2 * fc2.bias: the MFMA path's residual GEMMs take LayerScale folded into their operands
  void *qkvw, *projw, *fc1w, *fc2w;     // projw / fc2w: rows scaled by ls1 / ls2 before the conversion to the operand type (kernels.h, EPI_RESIDUAL)
};

struct GraphKey {
  int N, steps, lp;
  bool operator<(const GraphKey& o) const {
    if (N != o.N) return N < o.N;
    if (steps != o.steps) return steps < o.steps;
    return lp < o.lp;
  }
};

}  // namespace

struct pio_context {
  pio_config cfg;
  int n = 0, n2 = 0, T = 0, Tp = 0, Tk = 0, G = 0, D = 0, Kpe = 0, Kpad = 0, H = 0;
  OperandType op = OP_F16;
  bool finalized = false, has_vit = false, has_dec = false, has_inv = false;
  std::unordered_map<std::string, HostTensor> host;
  std::vector<void*> allocs;
  pio_context* parent = nullptr;   // pio_clone_decoder: the handle whose decoder weights this one borrows
  int clones = 0;                  // live decoder clones of this handle (it cannot be destroyed before them)

  // ViT weights
  void* pe_w = nullptr; float* pe_b = nullptr; float* pos = nullptr; float* cls = nullptr; float* reg = nullptr;
  float *norm_w = nullptr, *norm_b = nullptr;
  float *npre_w = nullptr, *npre_b = nullptr;   // CLIP: norm_pre
  float* vhead_w = nullptr;                      // CLIP: head.weight [vit_out_dim][D], fp32
  float* ones = nullptr;                         // CLIP: "LayerScale" of ones (x += 1 * branch is exact)
  float* tok_tmp = nullptr;                      // CLIP: final-norm tokens [max_batch * T][D] before the head
  int Dout = 0;                                  // width of the returned tokens
  // exact-fp32 parity mode (vit_operand_type = 2, vit_fp32.hip): fp32 copies of the GEMM weights and fp32 activations
  bool vit_f32 = false;
  struct VitLayerF32 { float *qkvw, *projw, *fc1w, *fc2w; };
  std::vector<VitLayerF32> vl32;
  float *pe_w32 = nullptr, *f_ape = nullptr, *f_emb = nullptr, *f_xn = nullptr, *f_qkv = nullptr, *f_ao = nullptr, *f_h = nullptr;
  std::vector<VitLayerDev> vl;
  // ViT workspaces
  float* x = nullptr; void* xn = nullptr; void* ao = nullptr; void* hbuf = nullptr; void* ape = nullptr;
  void *q = nullptr, *k = nullptr, *vT = nullptr;
  int32_t* seq_lens = nullptr;   // double-DINO boxes: token count of each sequence of the current chunk
  // read-out workspaces
  float* head_logits = nullptr; float* head_sm = nullptr; int32_t* head_img = nullptr;
  // decoder weights
  float *clip_w = nullptr, *clip_b = nullptr, *wte = nullptr, *wpe = nullptr, *head_w = nullptr, *head_c = nullptr,
        *head_d = nullptr;
  uint16_t* head_w16 = nullptr; float head_w16_unscale = 1.f, head_bound_coef = 0.f;   // fp16 arg-max filter of the LM head
  void* dec_xh = nullptr; float* lm_stats = nullptr; float* lm_gmax = nullptr;
  std::vector<DecLayerW> dl;
  // decoder workspaces
  float *dx = nullptr, *dqkv = nullptr, *datt = nullptr, *dhid = nullptr, *kcache = nullptr,
        *vcache = nullptr, *logits = nullptr, *splitk_ws = nullptr, *prefix_buf = nullptr, *logprob_buf = nullptr;
  int32_t* ids_buf = nullptr;
  unsigned* splitk_cnt = nullptr;
  std::map<GraphKey, hipGraphExec_t> graphs;
  hipStream_t capture_stream = nullptr;
  bool use_graph = true;
  bool lm_tail = false;                           // <= 16 prefixes: the arg-max filter as the ticketed tail of the LM head kernel (PIO_LM_TAIL=1 at pio_create; measured slower, decoder.hip)
  bool batched_prefill = true;                    // prompted decodes take their prompt through the layers in one batch (PIO_DEC_PREFILL=0 at pio_create: position by position)
  // inversion
  float *A_pinv = nullptr, *inv_b = nullptr;
  // ViECap head (viecap.hip): mapping network weights / workspaces, entity embeddings, prompt buffer
  bool has_clip_project = false, has_map = false;
  int map_C = 0, map_Lp = 0, map_Lc = 0, map_hidden = 0, map_layers = 0;
  float *map_lin_w = nullptr, *map_lin_b = nullptr, *map_prefix = nullptr;
  std::vector<ViecapMapLayerW> ml;
  float *map_lin = nullptr, *map_x = nullptr, *map_ln = nullptr, *map_q = nullptr, *map_kv = nullptr, *map_att = nullptr, *map_hid = nullptr;
  float* ent = nullptr; int ent_K = 0;            // [K][C] L2-normalised entity text embeddings
  float* prompt_buf = nullptr;                    // [max_prefixes][max_steps][E]
  float *pre_x = nullptr, *pre_qkv = nullptr, *pre_att = nullptr, *pre_hid = nullptr;   // batched prompt prefill (decoder.hip), kPrefillRows rows
  static constexpr int kPrefillRows = 2048;
  int32_t* tok_buf = nullptr;                     // [max_prefixes][max_steps]
  struct PKey { int N, P, steps; bool operator<(const PKey& o) const { return N != o.N ? N < o.N : (P != o.P ? P < o.P : steps < o.steps); } };
  // prompted-decode graphs, keyed by (rows, prompt positions, steps).  The prompt length follows the hard prompt of every batch,
  // so a long run meets many keys: the cache is a small LRU (a graph is ~ (P + steps) x 60 kernel nodes) and evicted execs are destroyed.
  struct PGraph { hipGraphExec_t exec; uint64_t last_use; hipEvent_t done; };     // done: recorded behind the graph's last launch (valid after the caller's stream has gone)
  std::map<PKey, PGraph> pgraphs;
  uint64_t pgraph_clock = 0;
  static constexpr size_t kMaxPGraphs = 12;
  // memory bank
  float* bank = nullptr; float* bank_inv = nullptr; int64_t bank_rows = 0; int bank_dim = 0;
  float bank_scale = 0.f;   // > 0: the projection runs on split fp16 operands of bank * bank_scale (project.hip) ...
  float* bank_split = nullptr;   // ... [rows][2][dim] fp16 (hi plane, lo plane), the same bytes again as the fp32 bank
  float *beam_k = nullptr, *beam_v = nullptr, *beam_stats = nullptr;   // beam search: KV gather scratch (allocated on first use), row statistics
  float *part_acc = nullptr, *part_ml = nullptr, *sims = nullptr;
  int parts = 512;   // two k_project workgroups per CU
  // device-side image transforms (pio_preprocess): growable intermediate image, a ring of table slots
  struct PrepSlot { void* host = nullptr; void* dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; };
  PrepSlot prep_slots[4];
  int prep_next = 0;
  uint8_t* prep_tmp = nullptr; size_t prep_tmp_cap = 0;
  float* prep_lut = nullptr;
  // live HIP-event profiling (pio_profile_*): one (start, stop) pair per bracketed launch
  struct ProfRec { int cls; hipEvent_t a, b; double flops, bytes; };
  bool prof_on = false;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  hipEvent_t next_event() {
    if (ev_used == ev_pool.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      ev_pool.push_back(e);
    }
    return ev_pool[ev_used++];
  }

  template <typename Tt>
  int dmalloc(Tt** p, size_t count, bool zero = false) {
    void* ptr = nullptr;
    HIP_OK(hipMalloc(&ptr, count * sizeof(Tt) > 0 ? count * sizeof(Tt) : 16));
    if (zero) HIP_OK(hipMemset(ptr, 0, count * sizeof(Tt)));
    allocs.push_back(ptr);
    *p = (Tt*)ptr;
    return PIO_OK;
  }
  int dmalloc_bytes(void** p, size_t bytes, bool zero = false) {
    HIP_OK(hipMalloc(p, bytes > 0 ? bytes : 16));
    if (zero) HIP_OK(hipMemset(*p, 0, bytes));
    allocs.push_back(*p);
    return PIO_OK;
  }
};

namespace {

size_t op_size(OperandType) { return 2; }

const HostTensor* find(pio_context* c, const std::string& key) {
  auto it = c->host.find(key);
  return it == c->host.end() ? nullptr : &it->second;
}

int need(pio_context* c, const std::string& key, std::vector<int64_t> shape, const HostTensor** out) {
  const HostTensor* t = find(c, key);
  if (!t) return fail(PIO_ERR_NOT_READY, "missing weight '" + key + "'");
  int64_t want = 1;
  for (auto s : shape) want *= s;
  if (t->numel() != want) {
    return fail(PIO_ERR_SHAPE, "weight '" + key + "' has " + std::to_string(t->numel()) + " elements, expected " +
                                   std::to_string(want));
  }
  *out = t;
  return PIO_OK;
}

int upload_f32(pio_context* c, const float* src, size_t count, float** dst) {
  int rc = c->dmalloc(dst, count);
  if (rc) return rc;
  HIP_OK(hipMemcpy(*dst, src, count * sizeof(float), hipMemcpyHostToDevice));
  return PIO_OK;
}

// fp32 [rows][cols] -> operand-precision [rows][cols_pad] (zero padded), uploaded
int upload_op(pio_context* c, const float* src, int64_t rows, int64_t cols, int64_t cols_pad, void** dst) {
  std::vector<uint16_t> tmp((size_t)rows * cols_pad, 0);
  for (int64_t r = 0; r < rows; ++r) convert_row(c->op == OP_F16, src + r * cols, cols, tmp.data() + r * cols_pad);
  int rc = c->dmalloc_bytes(dst, tmp.size() * 2);
  if (rc) return rc;
  HIP_OK(hipMemcpy(*dst, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  return PIO_OK;
}

// Conv1D weight [in][out] -> [out][in] fp32, uploaded
int upload_transposed(pio_context* c, const float* src, int64_t in, int64_t out, float** dst) {
  std::vector<float> tmp((size_t)in * out);
  for (int64_t i = 0; i < in; ++i)
    for (int64_t o = 0; o < out; ++o) tmp[(size_t)o * in + i] = src[(size_t)i * out + o];
  return upload_f32(c, tmp.data(), tmp.size(), dst);
}

int finalize_vit(pio_context* c) {
  const int D = c->D, depth = c->cfg.depth, p = c->cfg.patch_size, R = c->cfg.num_registers;
  const HostTensor* t;
  int rc;
  const bool clip = c->cfg.vit_arch == 1;
  if ((rc = need(c, "patch_embed.proj.weight", {D, 3, p, p}, &t))) return rc;
  if ((rc = upload_op(c, t->data.data(), D, c->Kpe, c->Kpad, &c->pe_w))) return rc;
  if (clip && !find(c, "patch_embed.proj.bias")) {          // CLIP's conv1 has no bias (timm: patch_embed.proj.bias absent)
    const std::vector<float> zeros(D, 0.f);
    if ((rc = upload_f32(c, zeros.data(), D, &c->pe_b))) return rc;
  } else {
    if ((rc = need(c, "patch_embed.proj.bias", {D}, &t))) return rc;
    if ((rc = upload_f32(c, t->data.data(), D, &c->pe_b))) return rc;
  }
  if ((rc = need(c, "cls_token", {D}, &t))) return rc;
  if ((rc = upload_f32(c, t->data.data(), D, &c->cls))) return rc;
  if (R > 0) {
    if ((rc = need(c, "register_tokens", {R, D}, &t))) return rc;
    if ((rc = upload_f32(c, t->data.data(), (size_t)R * D, &c->reg))) return rc;
  }
  const int g = c->cfg.pretrain_grid;
  if ((rc = need(c, "pos_embed", {1 + (int64_t)g * g, D}, &t))) return rc;
  {
    std::vector<float> pos((size_t)(1 + c->n2) * D);
    // hub variants: *_reg models interpolate with antialias and offset 0, the models without registers without
    // antialias and with the historical 0.1 offset (dinov2/hub/backbones.py)
    // CLIP (timm.create_model(..., img_size=resize_dim), P/src/model.py:371): the checkpoint's table is resampled at load by
    // timm's resample_abs_pos_embed -- class position kept apart, F.interpolate(size=, mode="bicubic", antialias=True) on the
    // grid -- the same arithmetic as the *_reg DINOv2 models; identity at the native 224 x 224 (configs/decap_B16_resize.k.yaml: 592)
    if (clip || R > 0) interpolate_pos_embed(t->data.data(), g, D, c->n, pos.data());
    else interpolate_pos_embed_plain(t->data.data(), g, D, c->n, 0.1, pos.data());
    if ((rc = upload_f32(c, pos.data(), pos.size(), &c->pos))) return rc;
  }
  if ((rc = need(c, "norm.weight", {D}, &t))) return rc;
  if ((rc = upload_f32(c, t->data.data(), D, &c->norm_w))) return rc;
  if ((rc = need(c, "norm.bias", {D}, &t))) return rc;
  if ((rc = upload_f32(c, t->data.data(), D, &c->norm_b))) return rc;
  if (clip) {
    if ((rc = need(c, "norm_pre.weight", {D}, &t))) return rc;
    if ((rc = upload_f32(c, t->data.data(), D, &c->npre_w))) return rc;
    if ((rc = need(c, "norm_pre.bias", {D}, &t))) return rc;
    if ((rc = upload_f32(c, t->data.data(), D, &c->npre_b))) return rc;
    const std::vector<float> ones(D, 1.f);
    if ((rc = upload_f32(c, ones.data(), D, &c->ones))) return rc;
    if (c->Dout != D || find(c, "head.weight")) {
      if ((rc = need(c, "head.weight", {c->Dout, D}, &t))) return rc;
      if ((rc = upload_f32(c, t->data.data(), t->data.size(), &c->vhead_w))) return rc;
      if ((rc = c->dmalloc(&c->tok_tmp, (size_t)c->cfg.max_batch * c->T * D, true))) return rc;
    }
  } else if (c->Dout != D) {
    return fail(PIO_ERR_INVALID_ARG, "vit_out_dim != embed_dim needs vit_arch 1");
  }
  c->vl.resize(depth);
  for (int l = 0; l < depth; ++l) {
    VitLayerDev& L = c->vl[l];
    const std::string pre = "blocks." + std::to_string(l) + ".";
    struct F { const char* key; int64_t n; float** dst; };
    F fs[] = {{"norm1.weight", D, &L.n1w}, {"norm1.bias", D, &L.n1b}, {"attn.qkv.bias", 3 * D, &L.qkvb},
              {"attn.proj.bias", D, &L.projb}, {"ls1.gamma", D, &L.ls1}, {"norm2.weight", D, &L.n2w},
              {"norm2.bias", D, &L.n2b}, {"mlp.fc1.bias", 4 * D, &L.fc1b}, {"mlp.fc2.bias", D, &L.fc2b},
              {"ls2.gamma", D, &L.ls2}};
    for (auto& f : fs) {
      if (clip && (f.dst == &L.ls1 || f.dst == &L.ls2)) { *f.dst = c->ones; continue; }     // no LayerScale
      if ((rc = need(c, pre + f.key, {f.n}, &t))) return rc;
      if ((rc = upload_f32(c, t->data.data(), f.n, f.dst))) return rc;
    }
    struct Wm { const char* key; int64_t rows, cols; void** dst; };
    Wm ws[] = {{"attn.qkv.weight", 3 * D, D, &L.qkvw}, {"attn.proj.weight", D, D, &L.projw},
               {"mlp.fc1.weight", 4 * D, D, &L.fc1w}, {"mlp.fc2.weight", D, 4 * D, &L.fc2w}};
    for (auto& w : ws) {
      if ((rc = need(c, pre + w.key, {w.rows, w.cols}, &t))) return rc;
      const char* lsk = w.dst == &L.projw ? "ls1.gamma" : (w.dst == &L.fc2w ? "ls2.gamma" : nullptr);
      if (lsk == nullptr || clip) {
        if ((rc = upload_op(c, t->data.data(), w.rows, w.cols, w.cols, w.dst))) return rc;
        if (lsk != nullptr) (w.dst == &L.projw ? L.projb_ls : L.fc2b_ls) = (w.dst == &L.projw ? L.projb : L.fc2b);     // CLIP: no LayerScale
        continue;
      }
      // proj / fc2: x += ls * (A W^T + b) = A (ls W)^T + ls b -- LayerScale folded into the operands in fp32, rounded to the operand
      // type ONCE (the same relative rounding as W itself: the product's error is unchanged), so that the residual GEMM's sum can take
      // the old x in as one of its terms (kernels.h, EPI_RESIDUAL); the exact-fp32 parity mode keeps W, b and ls apart (vit_fp32.hip)
      const HostTensor *tg, *tb;
      if ((rc = need(c, pre + lsk, {w.rows}, &tg))) return rc;
      if ((rc = need(c, pre + (w.dst == &L.projw ? "attn.proj.bias" : "mlp.fc2.bias"), {w.rows}, &tb))) return rc;
      std::vector<float> wf((size_t)w.rows * w.cols), bf(w.rows);
      for (int64_t r = 0; r < w.rows; ++r) {
        const float gmm = tg->data[r];
        bf[r] = gmm * tb->data[r];
        for (int64_t k = 0; k < w.cols; ++k) wf[(size_t)r * w.cols + k] = gmm * t->data[(size_t)r * w.cols + k];
      }
      if ((rc = upload_op(c, wf.data(), w.rows, w.cols, w.cols, w.dst))) return rc;
      if ((rc = upload_f32(c, bf.data(), w.rows, w.dst == &L.projw ? &L.projb_ls : &L.fc2b_ls))) return rc;
    }
    if (c->vit_f32) {
      if ((int)c->vl32.size() != depth) c->vl32.resize(depth);
      float** d32[] = {&c->vl32[l].qkvw, &c->vl32[l].projw, &c->vl32[l].fc1w, &c->vl32[l].fc2w};
      for (int i = 0; i < 4; ++i) {
        if ((rc = need(c, pre + ws[i].key, {ws[i].rows, ws[i].cols}, &t))) return rc;
        if ((rc = upload_f32(c, t->data.data(), t->data.size(), d32[i]))) return rc;
      }
    }
  }
  if (c->vit_f32) {
    if ((rc = need(c, "patch_embed.proj.weight", {D, 3, p, p}, &t))) return rc;
    std::vector<float> pw((size_t)D * c->Kpad, 0.f);
    for (int r = 0; r < D; ++r) memcpy(&pw[(size_t)r * c->Kpad], &t->data[(size_t)r * c->Kpe], (size_t)c->Kpe * 4);
    if ((rc = upload_f32(c, pw.data(), pw.size(), &c->pe_w32))) return rc;
    const size_t Bm = c->cfg.max_batch, Mm = Bm * c->Tp;
    if ((rc = c->dmalloc(&c->f_ape, Bm * c->n2 * c->Kpad, true))) return rc;
    if ((rc = c->dmalloc(&c->f_emb, Bm * c->n2 * D, true))) return rc;
    if ((rc = c->dmalloc(&c->f_xn, Mm * D, true))) return rc;
    if ((rc = c->dmalloc(&c->f_qkv, Mm * 3 * D, true))) return rc;
    if ((rc = c->dmalloc(&c->f_ao, Mm * D, true))) return rc;
    if ((rc = c->dmalloc(&c->f_h, Mm * 4 * D, true))) return rc;
  }
  // workspaces
  const size_t B = c->cfg.max_batch, M = B * c->Tp;
  if ((rc = c->dmalloc(&c->x, M * D, true))) return rc;
  if ((rc = c->dmalloc_bytes(&c->xn, M * D * 2, true))) return rc;
  if ((rc = c->dmalloc_bytes(&c->ao, M * D * 2, true))) return rc;
  if ((rc = c->dmalloc_bytes(&c->hbuf, M * 4 * D * 2, true))) return rc;
  if ((rc = c->dmalloc_bytes(&c->ape, B * c->n2 * c->Kpad * 2, true))) return rc;
  const size_t qk = B * c->H * c->Tk * 64 * 2;
  if ((rc = c->dmalloc_bytes(&c->q, qk, true))) return rc;
  if ((rc = c->dmalloc_bytes(&c->k, qk, true))) return rc;
  if ((rc = c->dmalloc_bytes(&c->vT, qk, true))) return rc;
  const int Hr = c->cfg.readout_heads;
  if ((rc = c->dmalloc(&c->head_logits, B * Hr * c->n2))) return rc;
  if ((rc = c->dmalloc(&c->head_sm, B * Hr * c->n2))) return rc;
  {
    std::vector<int32_t> idx(B * Hr);
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int32_t)(i / Hr);
    if ((rc = c->dmalloc(&c->head_img, idx.size()))) return rc;
    HIP_OK(hipMemcpy(c->head_img, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  }
  c->has_vit = true;
  return PIO_OK;
}

// W [in][out] (Conv1D) or [out][in] (Linear) -> LayerNorm-folded [out][in] weight plus the two epilogue
// vectors of decoder.hip:  w'_jk = W_jk ln_w_k,  c_j = sum_k w'_jk,  d_j = sum_k ln_b_k W_jk + b_j.
int upload_ln_folded(pio_context* c, const float* W, bool in_major, int64_t in, int64_t out, const float* lnw,
                     const float* lnb, const float* bias, float** dw, float** dc, float** dd) {
  std::vector<float> w((size_t)in * out), cv(out), dv(out);
  for (int64_t j = 0; j < out; ++j) {
    double cs = 0.0, ds = bias ? (double)bias[j] : 0.0;
    for (int64_t k = 0; k < in; ++k) {
      const float wjk = in_major ? W[(size_t)k * out + j] : W[(size_t)j * in + k];
      const float f = wjk * lnw[k];
      w[(size_t)j * in + k] = f;
      cs += (double)f;
      ds += (double)lnb[k] * (double)wjk;
    }
    cv[j] = (float)cs;
    dv[j] = (float)ds;
  }
  int rc;
  if ((rc = upload_f32(c, w.data(), w.size(), dw))) return rc;
  if ((rc = upload_f32(c, cv.data(), cv.size(), dc))) return rc;
  return upload_f32(c, dv.data(), dv.size(), dd);
}

int alloc_decoder_workspaces(pio_context* c);

int finalize_decoder(pio_context* c) {
  const int E = c->cfg.dec_embd, V = c->cfg.dec_vocab, P = c->cfg.dec_positions, L = c->cfg.dec_layers;
  const int PS = c->cfg.prefix_size;
  if (E != 768) return fail(PIO_ERR_SHAPE, "decoder kernels are built for n_embd = 768 (the DeCap GPT-2 config)");
  const HostTensor *t, *tw, *tb, *tl, *tlb;
  int rc;
  if (find(c, "clip_project.model.0.weight")) {        // DeCap's prefix projection; a ViECap language model has none
    if ((rc = need(c, "clip_project.model.0.weight", {E, PS}, &t))) return rc;
    if ((rc = upload_f32(c, t->data.data(), (size_t)E * PS, &c->clip_w))) return rc;
    if ((rc = need(c, "clip_project.model.0.bias", {E}, &t))) return rc;
    if ((rc = upload_f32(c, t->data.data(), E, &c->clip_b))) return rc;
    c->has_clip_project = true;
  }
  if ((rc = need(c, "decoder.transformer.wte.weight", {V, E}, &tw))) return rc;
  if ((rc = upload_f32(c, tw->data.data(), (size_t)V * E, &c->wte))) return rc;
  if ((rc = need(c, "decoder.transformer.wpe.weight", {P, E}, &t))) return rc;
  if ((rc = upload_f32(c, t->data.data(), (size_t)P * E, &c->wpe))) return rc;
  // tied LM head with ln_f folded in (a second, scaled copy of wte)
  if ((rc = need(c, "decoder.transformer.ln_f.weight", {E}, &tl))) return rc;
  if ((rc = need(c, "decoder.transformer.ln_f.bias", {E}, &tlb))) return rc;
  if ((rc = upload_ln_folded(c, tw->data.data(), false, E, V, tl->data.data(), tlb->data.data(), nullptr, &c->head_w,
                             &c->head_c, &c->head_d))) return rc;
  {
    // fp16 copy of the LN-folded head for the arg-max filter (decoder.hip, "LM head with an fp16 filter"): scaled by a
    // power of two so that the largest weight sits near 2^14; the bound coefficient uses the largest row norm.
    const float* wt = tw->data.data();
    const float* lw = tl->data.data();
    double amax = 0.0, nmax = 0.0;
    for (int64_t j = 0; j < V; ++j) {
      double nn = 0.0;
      for (int64_t k = 0; k < E; ++k) {
        const double f = (double)(wt[(size_t)j * E + k] * lw[k]);
        nn += f * f;
        amax = std::max(amax, std::fabs(f));
      }
      nmax = std::max(nmax, nn);
    }
    int sh = 0;
    if (amax > 0.0) sh = 13 - std::ilogb(amax);                 // |W'| * 2^sh < 2^14
    const float up = std::ldexp(1.0f, sh);
    std::vector<uint16_t> w16((size_t)V * E);
    for (int64_t j = 0; j < V; ++j)
      for (int64_t k = 0; k < E; ++k) w16[(size_t)j * E + k] = f32_to_f16_bits((wt[(size_t)j * E + k] * lw[k]) * up);
    void* d16 = nullptr;
    if ((rc = c->dmalloc_bytes(&d16, w16.size() * 2))) return rc;
    HIP_OK(hipMemcpy(d16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    c->head_w16 = (uint16_t*)d16;
    c->head_w16_unscale = std::ldexp(1.0f, -sh);
    c->head_bound_coef = (float)(1.25e-3 * std::sqrt(nmax) * (1.0 + 1e-6));
  }
  c->dl.resize(L);
  for (int l = 0; l < L; ++l) {
    DecLayerW& w = c->dl[l];
    const std::string pre = "decoder.transformer.h." + std::to_string(l) + ".";
    float *dw, *dc, *dd;
    // ln_1 -> attn.c_attn
    if ((rc = need(c, pre + "attn.c_attn.weight", {E, 3 * E}, &tw))) return rc;
    if ((rc = need(c, pre + "attn.c_attn.bias", {3 * E}, &tb))) return rc;
    if ((rc = need(c, pre + "ln_1.weight", {E}, &tl))) return rc;
    if ((rc = need(c, pre + "ln_1.bias", {E}, &tlb))) return rc;
    if ((rc = upload_ln_folded(c, tw->data.data(), true, E, 3 * E, tl->data.data(), tlb->data.data(), tb->data.data(),
                               &dw, &dc, &dd))) return rc;
    w.attn_w = dw; w.attn_c = dc; w.attn_d = dd;
    // ln_2 -> mlp.c_fc
    if ((rc = need(c, pre + "mlp.c_fc.weight", {E, 4 * E}, &tw))) return rc;
    if ((rc = need(c, pre + "mlp.c_fc.bias", {4 * E}, &tb))) return rc;
    if ((rc = need(c, pre + "ln_2.weight", {E}, &tl))) return rc;
    if ((rc = need(c, pre + "ln_2.bias", {E}, &tlb))) return rc;
    if ((rc = upload_ln_folded(c, tw->data.data(), true, E, 4 * E, tl->data.data(), tlb->data.data(), tb->data.data(),
                               &dw, &dc, &dd))) return rc;
    w.fc_w = dw; w.fc_c = dc; w.fc_d = dd;
    // residual-branch outputs: plain [out][in] transposes
    if ((rc = need(c, pre + "attn.c_proj.weight", {E, E}, &tw))) return rc;
    if ((rc = upload_transposed(c, tw->data.data(), E, E, &dw))) return rc;
    w.proj_w = dw;
    if ((rc = need(c, pre + "attn.c_proj.bias", {E}, &tb))) return rc;
    if ((rc = upload_f32(c, tb->data.data(), E, &dw))) return rc;
    w.proj_b = dw;
    if ((rc = need(c, pre + "mlp.c_proj.weight", {4 * E, E}, &tw))) return rc;
    if ((rc = upload_transposed(c, tw->data.data(), 4 * E, E, &dw))) return rc;
    w.fc2_w = dw;
    if ((rc = need(c, pre + "mlp.c_proj.bias", {E}, &tb))) return rc;
    if ((rc = upload_f32(c, tb->data.data(), E, &dw))) return rc;
    w.fc2_b = dw;
  }
  // split-fp16 copies of the three wide layer matrices for decodes of more than 64 prefixes (decoder.hip: k_dec_gemm_s; same bytes as
  // the fp32 ones; PIO_DEC_SPLIT=0 keeps the fp32 kernels everywhere)
  {
    const char* ev = getenv("PIO_DEC_SPLIT");
    if (E == 768 && !(ev && ev[0] == '0')) {
      for (int l = 0; l < L; ++l) {
        DecLayerW& w = c->dl[l];
        struct { const float* src; size_t n; const void** dst; float* un; } m[3] = {
            {w.attn_w, (size_t)3 * E * E, &w.attn_ws, &w.attn_un}, {w.fc_w, (size_t)4 * E * E, &w.fc_ws, &w.fc_un},
            {w.fc2_w, (size_t)4 * E * E, &w.fc2_ws, &w.fc2_un}};
        for (auto& e : m) {
          void* d = nullptr;
          if ((rc = c->dmalloc_bytes(&d, e.n * 4))) return rc;
          float un = 0.f;
          if (dec_split_weights(e.src, e.n, d, &un, nullptr) != hipSuccess) continue;    // non-finite weights: the fp32 kernels serve this matrix
          *e.dst = d; *e.un = un;
        }
      }
    }
  }
  HIP_OK(decoder_init());
  if ((rc = alloc_decoder_workspaces(c))) return rc;
  c->has_dec = true;
  return PIO_OK;
}

// the decoder's per-handle state: activations, KV caches, split-K slabs, head scratch, staging buffers
int alloc_decoder_workspaces(pio_context* c) {
  int rc;
  const size_t E = c->cfg.dec_embd, V = c->cfg.dec_vocab, L = c->cfg.dec_layers, PS = c->cfg.prefix_size;
  const size_t N = c->cfg.max_prefixes, S = c->cfg.max_steps;
  if ((rc = c->dmalloc(&c->dx, N * E, true))) return rc;
  if ((rc = c->dmalloc(&c->dqkv, N * 3 * E, true))) return rc;
  if ((rc = c->dmalloc(&c->datt, N * E, true))) return rc;
  if ((rc = c->dmalloc(&c->dhid, N * 4 * E, true))) return rc;
  if ((rc = c->dmalloc(&c->splitk_ws, (size_t)DEC_SPLITK_COUNTERS * 4 * 8 * 256, true))) return rc;   // tiles x 4 k-slices x 8 (column, row) group pairs
  if ((rc = c->dmalloc(&c->splitk_cnt, DEC_TICKET_WORDS, true))) return rc;
  if ((rc = c->dmalloc(&c->kcache, (size_t)L * N * S * E, true))) return rc;
  if ((rc = c->dmalloc(&c->vcache, (size_t)L * N * S * E, true))) return rc;
  if ((rc = c->dmalloc(&c->logits, N * (size_t)round_up(V, 64), true))) return rc;
  if ((rc = c->dmalloc_bytes(&c->dec_xh, N * E * 2, true))) return rc;
  if ((rc = c->dmalloc(&c->lm_stats, N * 4, true))) return rc;
  if ((rc = c->dmalloc(&c->lm_gmax, N * (size_t)round_up((V + 15) / 16, 64), true))) return rc;
  if ((rc = c->dmalloc(&c->prefix_buf, N * PS, true))) return rc;
  if ((rc = c->dmalloc(&c->logprob_buf, N * S, true))) return rc;
  if ((rc = c->dmalloc(&c->ids_buf, N * S, true))) return rc;
  return PIO_OK;
}

// MappingNetwork of the ViECap head (P/src/viecap/ClipCap.py:122-153); 8 heads (ClipCaptionModel's default, :171)
int finalize_viecap_map(pio_context* c) {
  const int E = c->cfg.dec_embd;
  const HostTensor* t;
  int rc;
  const HostTensor* lw = find(c, "mapping_network.linear.weight");
  const HostTensor* pc = find(c, "mapping_network.prefix_const");
  if (!lw || !pc || lw->shape.size() != 2 || pc->shape.size() != 2 || pc->shape[1] != E || lw->shape[0] % E != 0)
    return fail(PIO_ERR_SHAPE, "mapping_network.linear.weight [Lp*E, C] / prefix_const [Lc, E] missing or mis-shaped");
  c->map_C = (int)lw->shape[1]; c->map_Lp = (int)(lw->shape[0] / E); c->map_Lc = (int)pc->shape[0];
  if (c->map_C % 32 != 0 || c->map_Lp + c->map_Lc > 32) return fail(PIO_ERR_SHAPE, "mapping network: C % 32 != 0 or more than 32 tokens");
  if ((rc = upload_f32(c, lw->data.data(), lw->data.size(), &c->map_lin_w))) return rc;
  if ((rc = need(c, "mapping_network.linear.bias", {(int64_t)c->map_Lp * E}, &t))) return rc;
  if ((rc = upload_f32(c, t->data.data(), t->data.size(), &c->map_lin_b))) return rc;
  if ((rc = upload_f32(c, pc->data.data(), pc->data.size(), &c->map_prefix))) return rc;
  int L = 0;
  while (find(c, "mapping_network.transformer.layers." + std::to_string(L) + ".norm1.weight")) ++L;
  if (L < 1) return fail(PIO_ERR_SHAPE, "mapping network without transformer layers");
  const HostTensor* f1 = find(c, "mapping_network.transformer.layers.0.mlp.fc1.weight");
  if (!f1 || f1->shape.size() != 2) return fail(PIO_ERR_SHAPE, "mapping network: mlp.fc1.weight missing");
  const int H = (int)f1->shape[0];
  if (H % 32 != 0) return fail(PIO_ERR_SHAPE, "mapping network: hidden size % 32 != 0");
  c->map_layers = L; c->map_hidden = H;
  c->ml.resize(L);
  for (int l = 0; l < L; ++l) {
    const std::string pre = "mapping_network.transformer.layers." + std::to_string(l) + ".";
    ViecapMapLayerW& w = c->ml[l];
    struct F { const char* key; std::vector<int64_t> shape; const float** dst; };
    F fs[] = {{"norm1.weight", {E}, &w.n1w}, {"norm1.bias", {E}, &w.n1b}, {"attn.to_queries.weight", {E, E}, &w.q_w},
              {"attn.to_keys_values.weight", {2 * E, E}, &w.kv_w}, {"attn.project.weight", {E, E}, &w.proj_w},
              {"attn.project.bias", {E}, &w.proj_b}, {"norm2.weight", {E}, &w.n2w}, {"norm2.bias", {E}, &w.n2b},
              {"mlp.fc1.weight", {H, E}, &w.fc1_w}, {"mlp.fc1.bias", {H}, &w.fc1_b}, {"mlp.fc2.weight", {E, H}, &w.fc2_w},
              {"mlp.fc2.bias", {E}, &w.fc2_b}};
    for (auto& f : fs) {
      if ((rc = need(c, pre + f.key, f.shape, &t))) return rc;
      float* d = nullptr;
      if ((rc = upload_f32(c, t->data.data(), t->data.size(), &d))) return rc;
      *f.dst = d;
    }
    if (find(c, pre + "attn.to_queries.bias")) return fail(PIO_ERR_SHAPE, "mapping network with biased q / kv projections (the reference builds them bias-free)");
  }
  const size_t N = c->cfg.max_prefixes, M = N * (c->map_Lp + c->map_Lc);
  if ((rc = c->dmalloc(&c->map_lin, N * c->map_Lp * E))) return rc;
  if ((rc = c->dmalloc(&c->map_x, M * E))) return rc;
  if ((rc = c->dmalloc(&c->map_ln, M * E))) return rc;
  if ((rc = c->dmalloc(&c->map_q, M * E))) return rc;
  if ((rc = c->dmalloc(&c->map_kv, M * 2 * E))) return rc;
  if ((rc = c->dmalloc(&c->map_att, M * E))) return rc;
  if ((rc = c->dmalloc(&c->map_hid, M * H))) return rc;
  c->has_map = true;
  return PIO_OK;
}

bool is_ignorable_key(const std::string& k) {
  auto ends = [&](const char* s) { size_t n = strlen(s); return k.size() >= n && k.compare(k.size() - n, n, s) == 0; };
  return k == "mask_token" || k == "decoder.lm_head.weight" || ends(".attn.bias") || ends(".attn.masked_bias");
}

bool is_known_key(const std::string& k) {
  static const char* exact[] = {"cls_token", "pos_embed", "register_tokens", "patch_embed.proj.weight",
                                "patch_embed.proj.bias", "norm.weight", "norm.bias", "norm_pre.weight", "norm_pre.bias", "head.weight", "talk2dino.A_pinv",
                                "talk2dino.b"};
  for (auto e : exact) if (k == e) return true;
  return k.rfind("blocks.", 0) == 0 || k.rfind("decoder.transformer.", 0) == 0 || k.rfind("clip_project.model.0.", 0) == 0 ||
         k.rfind("mapping_network.", 0) == 0;
}

// ---- Pillow's resampling coefficients (src/libImaging/Resample.c: bicubic_filter, precompute_coeffs,
//      normalize_coeffs_8bpc), box = the whole axis, for the output samples [first, first + count) only.
//      Double arithmetic in Pillow's operation order, no FMA contraction: the int tables must be Pillow's bit for bit.
#pragma clang fp contract(off)
double pil_bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

struct PilAxis { int ksize = 0; std::vector<int32_t> kk, bounds; };

void pil_axis_table(int in_size, int out_size, int first, int count, PilAxis& t) {
  const double in0 = 0.0, in1 = (double)(float)in_size;
  const double scale = (in1 - in0) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  t.ksize = (int)ceil(support) * 2 + 1;
  t.kk.assign((size_t)count * t.ksize, 0);
  t.bounds.assign((size_t)count * 2, 0);
  const double ss = 1.0 / filterscale;
  std::vector<double> w((size_t)t.ksize + 2);
  for (int i = 0; i < count; ++i) {
    const int xx = first + i;
    const double center = in0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      w[x] = pil_bicubic_filter((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
      double k = w[x];
      if (ww != 0.0) k /= ww;
      t.kk[(size_t)i * t.ksize + x] = k < 0 ? (int)(-0.5 + k * (1 << 22)) : (int)(0.5 + k * (1 << 22));
    }
    t.bounds[2 * i] = xmin;
    t.bounds[2 * i + 1] = xmax;
  }
}

// torchvision F.center_crop: origin of the crop window in the coordinates of the (w, h) image (negative = zero padding)
void center_crop_origin(int w, int h, int crop, int* left, int* top) {
  const int pl = crop > w ? (crop - w) / 2 : 0, pt = crop > h ? (crop - h) / 2 : 0;
  const int pw = w + pl + (crop > w ? (crop - w + 1) / 2 : 0), ph = h + pt + (crop > h ? (crop - h + 1) / 2 : 0);
  if (pw == crop && ph == crop) { *left = -pl; *top = -pt; return; }
  *left = (int)nearbyint((pw - crop) / 2.0) - pl;   // Python round(): half to even, as nearbyint in the default mode
  *top = (int)nearbyint((ph - crop) / 2.0) - pt;
}

}  // namespace

extern "C" {

const char* pio_last_error(void) { return g_err.c_str(); }
const char* pio_version(void) { return "patchioner_hip 0.1.0 (gfx950)"; }

int pio_stream_create(int32_t device, int32_t skip_cus, int32_t n_cus, void** stream) {
  if (!stream || skip_cus < 0) return fail(PIO_ERR_INVALID_ARG, "pio_stream_create: bad argument");
  HIP_OK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, device));
  const int total = prop.multiProcessorCount;
  hipStream_t s = nullptr;
  if (n_cus <= 0 || (skip_cus == 0 && n_cus >= total)) {
    HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  } else {
    if (skip_cus + n_cus > total) return fail(PIO_ERR_INVALID_ARG, "pio_stream_create: CU window exceeds the device");
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int i = skip_cus; i < skip_cus + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    HIP_OK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  }
  *stream = (void*)s;
  return PIO_OK;
}

int pio_stream_destroy(void* stream) {
  if (stream) {
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));      // nothing of the caller's may still be queued on it
    HIP_OK(hipStreamDestroy((hipStream_t)stream));
  }
  return PIO_OK;
}

int pio_create(const pio_config* cfg, pio_handle* out) {
  if (!cfg || !out) return fail(PIO_ERR_INVALID_ARG, "pio_create: null argument");
  if (cfg->embed_dim % 64 != 0 || cfg->embed_dim / 64 != cfg->num_heads)
    return fail(PIO_ERR_INVALID_ARG, "pio_create: backbone head_dim must be 64 (embed_dim = 64 * num_heads)");
  if (cfg->crop_dim % cfg->patch_size != 0)
    return fail(PIO_ERR_INVALID_ARG, "pio_create: crop_dim must be a multiple of patch_size");
  if (cfg->max_batch < 1 || cfg->max_prefixes < 1 || cfg->max_prefixes > DEC_MAX_PREFIXES || cfg->max_steps < 1 || cfg->max_steps > 256)
    return fail(PIO_ERR_INVALID_ARG, "pio_create: capacities out of range (prefixes <= 256, decoder positions <= 256)");
  if (cfg->readout_heads != 16 && cfg->readout_heads * 64 != cfg->embed_dim)
    return fail(PIO_ERR_INVALID_ARG, "pio_create: readout_heads must be 16, or embed_dim / 64 (ViT-S: 6)");
  if (cfg->vit_arch != 0 && cfg->vit_arch != 1) return fail(PIO_ERR_INVALID_ARG, "pio_create: vit_arch must be 0 (DINOv2) or 1 (OpenAI-CLIP ViT)");
  if (cfg->vit_arch == 1 && cfg->num_registers != 0)
    return fail(PIO_ERR_INVALID_ARG, "pio_create: the CLIP ViT has no registers");
  if (cfg->vit_out_dim < 0 || (cfg->vit_out_dim > 0 && cfg->vit_out_dim % 32 != 0) || (cfg->vit_arch == 0 && cfg->vit_out_dim != 0 && cfg->vit_out_dim != cfg->embed_dim))
    return fail(PIO_ERR_INVALID_ARG, "pio_create: vit_out_dim must be 0, or a multiple of 32 with vit_arch 1");
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(PIO_ERR_INVALID_ARG, "pio_create: no such HIP device");
  HIP_OK(hipSetDevice(cfg->device));
  pio_context* c = new pio_context();
  c->cfg = *cfg;
  c->D = cfg->embed_dim;
  c->H = cfg->num_heads;
  c->n = cfg->crop_dim / cfg->patch_size;
  c->n2 = c->n * c->n;
  c->G = 1 + cfg->num_registers;
  c->T = c->G + c->n2;
  c->Tp = round_up(c->T, 8);
  c->Tk = round_up(c->T, 64);
  c->Kpe = 3 * cfg->patch_size * cfg->patch_size;
  c->Kpad = round_up(c->Kpe, 64);
  c->op = cfg->vit_operand_type == 1 ? OP_BF16 : OP_F16;
  c->vit_f32 = cfg->vit_operand_type == 2;
  c->Dout = cfg->vit_out_dim > 0 ? cfg->vit_out_dim : cfg->embed_dim;
  const char* ng = getenv("PIO_NO_GRAPH");
  c->use_graph = !(ng && ng[0] == '1');
  const char* pf = getenv("PIO_DEC_PREFILL");
  c->batched_prefill = !(pf && pf[0] == '0');
  const char* lt = getenv("PIO_LM_TAIL");
  c->lm_tail = lt && lt[0] == '1';
  hipError_t e = hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return fail(PIO_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
  *out = c;
  return PIO_OK;
}

int pio_clone_decoder(pio_handle src, pio_handle* out) {
  if (!src || !out) return fail(PIO_ERR_INVALID_ARG, "pio_clone_decoder: null argument");
  if (!src->finalized || !src->has_dec) return fail(PIO_ERR_NOT_READY, "pio_clone_decoder: decoder weights not finalized");
  if (src->parent) return fail(PIO_ERR_INVALID_ARG, "pio_clone_decoder: clone the owner of the weights, not a clone");
  HIP_OK(hipSetDevice(src->cfg.device));
  pio_context* c = new pio_context();
  c->cfg = src->cfg;
  c->n = src->n; c->n2 = src->n2; c->T = src->T; c->Tp = src->Tp; c->Tk = src->Tk; c->G = src->G; c->D = src->D;
  c->Kpe = src->Kpe; c->Kpad = src->Kpad; c->H = src->H; c->op = src->op; c->Dout = src->Dout;
  c->use_graph = src->use_graph;
  c->batched_prefill = src->batched_prefill;
  c->lm_tail = src->lm_tail;
  // borrowed, read-only: the decoder's weights (freed by the owner only)
  c->clip_w = src->clip_w; c->clip_b = src->clip_b; c->wte = src->wte; c->wpe = src->wpe;
  c->head_w = src->head_w; c->head_c = src->head_c; c->head_d = src->head_d;
  c->head_w16 = src->head_w16; c->head_w16_unscale = src->head_w16_unscale; c->head_bound_coef = src->head_bound_coef;
  c->dl = src->dl;
  c->has_clip_project = src->has_clip_project;
  hipError_t e = hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return fail(PIO_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
  const int rc = alloc_decoder_workspaces(c);
  if (rc != PIO_OK) {
    for (void* p : c->allocs) (void)hipFree(p);
    (void)hipStreamDestroy(c->capture_stream);
    delete c;
    return rc;
  }
  {
    const hipError_t se = hipDeviceSynchronize();   // the zero-fills above are done before any stream touches the new workspaces
    if (se != hipSuccess) {
      for (void* p : c->allocs) (void)hipFree(p);
      (void)hipStreamDestroy(c->capture_stream);
      delete c;
      return fail(PIO_ERR_HIP, std::string("pio_clone_decoder: ") + hipGetErrorString(se));
    }
  }
  c->has_dec = true;
  c->finalized = true;
  c->parent = src;
  src->clones += 1;
  *out = c;
  return PIO_OK;
}

int pio_destroy(pio_handle c) {
  if (!c) return PIO_OK;
  if (c->clones > 0) return fail(PIO_ERR_INVALID_ARG, "pio_destroy: decoder clones of this handle are still alive");
  if (c->parent) c->parent->clones -= 1;
  (void)hipSetDevice(c->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.second);
  for (auto& g : c->pgraphs) { (void)hipGraphExecDestroy(g.second.exec); if (g.second.done) (void)hipEventDestroy(g.second.done); }
  for (void* p : c->allocs) (void)hipFree(p);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (auto& sl : c->prep_slots) {
    if (sl.host) (void)hipHostFree(sl.host);
    if (sl.dev) (void)hipFree(sl.dev);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  if (c->prep_tmp) (void)hipFree(c->prep_tmp);
  if (c->prep_lut) (void)hipFree(c->prep_lut);
  if (c->capture_stream) (void)hipStreamDestroy(c->capture_stream);
  delete c;
  return PIO_OK;
}

int pio_load_weight(pio_handle c, const char* key, const float* host_data, const int64_t* shape, int32_t ndim) {
  if (!c || !key || !host_data || (ndim > 0 && !shape)) return fail(PIO_ERR_INVALID_ARG, "pio_load_weight: null argument");
  if (c->finalized) return fail(PIO_ERR_INVALID_ARG, "pio_load_weight: weights already finalized");
  std::string k(key);
  // a ViECap checkpoint (ClipCaptionModel.state_dict(), P/src/viecap/ClipCap.py:155-200) names its language model
  // `gpt.*`: the same GPT2LMHeadModel keys the DeCap checkpoint has under `decoder.*`
  if (k.rfind("gpt.", 0) == 0) k = "decoder." + k.substr(4);
  if (is_ignorable_key(k)) return PIO_OK;
  if (!is_known_key(k)) return fail(PIO_ERR_UNKNOWN_WEIGHT, "unknown weight key '" + k + "'");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(host_data, host_data + t.numel());
  c->host[k] = std::move(t);
  return PIO_OK;
}

int pio_finalize_weights(pio_handle c) {
  if (!c) return fail(PIO_ERR_INVALID_ARG, "null handle");
  if (c->finalized) return fail(PIO_ERR_INVALID_ARG, "pio_finalize_weights: called twice");
  HIP_OK(hipSetDevice(c->cfg.device));
  int rc;
  if (find(c, "cls_token")) {
    if ((rc = finalize_vit(c))) return rc;
  }
  if (find(c, "decoder.transformer.wte.weight")) {
    if ((rc = finalize_decoder(c))) return rc;
  }
  if (find(c, "mapping_network.linear.weight")) {
    if ((rc = finalize_viecap_map(c))) return rc;
  }
  if (find(c, "talk2dino.A_pinv")) {
    const HostTensor* t;
    if ((rc = need(c, "talk2dino.A_pinv", {c->cfg.prefix_size, c->D}, &t))) return rc;
    if ((rc = upload_f32(c, t->data.data(), t->data.size(), &c->A_pinv))) return rc;
    if ((rc = need(c, "talk2dino.b", {c->D}, &t))) return rc;
    if ((rc = upload_f32(c, t->data.data(), t->data.size(), &c->inv_b))) return rc;
    c->has_inv = true;
  }
  c->host.clear();
  c->finalized = true;
  HIP_OK(hipDeviceSynchronize());
  return PIO_OK;
}

static int bank_common(pio_context* c, int64_t rows, int32_t dim) {
  c->bank_rows = rows;
  c->bank_dim = dim;
  int rc;
  if ((rc = c->dmalloc(&c->bank_inv, (size_t)rows))) return rc;
  HIP_OK(launch_row_inv_norm(c->bank, rows, dim, c->bank_inv, nullptr));
  // the scale of the split-fp16 GEMM2: the power of two that brings the bank's largest magnitude into (2^14, 2^15].  Not in the
  // fp32 parity mode (vit_operand_type 2: every stage exact), not for a bank with a non-finite or vanishing maximum.
  c->bank_scale = 0.f;
  if (c->cfg.vit_operand_type != 2 && getenv("PIO_PROJECT_EXACT") == nullptr) {
    uint32_t* d_max = nullptr;
    HIP_OK(hipMalloc(&d_max, 4));
    uint32_t bits = 0;
    hipError_t e = launch_abs_max(c->bank, rows * dim, d_max, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&bits, d_max, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_max);
    HIP_OK(e);
    float amax;
    memcpy(&amax, &bits, 4);
    int ex = 0;
    if (bits < 0x7F800000u && amax > 0.f) {
      (void)std::frexp(amax, &ex);                 // amax = m 2^ex, m in [0.5, 1)  =>  amax 2^(15 - ex) <= 2^15
      if (ex > -80 && ex < 80) c->bank_scale = std::ldexp(1.0f, 15 - ex);
    }
    if (c->bank_scale > 0.f) {
      if ((rc = c->dmalloc(&c->bank_split, (size_t)rows * dim))) return rc;      // rows * 2 * dim fp16
      HIP_OK(launch_split_bank(c->bank, rows, dim, c->bank_scale, c->bank_split, nullptr));
    }
  }
  if ((rc = c->dmalloc(&c->part_acc, (size_t)c->parts * 24 * dim))) return rc;      // 24: 48 queries x parts / 2 workgroups (project.hip)
  if ((rc = c->dmalloc(&c->part_ml, (size_t)c->parts * 24 * 2))) return rc;
  if ((rc = c->dmalloc(&c->sims, (size_t)16 * rows))) return rc;
  HIP_OK(hipDeviceSynchronize());
  return PIO_OK;
}

int pio_set_memory_bank(pio_handle c, const float* host_bank, int64_t rows, int32_t dim, int64_t* rows_kept) {
  if (!c || !host_bank || rows < 1) return fail(PIO_ERR_INVALID_ARG, "pio_set_memory_bank: bad argument");
  if (dim != 384 && dim != 512 && dim != 768) return fail(PIO_ERR_SHAPE, "memory bank dim must be 384, 512 or 768");
  if (c->bank) return fail(PIO_ERR_INVALID_ARG, "memory bank already set");
  HIP_OK(hipSetDevice(c->cfg.device));
  // the reference drops rows whose norm is 0 when it loads the bank (im2txtprojection.py:343-345)
  std::vector<int64_t> keep;
  keep.reserve(rows);
  for (int64_t r = 0; r < rows; ++r) {
    const float* p = host_bank + r * dim;
    // the reference's predicate is on the fp32 NORM (embs.norm(dim=-1) != 0), not on the elements: a row of tiny values
    // whose squares underflow is dropped too, exactly as on the device path (engine.py: bank.norm(dim=-1) != 0)
    float ss = 0.f;
    for (int d = 0; d < dim; ++d) ss += p[d] * p[d];
    if (std::sqrt(ss) != 0.f) keep.push_back(r);
  }
  const int64_t kept = (int64_t)keep.size();
  if (kept == 0) return fail(PIO_ERR_SHAPE, "memory bank has no non-zero row");
  int rc;
  if ((rc = c->dmalloc(&c->bank, (size_t)kept * dim))) return rc;
  if (kept == rows) {
    HIP_OK(hipMemcpy(c->bank, host_bank, (size_t)rows * dim * 4, hipMemcpyHostToDevice));
  } else {
    int64_t run_start = 0;   // copy maximal runs of consecutive kept rows
    while (run_start < kept) {
      int64_t run_end = run_start + 1;
      while (run_end < kept && keep[run_end] == keep[run_end - 1] + 1) ++run_end;
      HIP_OK(hipMemcpy(c->bank + run_start * dim, host_bank + keep[run_start] * dim,
                       (size_t)(run_end - run_start) * dim * 4, hipMemcpyHostToDevice));
      run_start = run_end;
    }
  }
  if (rows_kept) *rows_kept = kept;
  return bank_common(c, kept, dim);
}

int pio_set_memory_bank_device(pio_handle c, const float* dev_bank, int64_t rows, int32_t dim) {
  if (!c || !dev_bank || rows < 1) return fail(PIO_ERR_INVALID_ARG, "pio_set_memory_bank_device: bad argument");
  if (dim != 384 && dim != 512 && dim != 768) return fail(PIO_ERR_SHAPE, "memory bank dim must be 384, 512 or 768");
  if (c->bank) return fail(PIO_ERR_INVALID_ARG, "memory bank already set");
  HIP_OK(hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = c->dmalloc(&c->bank, (size_t)rows * dim))) return rc;
  HIP_OK(hipMemcpy(c->bank, dev_bank, (size_t)rows * dim * 4, hipMemcpyDeviceToDevice));
  return bank_common(c, rows, dim);
}

// The same block in exact fp32 (vit_fp32.hip; parity mode): LN -> qkv -> attention -> proj -> x += ls1 * branch -> LN -> fc1 ->
// GELU -> fc2 -> x += ls2 * branch.  qkv_last: the fused-QKV output of the T valid rows of every image (the hook's tensor).
static int run_vit_block_f32(pio_handle c, int l, int B, float* qkv_last, const int32_t* lens, hipStream_t s) {
  const VitLayerDev& L = c->vl[l];
  const pio_context::VitLayerF32& W = c->vl32[l];
  const int D = c->D, M = B * c->Tp;
  const float eps = c->cfg.vit_ln_eps;
  HIP_OK(launch_layernorm_f32(c->x, L.n1w, L.n1b, eps, M, D, c->f_xn, s));
  HIP_OK(launch_sgemm_tn(c->f_xn, D, W.qkvw, D, L.qkvb, 1.f, c->f_qkv, 3 * D, M, 3 * D, D, 0, 0, s));
  if (qkv_last)
    for (int b = 0; b < B; ++b)
      HIP_OK(hipMemcpyAsync(qkv_last + (size_t)b * c->T * 3 * D, c->f_qkv + (size_t)b * c->Tp * 3 * D, (size_t)c->T * 3 * D * 4,
                            hipMemcpyDeviceToDevice, s));
  HIP_OK(launch_attention_f32(c->f_qkv, B, c->H, c->T, c->Tp, D, 0.125f, lens, c->f_ao, s));
  HIP_OK(launch_sgemm_tn(c->f_ao, D, W.projw, D, L.projb, 1.f, c->f_xn, D, M, D, D, 0, 0, s));
  HIP_OK(launch_resid_ls_f32(c->x, c->f_xn, L.ls1, (size_t)M * D, D, s));
  HIP_OK(launch_layernorm_f32(c->x, L.n2w, L.n2b, eps, M, D, c->f_xn, s));
  HIP_OK(launch_sgemm_tn(c->f_xn, D, W.fc1w, D, L.fc1b, 1.f, c->f_h, 4 * D, M, 4 * D, D, 0, 0, s));
  HIP_OK(launch_gelu_f32(c->f_h, (size_t)M * 4 * D, c->cfg.vit_arch == 1 ? 1 : 0, s));
  HIP_OK(launch_sgemm_tn(c->f_h, 4 * D, W.fc2w, 4 * D, L.fc2b, 1.f, c->f_xn, D, M, D, 4 * D, 0, 0, s));
  HIP_OK(launch_resid_ls_f32(c->x, c->f_xn, L.ls2, (size_t)M * D, D, s));
  return PIO_OK;
}

// One pre-LN DINOv2 block on the B sequences of c->x (in place): LN1 -> qkv -> attention -> proj (+LayerScale,
// +residual) -> LN2 -> fc1 + GELU -> fc2 (+LayerScale, +residual).  `at.lens` (optional) = per-sequence token counts.
static int run_vit_block(pio_handle c, const VitLayerDev& L, int B, const GemmArgs& g, const VitAttnArgs& at,
                         float* qkv_last, hipStream_t s, const VitLayerDev* next = nullptr) {
  const int D = c->D, M = B * c->Tp;
  const double Malg = (double)B * c->T;   // algorithmic rows: no pad tokens
  // TIMING ablation, compiled only into diagnostic builds (-DPIO_ABLATIONS; tools/microbench/ln_ablation.sh): PIO_ABL_SKIP_LN=1 leaves
  // the two LayerNorm launches of every block out (wrong results) -- the upper bound of what ANY folding of LayerNorm into its
  // neighbours could save: +9.8 % captions/s through the pipeline, +3.9 % on a synchronous forward (profiles/r04_bench_sweep.log)
#ifdef PIO_ABLATIONS
  static const bool skip_ln = getenv("PIO_ABL_SKIP_LN") != nullptr && getenv("PIO_ABL_SKIP_LN")[0] == '1';
#else
  constexpr bool skip_ln = false;
#endif
  if (!skip_ln)
  PROF(c, PIO_PROF_VIT_LN, 0, Malg * D * 6.0, s,
       launch_layernorm(c->op, c->x, L.n1w, L.n1b, c->cfg.vit_ln_eps, M, D, c->xn, nullptr, c->T, c->Tp, s));
  {
    GemmArgs a = g;
    a.A = c->xn; a.lda = D; a.W = L.qkvw; a.bias = L.qkvb; a.M = M; a.N = 3 * D; a.K = D;
    a.qkv_last = qkv_last;
    a.pf = L.projw; a.pf_bytes = D * D * 2;          // each GEMM's spare workgroups warm the next one's weights (GemmArgs::pf)
    PROF(c, PIO_PROF_VIT_GEMM, 2.0 * Malg * 3.0 * D * D, 0, s, launch_vit_gemm(c->op, EPI_QKV, a, s));
  }
  PROF(c, PIO_PROF_VIT_ATTN, 4.0 * B * (double)c->T * c->T * D, 0, s, launch_vit_attention(c->op, at, s));
  {
    GemmArgs a = g;
    a.A = c->ao; a.lda = D; a.W = L.projw; a.bias = L.projb_ls; a.M = M; a.N = D; a.K = D;
    a.pf = L.fc1w; a.pf_bytes = 4 * D * D * 2;
    PROF(c, PIO_PROF_VIT_GEMM, 2.0 * Malg * D * D, 0, s, launch_vit_gemm(c->op, EPI_RESIDUAL, a, s));
  }
  if (!skip_ln)
  PROF(c, PIO_PROF_VIT_LN, 0, Malg * D * 6.0, s,
       launch_layernorm(c->op, c->x, L.n2w, L.n2b, c->cfg.vit_ln_eps, M, D, c->xn, nullptr, c->T, c->Tp, s));
  {
    GemmArgs a = g;
    a.A = c->xn; a.lda = D; a.W = L.fc1w; a.bias = L.fc1b; a.out16 = c->hbuf; a.M = M; a.N = 4 * D; a.K = D;
    a.act = c->cfg.vit_arch == 1 ? 1 : 0;
    a.pf = L.fc2w; a.pf_bytes = 4 * D * D * 2;
    PROF(c, PIO_PROF_VIT_GEMM, 2.0 * Malg * 4.0 * D * D, 0, s, launch_vit_gemm(c->op, EPI_GELU, a, s));
  }
  {
    GemmArgs a = g;
    a.A = c->hbuf; a.lda = 4 * D; a.W = L.fc2w; a.bias = L.fc2b_ls; a.M = M; a.N = D; a.K = 4 * D;
    if (next != nullptr) { a.pf = next->qkvw; a.pf_bytes = 3 * D * D * 2; }
    PROF(c, PIO_PROF_VIT_GEMM, 2.0 * Malg * 4.0 * D * D, 0, s, launch_vit_gemm(c->op, EPI_RESIDUAL, a, s));
  }
  return PIO_OK;
}

int pio_vit_forward(pio_handle c, const float* imgs, int32_t B, float* tokens, float* qkv_last, pio_stream stream) {
  if (!c || !imgs || !tokens) return fail(PIO_ERR_INVALID_ARG, "pio_vit_forward: null argument");
  if (c->cfg.vit_arch == 1 && qkv_last) return fail(PIO_ERR_INVALID_ARG, "pio_vit_forward: the CLIP ViT has no qkv capture (no hook in the reference, P/src/model.py:586-590 is DINOv2 only)");
  if (!c->has_vit) return fail(PIO_ERR_NOT_READY, "pio_vit_forward: backbone weights not loaded");
  if (B < 1 || B > c->cfg.max_batch) return fail(PIO_ERR_CAPACITY, "pio_vit_forward: batch above max_batch");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const int D = c->D, M = B * c->Tp;
  if (c->vit_f32) {
    HIP_OK(launch_im2col_f32(imgs, B, c->cfg.crop_dim, c->cfg.patch_size, c->n, c->Kpad, c->f_ape, s));
    HIP_OK(launch_token_init(c->x, c->cls, c->pos, c->reg, B, c->cfg.num_registers, c->T, c->Tp, D, s));
    HIP_OK(launch_sgemm_tn(c->f_ape, c->Kpad, c->pe_w32, c->Kpad, c->pe_b, 1.f, c->f_emb, D, B * c->n2, D, c->Kpad, 0, 0, s));
    HIP_OK(launch_embed_scatter_f32(c->f_emb, c->pos, B, c->n2, c->Tp, c->G, D, c->x, s));
    if (c->cfg.vit_arch == 1) HIP_OK(launch_layernorm_f32(c->x, c->npre_w, c->npre_b, c->cfg.vit_ln_eps, M, D, c->x, s));
    for (int l = 0; l < c->cfg.depth; ++l) {
      const int rc = run_vit_block_f32(c, l, B, (l == c->cfg.depth - 1) ? qkv_last : nullptr, nullptr, s);
      if (rc != PIO_OK) return rc;
    }
    if (c->vhead_w) {
      HIP_OK(launch_layernorm(c->op, c->x, c->norm_w, c->norm_b, c->cfg.vit_ln_eps, M, D, nullptr, c->tok_tmp, c->T, c->Tp, s));
      HIP_OK(launch_sgemm_tn(c->tok_tmp, D, c->vhead_w, D, nullptr, 1.f, tokens, c->Dout, B * c->T, c->Dout, D, 0, 0, s));
    } else {
      HIP_OK(launch_layernorm(c->op, c->x, c->norm_w, c->norm_b, c->cfg.vit_ln_eps, M, D, nullptr, tokens, c->T, c->Tp, s));
    }
    return PIO_OK;
  }
  HIP_OK(launch_im2col(c->op, imgs, B, c->cfg.crop_dim, c->cfg.patch_size, c->n, c->Kpad, c->ape, s));
  HIP_OK(launch_token_init(c->x, c->cls, c->pos, c->reg, B, c->cfg.num_registers, c->T, c->Tp, D, s));
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.T = c->T; g.Tp = c->Tp; g.Tk = c->Tk; g.G = c->G; g.n2 = c->n2; g.D = D; g.H = c->H;
  g.x = c->x; g.pos = c->pos; g.q = c->q; g.k = c->k; g.vT = c->vT;
  {
    GemmArgs a = g;
    a.A = c->ape; a.lda = c->Kpad; a.W = c->pe_w; a.bias = c->pe_b; a.M = B * c->n2; a.N = D; a.K = c->Kpad;
    PROF(c, PIO_PROF_VIT_GEMM, 2.0 * B * c->n2 * (double)D * c->Kpe, 0, s, launch_vit_gemm(c->op, EPI_PATCH_EMBED, a, s));
  }
  if (c->cfg.vit_arch == 1)     // CLIP: norm_pre on the position-added tokens, in place (fp32; every row is read whole before it is written)
    HIP_OK(launch_layernorm(c->op, c->x, c->npre_w, c->npre_b, c->cfg.vit_ln_eps, M, D, nullptr, c->x, c->Tp, c->Tp, s));
  VitAttnArgs at;
  at.q = c->q; at.k = c->k; at.vT = c->vT; at.out = c->ao; at.B = B; at.H = c->H; at.T = c->T; at.Tp = c->Tp;
  at.Tk = c->Tk; at.D = D; at.scale = 0.125f;  // 64^-0.5
  const int depth = c->cfg.depth;
  for (int l = 0; l < depth; ++l) {
    const int rc = run_vit_block(c, c->vl[l], B, g, at, (l == depth - 1) ? qkv_last : nullptr, s, l + 1 < depth ? &c->vl[l + 1] : nullptr);
    if (rc != PIO_OK) return rc;
  }
  if (c->vhead_w) {
    // CLIP: forward_features' final norm on every token, then the bias-free head on every token (model.py:788-790): exact fp32
    HIP_OK(launch_layernorm(c->op, c->x, c->norm_w, c->norm_b, c->cfg.vit_ln_eps, M, D, nullptr, c->tok_tmp, c->T, c->Tp, s));
    HIP_OK(launch_sgemm_tn(c->tok_tmp, D, c->vhead_w, D, nullptr, 1.f, tokens, c->Dout, B * c->T, c->Dout, D, 0, 0, s));
    return PIO_OK;
  }
  HIP_OK(launch_layernorm(c->op, c->x, c->norm_w, c->norm_b, c->cfg.vit_ln_eps, M, D, nullptr, tokens, c->T, c->Tp, s));
  return PIO_OK;
}

int pio_bbox_double_dino(pio_handle c, const float* tokens, const int32_t* slices, int32_t B, int32_t NB, int32_t use_cls,
                         int32_t return_type, float* out, pio_stream stream) {
  if (!c || !tokens || !slices || !out) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_double_dino: null argument");
  if (!c->has_vit) return fail(PIO_ERR_NOT_READY, "pio_bbox_double_dino: backbone weights not loaded");
  if (c->cfg.vit_arch == 1) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_double_dino: DINOv2 backbones only (P/src/bbox_utils.py:300-403 re-runs a DINOv2 block)");
  if (B < 1 || NB < 1) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_double_dino: empty batch");
  if (return_type < 0 || return_type > 1) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_double_dino: return_type 0 (cls) or 1 (avg)");
  if (return_type == 0 && !use_cls) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_double_dino: return_type cls needs use_cls");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  if (!c->seq_lens) {
    const int rc = c->dmalloc(&c->seq_lens, (size_t)c->cfg.max_batch);
    if (rc != PIO_OK) return rc;
  }
  const int D = c->D, Ns = B * NB, Gs = use_cls ? c->G : 0;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.T = c->T; g.Tp = c->Tp; g.Tk = c->Tk; g.G = c->G; g.n2 = c->n2; g.D = D; g.H = c->H;
  g.x = c->x; g.pos = c->pos; g.q = c->q; g.k = c->k; g.vT = c->vT;
  VitAttnArgs at;
  at.q = c->q; at.k = c->k; at.vT = c->vT; at.out = c->ao; at.H = c->H; at.T = c->T; at.Tp = c->Tp;
  at.Tk = c->Tk; at.D = D; at.scale = 0.125f; at.lens = c->seq_lens;
  const VitLayerDev& L = c->vl[c->cfg.depth - 1];
  for (int s0 = 0; s0 < Ns; s0 += c->cfg.max_batch) {         // chunks of max_batch sequences through the ViT workspace
    const int ns = std::min(c->cfg.max_batch, Ns - s0);
    at.B = ns;
    HIP_OK(launch_box_sequences(tokens, slices, s0, ns, NB, c->T, c->Tp, c->G, c->n, D, use_cls, c->x, c->seq_lens, s));
    const int rc = c->vit_f32 ? run_vit_block_f32(c, c->cfg.depth - 1, ns, nullptr, c->seq_lens, s) : run_vit_block(c, L, ns, g, at, nullptr, s);
    if (rc != PIO_OK) return rc;
    HIP_OK(launch_box_seq_reduce(c->x, c->seq_lens, ns, c->Tp, D, Gs, return_type, out + (size_t)s0 * D, s));
  }
  return PIO_OK;
}

int pio_cls_attention(pio_handle c, const float* qkv_last, const float* tokens, int32_t B, float* self_attn,
                      float* head_maps, float* avg_token, float* disentangled, pio_stream stream) {
  if (!c || !qkv_last || !self_attn) return fail(PIO_ERR_INVALID_ARG, "pio_cls_attention: null argument");
  if (!c->has_vit) return fail(PIO_ERR_NOT_READY, "pio_cls_attention: backbone not loaded");
  if (c->cfg.vit_arch == 1) return fail(PIO_ERR_INVALID_ARG, "pio_cls_attention: the CLIP ViT exposes no qkv hook (has_attention = False, P/src/model.py:864-865)");
  if (B < 1 || B > c->cfg.max_batch) return fail(PIO_ERR_CAPACITY, "pio_cls_attention: batch above max_batch");
  if ((avg_token || disentangled) && !tokens) return fail(PIO_ERR_INVALID_ARG, "pio_cls_attention: tokens required");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const int Hr = c->cfg.readout_heads;
  float* hl = head_maps ? head_maps : (disentangled ? c->head_logits : nullptr);
  HIP_OK(launch_cls_logits(qkv_last, B, c->T, c->G, c->D, Hr, c->cfg.readout_scale, self_attn, hl, s));
  HIP_OK(launch_softmax_rows(self_attn, self_attn, B, c->n2, s));
  if (avg_token)   // (self_attn[...,None] * patches).mean(1)
    HIP_OK(launch_region_reduce(tokens, c->T, c->G, c->D, c->n2, self_attn, nullptr, B, 1.0f / c->n2, avg_token, s));
  if (disentangled) {
    HIP_OK(launch_softmax_rows(hl, c->head_sm, B * Hr, c->n2, s));
    HIP_OK(launch_region_reduce(tokens, c->T, c->G, c->D, c->n2, c->head_sm, c->head_img, B * Hr, 1.0f / c->n2,
                                disentangled, s));
  }
  return PIO_OK;
}

int pio_trace_grids(pio_handle c, const double* xy, const int32_t* offsets, int32_t B, int32_t total_points,
                    float* grids, pio_stream stream) {
  if (!c || !offsets || !grids || (total_points > 0 && !xy)) return fail(PIO_ERR_INVALID_ARG, "pio_trace_grids: null argument");
  if (B < 1) return fail(PIO_ERR_INVALID_ARG, "pio_trace_grids: B < 1");
  HIP_OK(hipSetDevice(c->cfg.device));
  HIP_OK(launch_trace_grids(xy, offsets, B, c->n, grids, (hipStream_t)stream));
  return PIO_OK;
}

int pio_bbox_weights(pio_handle c, const int32_t* boxes, int32_t B, int32_t NB, int32_t mode, float variance,
                     const int32_t* center_choice, float* attn, float* weights, int32_t single_map, float* single,
                     pio_stream stream) {
  if (!c || !boxes || !weights || (single_map && !single)) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_weights: null argument");
  if (B < 1 || NB < 1) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_weights: empty batch");
  if (mode < 0 || mode > 3) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_weights: mode must be 0..3");
  if (mode == 1 && !(variance > 0.f)) return fail(PIO_ERR_INVALID_ARG, "pio_bbox_weights: gaussian mode needs variance > 0");
  HIP_OK(hipSetDevice(c->cfg.device));
  HIP_OK(launch_bbox_weights(boxes, B, NB, c->n, mode, variance, center_choice, attn, weights, single_map, single,
                             (hipStream_t)stream));
  return PIO_OK;
}

int pio_region_reduce(pio_handle c, const float* tokens, int32_t B, const float* weights, const int32_t* img_index,
                      int32_t R, float scale, float* out, pio_stream stream) {
  if (!c || !tokens || !weights || !out) return fail(PIO_ERR_INVALID_ARG, "pio_region_reduce: null argument");
  if (R < 1 || B < 1) return fail(PIO_ERR_INVALID_ARG, "pio_region_reduce: empty input");
  if (!img_index && R > B) return fail(PIO_ERR_INVALID_ARG, "pio_region_reduce: img_index required when R > B");
  HIP_OK(hipSetDevice(c->cfg.device));
  HIP_OK(launch_region_reduce(tokens, c->T, c->G, c->Dout, c->n2, weights, img_index, R, scale, out, (hipStream_t)stream));
  return PIO_OK;
}

int pio_gaussian_map(pio_handle c, float variance, float* map, pio_stream stream) {
  if (!c || !map) return fail(PIO_ERR_INVALID_ARG, "pio_gaussian_map: null argument");
  if (!(variance > 0.f)) return fail(PIO_ERR_INVALID_ARG, "pio_gaussian_map: variance must be > 0 (0 = host-drawn one-hot)");
  HIP_OK(hipSetDevice(c->cfg.device));
  HIP_OK(launch_gaussian_map(c->n, variance, map, (hipStream_t)stream));
  return PIO_OK;
}

int pio_mem_project(pio_handle c, float* q, int32_t N, float temperature, int32_t normalize, float* out,
                    int32_t n_best, float* best_sims, pio_stream stream) {
  if (!c || !q || !out) return fail(PIO_ERR_INVALID_ARG, "pio_mem_project: null argument");
  if (!c->bank) return fail(PIO_ERR_NOT_READY, "pio_mem_project: memory bank not set");
  if (N < 1) return fail(PIO_ERR_INVALID_ARG, "pio_mem_project: N < 1");
  if (n_best < 0 || n_best > 16 || (n_best > 0 && !best_sims)) return fail(PIO_ERR_INVALID_ARG, "pio_mem_project: n_best must be 0..16");
  if (!(temperature > 0.f)) return fail(PIO_ERR_INVALID_ARG, "pio_mem_project: temperature must be > 0");
  HIP_OK(hipSetDevice(c->cfg.device));
  ProjectArgs a;
  a.bank = c->bank; a.inv_norm = c->bank_inv; a.M = c->bank_rows; a.D = c->bank_dim; a.q = q; a.N = N;
  a.temperature = temperature; a.normalize = normalize; a.out = out; a.n_best = n_best; a.best_sims = best_sims;
  a.part_acc = c->part_acc; a.part_ml = c->part_ml; a.part_best = c->sims; a.parts = c->parts; a.n_best_cap = 16; a.part_rows = c->parts * 24;
  a.bank_scale = c->bank_scale; a.bank_split = c->bank_split;
  double passes = 0;                         // bank passes as launch_mem_project makes them: 48 queries while more than 32 are left (D = 768), 32 while more than 16
  for (int left = N; left > 0; left -= (left > 32 && !(left > 48 && left <= 64) && c->bank_dim == 768) ? 48 : (left > 16 ? 32 : 16)) passes += 1;
  PROF(c, PIO_PROF_MEM_PROJECT, 4.0 * N * (double)c->bank_rows * c->bank_dim,
       passes * ((double)c->bank_rows * c->bank_dim * 4.0 + (double)c->bank_rows * 4.0), (hipStream_t)stream,
       launch_mem_project(a, (hipStream_t)stream));
  return PIO_OK;
}

int pio_mem_topk(pio_handle c, float* q, int32_t N, int32_t k, float* best_sims, int64_t* best_rows, pio_stream stream) {
  if (!c || !q || !best_sims || !best_rows) return fail(PIO_ERR_INVALID_ARG, "pio_mem_topk: null argument");
  if (!c->bank) return fail(PIO_ERR_NOT_READY, "pio_mem_topk: memory bank not set");
  if (N < 1) return fail(PIO_ERR_INVALID_ARG, "pio_mem_topk: N < 1");
  if (k < 1 || k > 16 || k > c->bank_rows) return fail(PIO_ERR_INVALID_ARG, "pio_mem_topk: k must be 1..16 and <= the rows of the bank");
  HIP_OK(hipSetDevice(c->cfg.device));
  HIP_OK(launch_mem_topk(c->bank, c->bank_inv, c->bank_rows, c->bank_dim, q, N, k, c->sims, best_sims, best_rows, (hipStream_t)stream));
  return PIO_OK;
}

int pio_text_project(pio_handle c, const float* x, int32_t N, int32_t in_dim, const float* w1, const float* b1, int32_t out_dim,
                     const float* w2, const float* b2, int32_t act, float* hidden, float* out, pio_stream stream) {
  if (!c || !x || !w1 || !b1 || !out) return fail(PIO_ERR_INVALID_ARG, "pio_text_project: null argument");
  if (N < 1 || in_dim < 32 || in_dim % 32 || out_dim < 32 || out_dim % 32)
    return fail(PIO_ERR_INVALID_ARG, "pio_text_project: N >= 1 and widths that are multiples of 32");
  if (act < 0 || act > 3) return fail(PIO_ERR_INVALID_ARG, "pio_text_project: act is 0 none, 1 relu, 2 tanh, 3 sigmoid");
  if (w2 && (!b2 || !hidden)) return fail(PIO_ERR_INVALID_ARG, "pio_text_project: a hidden layer needs its bias and the [N, out_dim] scratch");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  float* first = w2 ? hidden : out;
  // linear_layer; the activation precedes every hidden layer (talk2dino.py:77-81), so it is fused here only when one follows
  HIP_OK(launch_sgemm_tn(x, in_dim, w1, in_dim, b1, 1.0f, first, out_dim, N, out_dim, in_dim, w2 && act == 1, 0, s));
  if (w2) {
    if (act >= 2) HIP_OK(launch_activation_f32(first, (size_t)N * out_dim, act, s));
    HIP_OK(launch_sgemm_tn(first, out_dim, w2, out_dim, b2, 1.0f, out, out_dim, N, out_dim, out_dim, 0, 0, s));
  }
  return PIO_OK;
}

int pio_revert_transformation(pio_handle c, const float* x, int32_t N, float* out, pio_stream stream) {
  if (!c || !x || !out || N < 1) return fail(PIO_ERR_INVALID_ARG, "pio_revert_transformation: bad argument");
  if (!c->has_inv) return fail(PIO_ERR_NOT_READY, "pio_revert_transformation: talk2dino.A_pinv / talk2dino.b not loaded");
  HIP_OK(hipSetDevice(c->cfg.device));
  HIP_OK(launch_revert(x, c->inv_b, c->A_pinv, N, c->D, c->cfg.prefix_size, out, (hipStream_t)stream));
  return PIO_OK;
}

int pio_decode_greedy(pio_handle c, const float* prefix, int32_t N, int32_t steps, int32_t* ids, float* logprob,
                      pio_stream stream) {
  if (!c || !prefix || !ids) return fail(PIO_ERR_INVALID_ARG, "pio_decode_greedy: null argument");
  if (!c->has_dec || !c->has_clip_project) return fail(PIO_ERR_NOT_READY, "pio_decode_greedy: decoder weights (with clip_project) not loaded");
  if (N < 1 || N > c->cfg.max_prefixes) return fail(PIO_ERR_CAPACITY, "pio_decode_greedy: N above max_prefixes");
  if (logprob && N > 64) return fail(PIO_ERR_CAPACITY, "pio_decode_greedy: log-probabilities are built for <= 64 prefixes per call");
  if (steps < 1 || steps > c->cfg.max_steps || steps > 64) return fail(PIO_ERR_CAPACITY, "pio_decode_greedy: steps above max_steps");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const int E = c->cfg.dec_embd, PS = c->cfg.prefix_size;
  DecoderArgs a;
  a.N = N; a.steps = steps; a.E = E; a.heads = c->cfg.dec_heads; a.layers = c->cfg.dec_layers; a.vocab = c->cfg.dec_vocab;
  a.prefix_size = PS; a.eps = c->cfg.dec_ln_eps; a.prefix = c->prefix_buf; a.clip_w = c->clip_w; a.clip_b = c->clip_b;
  a.wte = c->wte; a.wpe = c->wpe; a.head_w = c->head_w; a.head_c = c->head_c; a.head_d = c->head_d; a.layer = c->dl.data();
  a.x = c->dx; a.qkv = c->dqkv; a.att = c->datt; a.hid = c->dhid; a.splitk_ws = c->splitk_ws; a.splitk_cnt = c->splitk_cnt; a.lm_tail = c->lm_tail ? 1 : 0; a.kcache = c->kcache; a.vcache = c->vcache;
  a.max_steps = c->cfg.max_steps; a.logits = c->logits; a.ids = c->ids_buf; a.logprob = logprob ? c->logprob_buf : nullptr;
  a.head_w16 = c->head_w16; a.head_w16_unscale = c->head_w16_unscale; a.head_bound_coef = c->head_bound_coef;
  a.xh = c->dec_xh; a.lm_stats = c->lm_stats; a.lm_gmax = c->lm_gmax;
  HIP_OK(hipMemcpyAsync(c->prefix_buf, prefix, (size_t)N * PS * 4, hipMemcpyDeviceToDevice, s));
  HIP_OK(hipMemsetAsync(c->splitk_cnt, 0, DEC_TICKET_WORDS * sizeof(unsigned), s));   // tickets start at zero whatever happened before
  // algorithmic work of a KV-cached decode: per token 4 layers x 12 E^2 MACs + the tied LM head;
  // bytes = every fp32 weight read once per step
  const double layer_params = (double)c->cfg.dec_layers * 12.0 * E * E, head_params = (double)c->cfg.dec_vocab * E;
  const double dec_flops = 2.0 * N * steps * (layer_params + head_params);
  // bytes the path streams per step: the layers in fp32; the head in fp16 when ids only are wanted (fp16 filter,
  // plus a handful of fp32 rows for the exact re-evaluation), in fp32 when log-probabilities are
  const double dec_bytes = steps * (4.0 * layer_params + (logprob ? 4.0 : 2.0) * head_params);
  if (c->use_graph) {
    const GraphKey key{N, steps, logprob ? 1 : 0};
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
      hipGraph_t graph = nullptr;
      HIP_OK(hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeThreadLocal));
      hipError_t le = launch_decode_greedy(a, c->capture_stream);
      hipError_t ce = hipStreamEndCapture(c->capture_stream, &graph);
      if (le != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return fail(PIO_ERR_HIP, std::string("decode capture: ") + hipGetErrorString(le));
      }
      HIP_OK(ce);
      hipGraphExec_t exec = nullptr;
      const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      HIP_OK(ie);
      it = c->graphs.emplace(key, exec).first;
    }
    PROF(c, PIO_PROF_DECODE, dec_flops, dec_bytes, s, hipGraphLaunch(it->second, s));
  } else {
    PROF(c, PIO_PROF_DECODE, dec_flops, dec_bytes, s, launch_decode_greedy(a, s));
  }
  HIP_OK(hipMemcpyAsync(ids, c->ids_buf, (size_t)N * steps * 4, hipMemcpyDeviceToDevice, s));
  if (logprob) HIP_OK(hipMemcpyAsync(logprob, c->logprob_buf, (size_t)N * steps * 4, hipMemcpyDeviceToDevice, s));
  return PIO_OK;
}

int pio_viecap_set_entities(pio_handle c, const float* host_embeddings, int32_t K, int32_t C) {
  if (!c || !host_embeddings || K < 1) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_set_entities: bad argument");
  if (!c->has_map || C != c->map_C) return fail(PIO_ERR_SHAPE, "pio_viecap_set_entities: width differs from the mapping network's input");
  if (c->ent) return fail(PIO_ERR_INVALID_ARG, "entity embeddings already set");
  HIP_OK(hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = upload_f32(c, host_embeddings, (size_t)K * C, &c->ent))) return rc;
  HIP_OK(launch_l2norm_rows(c->ent, K, C, nullptr));        // texts_embeddings /= norm (retrieval_categories.py:90), once
  HIP_OK(hipDeviceSynchronize());
  c->ent_K = K;
  return PIO_OK;
}

int pio_viecap_prompt_length(pio_handle c) { return (c && c->has_map) ? c->map_Lc : 0; }

int pio_viecap_mapping(pio_handle c, float* feats, int32_t N, float* out, pio_stream stream) {
  if (!c || !feats || !out) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_mapping: null argument");
  if (!c->has_map) return fail(PIO_ERR_NOT_READY, "pio_viecap_mapping: mapping network not loaded");
  if (N < 1 || N > c->cfg.max_prefixes) return fail(PIO_ERR_CAPACITY, "pio_viecap_mapping: N above max_prefixes");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  HIP_OK(launch_l2norm_rows(feats, N, c->map_C, s));          // image_features /= norm, in place (entrypoint.py:108)
  ViecapMapArgs a;
  a.N = N; a.C = c->map_C; a.E = c->cfg.dec_embd; a.Lp = c->map_Lp; a.Lc = c->map_Lc; a.heads = 8; a.layers = c->map_layers;
  a.hidden = c->map_hidden; a.eps = 1e-5f; a.feats = feats; a.lin_w = c->map_lin_w; a.lin_b = c->map_lin_b; a.prefix_const = c->map_prefix;
  a.layer = c->ml.data(); a.lin = c->map_lin; a.x = c->map_x; a.ln = c->map_ln; a.q = c->map_q; a.kv = c->map_kv; a.att = c->map_att;
  a.hid = c->map_hid; a.out = out;
  HIP_OK(launch_viecap_mapping(a, s));
  return PIO_OK;
}

int pio_viecap_entity_logits(pio_handle c, const float* feats, int32_t N, float temperature, float* out, pio_stream stream) {
  if (!c || !feats || !out) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_entity_logits: null argument");
  if (!c->ent) return fail(PIO_ERR_NOT_READY, "pio_viecap_entity_logits: entity embeddings not set");
  if (N < 1 || !(temperature > 0.f)) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_entity_logits: bad argument");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  // softmax(f t^T / T) over the vocabulary (retrieval_categories.py:92-93); f is the already normalised feature
  HIP_OK(launch_sgemm_tn(feats, c->map_C, c->ent, c->map_C, nullptr, 1.0f / temperature, out, c->ent_K, N, c->ent_K, c->map_C, 0, 0, s));
  HIP_OK(launch_softmax_rows(out, out, N, c->ent_K, s));
  return PIO_OK;
}

int pio_lm_score(pio_handle c, const int32_t* tokens, const int32_t* lens, int32_t N, int32_t Lmax, float* nll, pio_stream stream) {
  if (!c || !tokens || !lens || !nll) return fail(PIO_ERR_INVALID_ARG, "pio_lm_score: null argument");
  if (!c->has_dec) return fail(PIO_ERR_NOT_READY, "pio_lm_score: language model not loaded");
  if (N < 1 || N > c->cfg.max_prefixes || N > 64) return fail(PIO_ERR_CAPACITY, "pio_lm_score: 1 <= N <= min(max_prefixes, 64) rows per call");
  if (Lmax < 1 || Lmax > c->cfg.max_steps || Lmax > 256) return fail(PIO_ERR_CAPACITY, "pio_lm_score: Lmax above max_steps");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  DecoderArgs a;
  a.N = N; a.steps = 1; a.E = c->cfg.dec_embd; a.heads = c->cfg.dec_heads; a.layers = c->cfg.dec_layers; a.vocab = c->cfg.dec_vocab;
  a.prefix_size = c->cfg.prefix_size; a.eps = c->cfg.dec_ln_eps; a.prefix = nullptr; a.clip_w = nullptr; a.clip_b = nullptr;
  a.wte = c->wte; a.wpe = c->wpe; a.head_w = c->head_w; a.head_c = c->head_c; a.head_d = c->head_d; a.layer = c->dl.data();
  a.x = c->dx; a.qkv = c->dqkv; a.att = c->datt; a.hid = c->dhid; a.splitk_ws = c->splitk_ws; a.splitk_cnt = c->splitk_cnt; a.lm_tail = c->lm_tail ? 1 : 0;
  a.kcache = c->kcache; a.vcache = c->vcache; a.max_steps = c->cfg.max_steps; a.logits = c->logits; a.ids = c->ids_buf; a.logprob = nullptr;
  a.head_w16 = c->head_w16; a.head_w16_unscale = c->head_w16_unscale; a.head_bound_coef = c->head_bound_coef;
  a.xh = c->dec_xh; a.lm_stats = c->lm_stats; a.lm_gmax = c->lm_gmax; a.pos_base = 0;
  HIP_OK(hipMemsetAsync(c->splitk_cnt, 0, DEC_TICKET_WORDS * sizeof(unsigned), s));
  HIP_OK(launch_lm_score(a, tokens, lens, Lmax, nll, s));
  return PIO_OK;
}

// DecoderArgs of a single-position / prompt pass over the handle's language model (no greedy loop)
static DecoderArgs lm_args(pio_context* c, int N) {
  DecoderArgs a;
  a.N = N; a.steps = 1; a.E = c->cfg.dec_embd; a.heads = c->cfg.dec_heads; a.layers = c->cfg.dec_layers; a.vocab = c->cfg.dec_vocab;
  a.prefix_size = c->cfg.prefix_size; a.eps = c->cfg.dec_ln_eps; a.prefix = nullptr; a.clip_w = nullptr; a.clip_b = nullptr;
  a.wte = c->wte; a.wpe = c->wpe; a.head_w = c->head_w; a.head_c = c->head_c; a.head_d = c->head_d; a.layer = c->dl.data();
  a.x = c->dx; a.qkv = c->dqkv; a.att = c->datt; a.hid = c->dhid; a.splitk_ws = c->splitk_ws; a.splitk_cnt = c->splitk_cnt; a.lm_tail = c->lm_tail ? 1 : 0;
  a.kcache = c->kcache; a.vcache = c->vcache; a.max_steps = c->cfg.max_steps; a.logits = c->logits; a.ids = c->ids_buf; a.logprob = nullptr;
  a.head_w16 = c->head_w16; a.head_w16_unscale = c->head_w16_unscale; a.head_bound_coef = c->head_bound_coef;
  a.xh = c->dec_xh; a.lm_stats = c->lm_stats; a.lm_gmax = c->lm_gmax; a.pos_base = 0;
  return a;
}

static int beam_scratch(pio_context* c) {
  if (c->beam_stats) return PIO_OK;
  int rc;
  // pio_lm_prefill / pio_lm_advance take at most 16 rows (the beams of one image) and k_kv_gather indexes [layer][a.N][max_steps][E]:
  // 16 rows of scratch, not max_prefixes (ViECap defaults: 0.3 GB for both instead of 2.4 GB).  Allocated on the first beam call
  // (a hipMalloc: it synchronises the device once).
  const size_t rows = c->cfg.max_prefixes < 16 ? c->cfg.max_prefixes : 16;
  const size_t kv = (size_t)c->cfg.dec_layers * rows * c->cfg.max_steps * c->cfg.dec_embd;
  if ((rc = c->dmalloc(&c->beam_k, kv))) return rc;
  if ((rc = c->dmalloc(&c->beam_v, kv))) return rc;
  return c->dmalloc(&c->beam_stats, (size_t)2 * 16);
}

int pio_viecap_build_prompt(pio_handle c, const float* cont, const int32_t* tokens, int32_t N, int32_t Lt, int32_t soft_first,
                            float* prompt, pio_stream stream) {
  if (!c || !prompt || (Lt > 0 && !tokens)) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_build_prompt: null argument");
  if (!cont && Lt < 1) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_build_prompt: neither a soft prompt nor prompt tokens");
  if (!c->has_dec || (cont && !c->has_map)) return fail(PIO_ERR_NOT_READY, "pio_viecap_build_prompt: language model / mapping network not loaded");
  if (N < 1 || Lt < 0) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_build_prompt: bad shape");
  HIP_OK(hipSetDevice(c->cfg.device));
  HIP_OK(launch_build_prompt(cont, tokens, c->wte, N, cont ? c->map_Lc : 0, Lt, c->cfg.dec_embd, c->cfg.dec_vocab, soft_first, prompt,
                             (hipStream_t)stream));
  return PIO_OK;
}

int pio_lm_prefill(pio_handle c, const float* embeds, int32_t N, int32_t P, float* logp, pio_stream stream) {
  if (!c || !embeds || !logp) return fail(PIO_ERR_INVALID_ARG, "pio_lm_prefill: null argument");
  if (!c->has_dec) return fail(PIO_ERR_NOT_READY, "pio_lm_prefill: language model not loaded");
  if (N < 1 || N > 16 || N > c->cfg.max_prefixes) return fail(PIO_ERR_CAPACITY, "pio_lm_prefill: 1 <= N <= min(16, max_prefixes) rows per call");
  if (P < 1 || P > c->cfg.max_steps || P > 256) return fail(PIO_ERR_CAPACITY, "pio_lm_prefill: P above max_steps");
  HIP_OK(hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = beam_scratch(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const DecoderArgs a = lm_args(c, N);
  HIP_OK(hipMemsetAsync(c->splitk_cnt, 0, DEC_TICKET_WORDS * sizeof(unsigned), s));
  HIP_OK(launch_lm_prefill(a, embeds, P, c->beam_stats, logp, s));
  return PIO_OK;
}

int pio_lm_advance(pio_handle c, const int32_t* tokens, const int32_t* src_rows, int32_t N, int32_t pos, float* logp, pio_stream stream) {
  if (!c || !tokens || !logp) return fail(PIO_ERR_INVALID_ARG, "pio_lm_advance: null argument");
  if (!c->has_dec) return fail(PIO_ERR_NOT_READY, "pio_lm_advance: language model not loaded");
  if (N < 1 || N > 16 || N > c->cfg.max_prefixes) return fail(PIO_ERR_CAPACITY, "pio_lm_advance: 1 <= N <= min(16, max_prefixes) rows per call");
  if (pos < 1 || pos + 1 > c->cfg.max_steps || pos + 1 > 256) return fail(PIO_ERR_CAPACITY, "pio_lm_advance: position above max_steps");
  HIP_OK(hipSetDevice(c->cfg.device));
  int rc;
  if ((rc = beam_scratch(c))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const DecoderArgs a = lm_args(c, N);
  HIP_OK(hipMemsetAsync(c->splitk_cnt, 0, DEC_TICKET_WORDS * sizeof(unsigned), s));
  HIP_OK(launch_lm_advance(a, tokens, src_rows, pos, c->beam_k, c->beam_v, c->beam_stats, logp, s));
  return PIO_OK;
}

int pio_beam_select(pio_handle c, const float* logp, const float* scores, const float* lens, const int32_t* stopped, int32_t W,
                    float* out_val, int64_t* out_idx, pio_stream stream) {
  if (!c || !logp || !out_val || !out_idx || (scores && (!lens || !stopped)))
    return fail(PIO_ERR_INVALID_ARG, "pio_beam_select: null argument");
  if (W < 1 || W > 8) return fail(PIO_ERR_INVALID_ARG, "pio_beam_select: beam width must be 1..8");
  HIP_OK(hipSetDevice(c->cfg.device));
  HIP_OK(launch_beam_select(logp, scores, lens, stopped, W, c->cfg.dec_vocab, out_val, out_idx, (hipStream_t)stream));
  return PIO_OK;
}

int pio_viecap_decode(pio_handle c, const float* cont, const int32_t* tokens, int32_t N, int32_t Lt, int32_t soft_first,
                      int32_t steps, int32_t* ids, pio_stream stream) {
  if (!c || !ids || (Lt > 0 && !tokens)) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_decode: null argument");
  if (!cont && Lt < 1) return fail(PIO_ERR_INVALID_ARG, "pio_viecap_decode: neither a soft prompt nor prompt tokens");
  if (!c->has_dec || (cont && !c->has_map)) return fail(PIO_ERR_NOT_READY, "pio_viecap_decode: language model / mapping network not loaded");
  if (N < 1 || N > c->cfg.max_prefixes) return fail(PIO_ERR_CAPACITY, "pio_viecap_decode: N above max_prefixes");
  const int Lc = cont ? c->map_Lc : 0;       // cont == NULL: only_hard_prompt (entrypoint.py:130-131), the word embeddings alone
  const int P = Lc + Lt;
  if (Lt < 0 || steps < 1 || steps > 64 || P + steps - 1 > c->cfg.max_steps)
    return fail(PIO_ERR_CAPACITY, "pio_viecap_decode: prompt + generated positions above max_steps");
  HIP_OK(hipSetDevice(c->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const int E = c->cfg.dec_embd;
  int rc;
  if (!c->prompt_buf) {
    if ((rc = c->dmalloc(&c->prompt_buf, (size_t)c->cfg.max_prefixes * c->cfg.max_steps * E))) return rc;
    if ((rc = c->dmalloc(&c->tok_buf, (size_t)c->cfg.max_prefixes * c->cfg.max_steps))) return rc;
  }
  const bool batched_prefill = c->batched_prefill;
  if (batched_prefill && !c->pre_x) {                 // workspace of the batched prompt prefill: 2 048 rows (56 MB)
    const size_t R = pio_context::kPrefillRows;
    if ((rc = c->dmalloc(&c->pre_x, R * E))) return rc;
    if ((rc = c->dmalloc(&c->pre_qkv, R * 3 * E))) return rc;
    if ((rc = c->dmalloc(&c->pre_att, R * E))) return rc;
    if ((rc = c->dmalloc(&c->pre_hid, R * 4 * E))) return rc;
  }
  if (Lt > 0) HIP_OK(hipMemcpyAsync(c->tok_buf, tokens, (size_t)N * Lt * 4, hipMemcpyDeviceToDevice, s));
  HIP_OK(launch_build_prompt(cont, c->tok_buf, c->wte, N, Lc, Lt, E, c->cfg.dec_vocab, soft_first, c->prompt_buf, s));
  DecoderArgs a;
  a.N = N; a.steps = steps; a.E = E; a.heads = c->cfg.dec_heads; a.layers = c->cfg.dec_layers; a.vocab = c->cfg.dec_vocab;
  a.prefix_size = c->cfg.prefix_size; a.eps = c->cfg.dec_ln_eps; a.prefix = nullptr; a.clip_w = nullptr; a.clip_b = nullptr;
  a.wte = c->wte; a.wpe = c->wpe; a.head_w = c->head_w; a.head_c = c->head_c; a.head_d = c->head_d; a.layer = c->dl.data();
  a.x = c->dx; a.qkv = c->dqkv; a.att = c->datt; a.hid = c->dhid; a.splitk_ws = c->splitk_ws; a.splitk_cnt = c->splitk_cnt; a.lm_tail = c->lm_tail ? 1 : 0;
  a.kcache = c->kcache; a.vcache = c->vcache; a.max_steps = c->cfg.max_steps; a.logits = c->logits; a.ids = c->ids_buf; a.logprob = nullptr;
  a.head_w16 = c->head_w16; a.head_w16_unscale = c->head_w16_unscale; a.head_bound_coef = c->head_bound_coef;
  a.xh = c->dec_xh; a.lm_stats = c->lm_stats; a.lm_gmax = c->lm_gmax; a.pos_base = P - 1;
  if (batched_prefill) { a.pre_x = c->pre_x; a.pre_qkv = c->pre_qkv; a.pre_att = c->pre_att; a.pre_hid = c->pre_hid; a.pre_rows = pio_context::kPrefillRows; }
  HIP_OK(hipMemsetAsync(c->splitk_cnt, 0, DEC_TICKET_WORDS * sizeof(unsigned), s));
  if (c->use_graph) {
    const pio_context::PKey key{N, P, steps};
    auto it = c->pgraphs.find(key);
    if (it == c->pgraphs.end()) {
      hipGraph_t graph = nullptr;
      HIP_OK(hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeThreadLocal));
      hipError_t le = launch_decode_prompted(a, c->prompt_buf, P, c->capture_stream);
      hipError_t ce = hipStreamEndCapture(c->capture_stream, &graph);
      if (le != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return fail(PIO_ERR_HIP, std::string("prompted decode capture: ") + hipGetErrorString(le));
      }
      HIP_OK(ce);
      hipGraphExec_t exec = nullptr;
      hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      HIP_OK(ie);
      if (c->pgraphs.size() >= pio_context::kMaxPGraphs) {       // evict the least recently used graph
        auto old = c->pgraphs.begin();
        for (auto jt = c->pgraphs.begin(); jt != c->pgraphs.end(); ++jt)
          if (jt->second.last_use < old->second.last_use) old = jt;
        // it may still be running where it was last launched: wait for the event recorded behind that launch (an event outlives the
        // caller's stream; round 4 synchronised the stored stream handle, which a caller may have destroyed since -- the error then
        // left the new exec leaked and the dead entry in place for every later decode).  Whatever the wait says, the entry goes.
        if (old->second.done) { (void)hipEventSynchronize(old->second.done); (void)hipEventDestroy(old->second.done); }
        (void)hipGraphExecDestroy(old->second.exec);
        c->pgraphs.erase(old);
      }
      hipEvent_t done = nullptr;
      const hipError_t ee = hipEventCreateWithFlags(&done, hipEventDisableTiming);
      if (ee != hipSuccess) { (void)hipGraphExecDestroy(exec); HIP_OK(ee); }
      it = c->pgraphs.emplace(key, pio_context::PGraph{exec, 0, done}).first;
    }
    it->second.last_use = ++c->pgraph_clock;
    HIP_OK(hipGraphLaunch(it->second.exec, s));
    HIP_OK(hipEventRecord(it->second.done, s));
  } else {
    HIP_OK(launch_decode_prompted(a, c->prompt_buf, P, s));
  }
  HIP_OK(hipMemcpyAsync(ids, c->ids_buf, (size_t)N * steps * 4, hipMemcpyDeviceToDevice, s));
  return PIO_OK;
}

int pio_profile_enable(pio_handle c, int32_t on) {
  if (!c) return fail(PIO_ERR_INVALID_ARG, "null handle");
  HIP_OK(hipSetDevice(c->cfg.device));
  c->prof_on = on != 0;
  c->prof.clear();
  c->ev_used = 0;
  return PIO_OK;
}

int pio_profile_read(pio_handle c, int32_t cls, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  if (!c || !total_ms || !launches || !flops || !bytes) return fail(PIO_ERR_INVALID_ARG, "pio_profile_read: null argument");
  HIP_OK(hipSetDevice(c->cfg.device));
  double ms = 0, fl = 0, by = 0;
  int64_t n = 0;
  for (auto& r : c->prof) {
    if (r.cls != cls) continue;
    HIP_OK(hipEventSynchronize(r.b));
    float t = 0.f;
    HIP_OK(hipEventElapsedTime(&t, r.a, r.b));
    ms += t; fl += r.flops; by += r.bytes; ++n;
  }
  *total_ms = ms; *launches = n; *flops = fl; *bytes = by;
  return PIO_OK;
}

int pio_ctx_clean(pio_handle c, const float* dirty, const float* ctx, int32_t R, int32_t D, int32_t rows_per_ctx,
                  int32_t cleaning_type, float alpha, int32_t normalize_inputs, float* out, pio_stream stream) {
  if (!c || !dirty || !ctx || !out) return fail(PIO_ERR_INVALID_ARG, "pio_ctx_clean: null argument");
  if (R < 1 || D < 1 || rows_per_ctx < 1) return fail(PIO_ERR_INVALID_ARG, "pio_ctx_clean: empty input");
  if (cleaning_type < 0 || cleaning_type > 1)
    return fail(PIO_ERR_INVALID_ARG, "pio_ctx_clean: cleaning_type 0 (orthogonal_projection) or 1 (contrastive_mask)");
  HIP_OK(hipSetDevice(c->cfg.device));
  if (D > 1024) return fail(PIO_ERR_SHAPE, "pio_ctx_clean: D above 1024");
  HIP_OK(launch_ctx_clean(dirty, ctx, R, D, rows_per_ctx, cleaning_type, alpha, normalize_inputs, out, (hipStream_t)stream));
  return PIO_OK;
}

int pio_preprocess(pio_handle c, const void* pixels, const int64_t* offsets, const int32_t* wh, int32_t B,
                   int32_t resize_dim, int32_t crop_dim, int32_t mode, float* out, void* stream) {
  if (!c || !pixels || !offsets || !wh || !out || B < 1 || resize_dim < 1 || (mode == 0 && crop_dim < 1) || mode < 0 || mode > 1)
    return fail(PIO_ERR_INVALID_ARG, "pio_preprocess: bad argument");
  hipStream_t s = (hipStream_t)stream;
  HIP_OK(hipSetDevice(c->cfg.device));
  const int S = mode == 0 ? crop_dim : resize_dim;
  if (!c->prep_lut) {   // ((v / 255) - mean) / std in IEEE fp32, the operations of ToTensor + Normalize
    // ImageNet statistics for DINOv2 (P/src/model.py:351), OpenAI-CLIP's for the CLIP ViT (:381-382)
    static const float mean_in[3] = {0.485f, 0.456f, 0.406f}, std_in[3] = {0.229f, 0.224f, 0.225f};
    static const float mean_cl[3] = {0.48145466f, 0.4578275f, 0.40821073f}, std_cl[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    const float* mean = c->cfg.vit_arch == 1 ? mean_cl : mean_in;
    const float* stdv = c->cfg.vit_arch == 1 ? std_cl : std_in;
    std::vector<float> lut(768);
    for (int ch = 0; ch < 3; ++ch)
      for (int v = 0; v < 256; ++v) {
        volatile float x = (float)v / 255.0f;
        volatile float y = x - mean[ch];
        lut[ch * 256 + v] = y / stdv[ch];
      }
    HIP_OK(hipMalloc((void**)&c->prep_lut, 768 * sizeof(float)));
    HIP_OK(hipMemcpy(c->prep_lut, lut.data(), 768 * sizeof(float), hipMemcpyHostToDevice));
  }
  // ---- host: sizes, crop windows and coefficient tables ----
  std::vector<PrepImage> imgs(B);
  std::vector<int32_t> tables;
  size_t tmp_bytes = 0;
  int max_tmp_elems = 0;
  PilAxis ax;
  for (int i = 0; i < B; ++i) {
    const int w = wh[2 * i], h = wh[2 * i + 1];
    if (w < 1 || h < 1) return fail(PIO_ERR_INVALID_ARG, "pio_preprocess: empty image");
    int nw, nh, left = 0, top = 0;
    if (mode == 0) {   // torchvision F.resize(int): the shorter side becomes resize_dim, the longer int(size * long / short)
      if (w <= h) { nw = resize_dim; nh = (int)((double)resize_dim * h / w); }
      else { nh = resize_dim; nw = (int)((double)resize_dim * w / h); }
      center_crop_origin(nw, nh, crop_dim, &left, &top);
    } else {
      nw = nh = resize_dim;
    }
    PrepImage& im = imgs[i];
    im.src = offsets[i]; im.W = w; im.H = h;
    im.x0 = left < 0 ? -left : 0; im.y0 = top < 0 ? -top : 0;
    const int cx = left + im.x0, cy = top + im.y0;              // first resized column / row inside the window
    im.nx = std::min(S - im.x0, nw - cx); im.ny = std::min(S - im.y0, nh - cy);
    if (im.nx < 0) im.nx = 0;
    if (im.ny < 0) im.ny = 0;
    im.r0 = 0; im.nr = 0; im.kh = im.kv = 0; im.coef_h = im.coef_v = im.bnd_h = im.bnd_v = 0; im.tmp = (int64_t)tmp_bytes;
    if (im.nx > 0 && im.ny > 0) {
      pil_axis_table(h, nh, cy, im.ny, ax);                      // vertical first: it tells which source rows are needed
      int r0 = h, r1 = 0;
      for (int y = 0; y < im.ny; ++y) { r0 = std::min(r0, ax.bounds[2 * y]); r1 = std::max(r1, ax.bounds[2 * y] + ax.bounds[2 * y + 1]); }
      for (int y = 0; y < im.ny; ++y) ax.bounds[2 * y] -= r0;
      im.r0 = r0; im.nr = r1 - r0; im.kv = ax.ksize;
      im.coef_v = (int32_t)tables.size(); tables.insert(tables.end(), ax.kk.begin(), ax.kk.end());
      im.bnd_v = (int32_t)tables.size(); tables.insert(tables.end(), ax.bounds.begin(), ax.bounds.end());
      pil_axis_table(w, nw, cx, im.nx, ax);
      im.kh = ax.ksize;
      im.coef_h = (int32_t)tables.size(); tables.insert(tables.end(), ax.kk.begin(), ax.kk.end());
      im.bnd_h = (int32_t)tables.size(); tables.insert(tables.end(), ax.bounds.begin(), ax.bounds.end());
      tmp_bytes += (size_t)im.nr * im.nx * 3;
      tmp_bytes = (tmp_bytes + 15) & ~(size_t)15;
      max_tmp_elems = std::max(max_tmp_elems, im.nr * im.nx);
    }
  }
  // ---- staging: [PrepImage x B | tables] through one pinned slot of the ring, one asynchronous copy ----
  const size_t img_bytes = (size_t)B * sizeof(PrepImage), tab_bytes = tables.size() * sizeof(int32_t);
  const size_t need = img_bytes + tab_bytes + 16;
  pio_context::PrepSlot& sl = c->prep_slots[c->prep_next];
  c->prep_next = (c->prep_next + 1) % 4;
  if (sl.done) HIP_OK(hipEventSynchronize(sl.done));            // the launch that last used this slot has finished
  else HIP_OK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  if (sl.cap < need) {
    if (sl.host) HIP_OK(hipHostFree(sl.host));
    if (sl.dev) HIP_OK(hipFree(sl.dev));
    sl.cap = need * 2;
    HIP_OK(hipHostMalloc(&sl.host, sl.cap, hipHostMallocDefault));
    HIP_OK(hipMalloc(&sl.dev, sl.cap));
  }
  if (c->prep_tmp_cap < tmp_bytes) {
    HIP_OK(hipStreamSynchronize(s));                             // nobody is still reading the old intermediate image
    if (c->prep_tmp) HIP_OK(hipFree(c->prep_tmp));
    c->prep_tmp_cap = tmp_bytes * 2;
    HIP_OK(hipMalloc((void**)&c->prep_tmp, c->prep_tmp_cap));
  }
  memcpy(sl.host, imgs.data(), img_bytes);
  memcpy((char*)sl.host + img_bytes, tables.data(), tab_bytes);
  HIP_OK(hipMemcpyAsync(sl.dev, sl.host, img_bytes + tab_bytes, hipMemcpyHostToDevice, s));
  HIP_OK(launch_preprocess((const uint8_t*)pixels, (const PrepImage*)sl.dev, (const int32_t*)((const char*)sl.dev + img_bytes),
                           c->prep_tmp, c->prep_lut, B, S, max_tmp_elems, out, s));
  HIP_OK(hipEventRecord(sl.done, s));
  return PIO_OK;
}

int pio_num_tokens(pio_handle c) { return c ? c->T : 0; }
int pio_grid_side(pio_handle c) { return c ? c->n : 0; }
int64_t pio_bank_rows(pio_handle c) { return c ? c->bank_rows : 0; }

/* host-only helper exported for the CPU tests: DINOv2 interpolate_pos_encoding (bicubic, antialias) */
int pio_host_interpolate_pos_embed(const float* pos, int32_t grid, int32_t dim, int32_t n, float* out) {
  if (!pos || !out || grid < 1 || dim < 1 || n < 1) return fail(PIO_ERR_INVALID_ARG, "pio_host_interpolate_pos_embed: bad argument");
  interpolate_pos_embed(pos, grid, dim, n, out);
  return PIO_OK;
}

int pio_host_interpolate_pos_embed_plain(const float* pos, int32_t grid, int32_t dim, int32_t n, double offset, float* out) {
  if (!pos || !out || grid < 1 || dim < 1 || n < 1) return fail(PIO_ERR_INVALID_ARG, "pio_host_interpolate_pos_embed_plain: bad argument");
  interpolate_pos_embed_plain(pos, grid, dim, n, offset, out);
  return PIO_OK;
}

int pio_host_pil_ksize(int32_t in_size, int32_t out_size) {
  if (in_size < 1 || out_size < 1) return fail(PIO_ERR_INVALID_ARG, "pio_host_pil_ksize: bad size");
  PilAxis ax;
  pil_axis_table(in_size, out_size, 0, 0, ax);
  return ax.ksize;
}

int pio_host_pil_table(int32_t in_size, int32_t out_size, int32_t first, int32_t count, int32_t* kk, int32_t* bounds) {
  if (in_size < 1 || out_size < 1 || first < 0 || count < 0 || first + count > out_size || !kk || !bounds)
    return fail(PIO_ERR_INVALID_ARG, "pio_host_pil_table: bad argument");
  PilAxis ax;
  pil_axis_table(in_size, out_size, first, count, ax);
  memcpy(kk, ax.kk.data(), ax.kk.size() * sizeof(int32_t));
  memcpy(bounds, ax.bounds.data(), ax.bounds.size() * sizeof(int32_t));
  return PIO_OK;
}

}  // extern "C"
