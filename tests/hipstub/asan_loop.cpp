// Drives the C ABI's host side through the create / load / finalize / bank / clone / decode / preprocess / destroy orders
// that crashed an in-process loop on the GPU in round 1 ("Open observations"), on the HIP stub, under
// AddressSanitizer + LeakSanitizer.  Exit code 0 = no finding (ASan aborts otherwise).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/patchioner_hip.h"

#define CK(x) do { int _rc = (x); if (_rc != PIO_OK) { printf("%s -> %d: %s\n", #x, _rc, pio_last_error()); exit(2); } } while (0)
#define EXPECT_FAIL(x) do { int _rc = (x); if (_rc == PIO_OK) { printf("%s unexpectedly succeeded\n", #x); exit(3); } } while (0)

static void load(pio_handle h, const std::string& key, std::vector<int64_t> shape) {
  int64_t n = 1;
  for (auto s : shape) n *= s;
  std::vector<float> v((size_t)n, 0.01f);
  CK(pio_load_weight(h, key.c_str(), v.data(), shape.data(), (int32_t)shape.size()));
}

static pio_handle make(bool viecap) {
  pio_config c;
  memset(&c, 0, sizeof(c));
  c.embed_dim = 64; c.depth = 1; c.num_heads = 1; c.patch_size = 14; c.num_registers = 4; c.pretrain_grid = 3; c.crop_dim = 28;
  c.vit_ln_eps = 1e-6f; c.readout_heads = 1; c.readout_scale = 0.125f; c.dec_layers = 1; c.dec_heads = viecap ? 12 : 4; c.dec_embd = 768;
  c.dec_vocab = 64; c.dec_positions = 160; c.prefix_size = 64; c.dec_ln_eps = 1e-5f; c.max_batch = 2; c.max_prefixes = 4;
  c.max_steps = viecap ? 128 : 30; c.vit_operand_type = 0; c.device = 0;
  pio_handle h = nullptr;
  CK(pio_create(&c, &h));
  const int64_t D = 64, E = 768;
  load(h, "cls_token", {1, 1, D}); load(h, "pos_embed", {1, 10, D}); load(h, "register_tokens", {1, 4, D});
  load(h, "patch_embed.proj.weight", {D, 3, 14, 14}); load(h, "patch_embed.proj.bias", {D}); load(h, "norm.weight", {D}); load(h, "norm.bias", {D});
  const char* b = "blocks.0.";
  for (const char* k : {"norm1.weight", "norm1.bias", "attn.proj.bias", "ls1.gamma", "norm2.weight", "norm2.bias", "mlp.fc2.bias", "ls2.gamma"})
    load(h, std::string(b) + k, {D});
  load(h, std::string(b) + "attn.qkv.bias", {3 * D}); load(h, std::string(b) + "mlp.fc1.bias", {4 * D});
  load(h, std::string(b) + "attn.qkv.weight", {3 * D, D}); load(h, std::string(b) + "attn.proj.weight", {D, D});
  load(h, std::string(b) + "mlp.fc1.weight", {4 * D, D}); load(h, std::string(b) + "mlp.fc2.weight", {D, 4 * D});
  const std::string p = viecap ? "gpt.transformer." : "decoder.transformer.";
  if (!viecap) { load(h, "clip_project.model.0.weight", {E, 64}); load(h, "clip_project.model.0.bias", {E}); }
  load(h, p + "wte.weight", {64, E}); load(h, p + "wpe.weight", {160, E}); load(h, p + "ln_f.weight", {E}); load(h, p + "ln_f.bias", {E});
  const std::string l = p + "h.0.";
  load(h, l + "ln_1.weight", {E}); load(h, l + "ln_1.bias", {E}); load(h, l + "ln_2.weight", {E}); load(h, l + "ln_2.bias", {E});
  load(h, l + "attn.c_attn.weight", {E, 3 * E}); load(h, l + "attn.c_attn.bias", {3 * E}); load(h, l + "attn.c_proj.weight", {E, E});
  load(h, l + "attn.c_proj.bias", {E}); load(h, l + "mlp.c_fc.weight", {E, 4 * E}); load(h, l + "mlp.c_fc.bias", {4 * E});
  load(h, l + "mlp.c_proj.weight", {4 * E, E}); load(h, l + "mlp.c_proj.bias", {E});
  if (viecap) {
    load(h, "mapping_network.linear.weight", {2 * E, 64}); load(h, "mapping_network.linear.bias", {2 * E}); load(h, "mapping_network.prefix_const", {3, E});
    const std::string m = "mapping_network.transformer.layers.0.";
    for (const char* k : {"norm1.weight", "norm1.bias", "attn.project.bias", "norm2.weight", "norm2.bias", "mlp.fc2.bias"}) load(h, m + k, {E});
    load(h, m + "attn.to_queries.weight", {E, E}); load(h, m + "attn.to_keys_values.weight", {2 * E, E}); load(h, m + "attn.project.weight", {E, E});
    load(h, m + "mlp.fc1.weight", {64, E}); load(h, m + "mlp.fc1.bias", {64}); load(h, m + "mlp.fc2.weight", {E, 64});
  }
  EXPECT_FAIL(pio_load_weight(h, "no.such.key", nullptr, nullptr, 0));
  CK(pio_finalize_weights(h));
  EXPECT_FAIL(pio_finalize_weights(h));
  return h;
}

// The CLIP ViT variant (vit_arch 1: no registers, native grid, norm_pre, bias-free patch conv, no LayerScale, head) in the
// exact-fp32 parity mode (vit_operand_type 2: fp32 weight copies and workspaces): load, finalize, forward, misuse, destroy.
static void clip_fp32_round() {
  pio_config c;
  memset(&c, 0, sizeof(c));
  c.embed_dim = 64; c.depth = 1; c.num_heads = 1; c.patch_size = 16; c.num_registers = 0; c.pretrain_grid = 2; c.crop_dim = 32;
  c.vit_ln_eps = 1e-5f; c.readout_heads = 1; c.readout_scale = 0.125f; c.dec_layers = 1; c.dec_heads = 4; c.dec_embd = 768;
  c.dec_vocab = 64; c.dec_positions = 160; c.prefix_size = 32; c.dec_ln_eps = 1e-5f; c.max_batch = 2; c.max_prefixes = 4;
  c.max_steps = 30; c.vit_operand_type = 2; c.device = 0; c.vit_arch = 1; c.vit_out_dim = 32;
  pio_handle bad = nullptr;
  pio_config w = c;
  w.num_registers = 4;
  EXPECT_FAIL(pio_create(&w, &bad));                                   // the CLIP ViT has no registers
  w = c; w.vit_arch = 0;
  EXPECT_FAIL(pio_create(&w, &bad));                                   // a head width other than embed_dim needs vit_arch 1
  pio_handle h = nullptr;
  CK(pio_create(&c, &h));
  const int64_t D = 64;
  load(h, "cls_token", {1, 1, D}); load(h, "pos_embed", {1, 5, D}); load(h, "patch_embed.proj.weight", {D, 3, 16, 16});
  load(h, "norm.weight", {D}); load(h, "norm.bias", {D}); load(h, "norm_pre.weight", {D}); load(h, "norm_pre.bias", {D});
  load(h, "head.weight", {32, D});
  const char* b = "blocks.0.";
  for (const char* k : {"norm1.weight", "norm1.bias", "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc2.bias"}) load(h, std::string(b) + k, {D});
  load(h, std::string(b) + "attn.qkv.bias", {3 * D}); load(h, std::string(b) + "mlp.fc1.bias", {4 * D});
  load(h, std::string(b) + "attn.qkv.weight", {3 * D, D}); load(h, std::string(b) + "attn.proj.weight", {D, D});
  load(h, std::string(b) + "mlp.fc1.weight", {4 * D, D}); load(h, std::string(b) + "mlp.fc2.weight", {D, 4 * D});
  CK(pio_finalize_weights(h));
  std::vector<float> imgs(2 * 3 * 32 * 32, 0.1f), tok(2 * 5 * 32), qkv(2 * 5 * 192), sa(2 * 4);
  CK(pio_vit_forward(h, imgs.data(), 2, tok.data(), nullptr, nullptr));
  EXPECT_FAIL(pio_vit_forward(h, imgs.data(), 2, tok.data(), qkv.data(), nullptr));          // no qkv capture on this backbone
  EXPECT_FAIL(pio_cls_attention(h, qkv.data(), tok.data(), 2, sa.data(), nullptr, nullptr, nullptr, nullptr));
  CK(pio_destroy(h));
}

int main() {
  for (int it = 0; it < 12; ++it) {
    if (it < 3) clip_fp32_round();
    pio_handle h = make(false);
    std::vector<float> bank(16 * 384, 0.5f);
    for (int d = 0; d < 384; ++d) bank[5 * 384 + d] = 0.f;             // one zero row: dropped at load
    int64_t kept = 0;
    CK(pio_set_memory_bank(h, bank.data(), 16, 384, &kept));
    if (kept != 15) { printf("kept %lld\n", (long long)kept); return 4; }
    EXPECT_FAIL(pio_set_memory_bank(h, bank.data(), 16, 384, &kept));   // already set
    pio_handle c1 = nullptr, c2 = nullptr, bad = nullptr;
    CK(pio_clone_decoder(h, &c1));
    CK(pio_clone_decoder(h, &c2));
    EXPECT_FAIL(pio_clone_decoder(c1, &bad));                            // clones of clones are refused
    EXPECT_FAIL(pio_destroy(h));                                         // the owner outlives its clones
    std::vector<float> prefix(4 * 64, 0.1f), tok(2 * 9 * 64), qkv(2 * 9 * 192);
    std::vector<int32_t> ids(4 * 30);
    std::vector<float> lp(4 * 30), imgs(2 * 3 * 28 * 28);
    for (pio_handle d : {h, c1, c2}) {
      CK(pio_decode_greedy(d, prefix.data(), 3, 30, ids.data(), nullptr, nullptr));
      CK(pio_decode_greedy(d, prefix.data(), 4, 30, ids.data(), lp.data(), nullptr));     // a second graph key
      CK(pio_decode_greedy(d, prefix.data(), 3, 30, ids.data(), nullptr, nullptr));     // replay
      EXPECT_FAIL(pio_decode_greedy(d, prefix.data(), 5, 30, ids.data(), nullptr, nullptr));
    }
    CK(pio_vit_forward(h, imgs.data(), 2, tok.data(), qkv.data(), nullptr));
    CK(pio_profile_enable(h, 1));
    CK(pio_vit_forward(h, imgs.data(), 1, tok.data(), nullptr, nullptr));
    double ms, fl, by; int64_t n;
    CK(pio_profile_read(h, 0, &ms, &n, &fl, &by));
    CK(pio_profile_enable(h, 0));
    void* s1 = nullptr; void* s2 = nullptr;
    CK(pio_stream_create(0, 0, 192, &s1));
    CK(pio_stream_create(0, 192, 64, &s2));
    CK(pio_decode_greedy(c1, prefix.data(), 2, 30, ids.data(), nullptr, s1));
    CK(pio_stream_destroy(s1));
    CK(pio_stream_destroy(s2));
    // teardown in the order that matters: clones (either order), then the owner
    if (it & 1) { CK(pio_destroy(c1)); CK(pio_destroy(c2)); } else { CK(pio_destroy(c2)); CK(pio_destroy(c1)); }
    CK(pio_destroy(h));
    // ViECap head: mapping network + entity table + prompted decode graphs
    pio_handle v = make(true);
    std::vector<float> ent(5 * 64, 0.3f), feats(4 * 64, 0.2f), cont(4 * 3 * 768), probs(4 * 5);
    std::vector<int32_t> toks(4 * 6, 1), vids(4 * 64);
    CK(pio_viecap_set_entities(v, ent.data(), 5, 64));
    EXPECT_FAIL(pio_viecap_set_entities(v, ent.data(), 5, 64));
    CK(pio_viecap_mapping(v, feats.data(), 4, cont.data(), nullptr));
    CK(pio_viecap_entity_logits(v, feats.data(), 4, 0.01f, probs.data(), nullptr));
    CK(pio_viecap_decode(v, cont.data(), toks.data(), 4, 6, 1, 64, vids.data(), nullptr));
    CK(pio_viecap_decode(v, cont.data(), toks.data(), 2, 4, 1, 64, vids.data(), nullptr));
    CK(pio_viecap_decode(v, cont.data(), nullptr, 2, 0, 1, 64, vids.data(), nullptr));
    EXPECT_FAIL(pio_viecap_decode(v, cont.data(), toks.data(), 4, 100, 1, 64, vids.data(), nullptr));   // positions above max_steps
    CK(pio_viecap_decode(v, nullptr, toks.data(), 2, 4, 0, 64, vids.data(), nullptr));                  // only_hard_prompt
    EXPECT_FAIL(pio_viecap_decode(v, nullptr, nullptr, 2, 0, 1, 64, vids.data(), nullptr));             // neither kind of prompt
    // beam search building blocks: prompt assembly, prefill, re-ordered advance, selection
    {
      const int vocab = 64;                       // dec_vocab of this configuration; max_prefixes = 4 beams
      std::vector<float> prompt(4 * 9 * 768), logp((size_t)4 * vocab), sc(4, -1.f), ln(4, 1.f), val(4);
      std::vector<int32_t> nt(4, 7), src{0, 0, 1, 3}, st(4, 0);
      std::vector<int64_t> idx(4);
      CK(pio_viecap_build_prompt(v, cont.data(), toks.data(), 4, 6, 1, prompt.data(), nullptr));
      CK(pio_viecap_build_prompt(v, nullptr, toks.data(), 4, 6, 0, prompt.data(), nullptr));
      EXPECT_FAIL(pio_viecap_build_prompt(v, nullptr, nullptr, 4, 0, 0, prompt.data(), nullptr));
      CK(pio_lm_prefill(v, prompt.data(), 4, 9, logp.data(), nullptr));
      CK(pio_beam_select(v, logp.data(), nullptr, nullptr, nullptr, 4, val.data(), idx.data(), nullptr));
      CK(pio_lm_advance(v, nt.data(), nullptr, 4, 9, logp.data(), nullptr));
      CK(pio_lm_advance(v, nt.data(), src.data(), 4, 10, logp.data(), nullptr));
      CK(pio_beam_select(v, logp.data(), sc.data(), ln.data(), st.data(), 4, val.data(), idx.data(), nullptr));
      EXPECT_FAIL(pio_beam_select(v, logp.data(), sc.data(), nullptr, st.data(), 4, val.data(), idx.data(), nullptr));
      EXPECT_FAIL(pio_beam_select(v, logp.data(), nullptr, nullptr, nullptr, 9, val.data(), idx.data(), nullptr));
      EXPECT_FAIL(pio_lm_prefill(v, prompt.data(), 5, 9, logp.data(), nullptr));        // above max_prefixes
      EXPECT_FAIL(pio_lm_advance(v, nt.data(), nullptr, 4, 4096, logp.data(), nullptr));
    }
    EXPECT_FAIL(pio_decode_greedy(v, prefix.data(), 2, 30, ids.data(), nullptr, nullptr));             // no clip_project in a ViECap model
    CK(pio_destroy(v));
  }
  CK(pio_destroy(nullptr));
  printf("asan loop ok\n");
  return 0;
}
