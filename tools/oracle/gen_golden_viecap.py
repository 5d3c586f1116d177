"""tests/golden/viecap.npz: the REFERENCE's ViECap pieces (through refshim) on seeded inputs -- ``MappingNetwork``
(P/src/viecap/ClipCap.py:122-153), ``image_text_simiarlity`` / ``top_k_categories`` (retrieval_categories.py:61-116),
``compose_discrete_prompts`` (utils.py:55-74) and ``greedy_search`` (search.py:108-191) wired exactly as ``VieCap.forward``
wires them (entrypoint.py:98-153, hard prompt, soft prompt first, greedy search: the shipped config), on a
``GPT2LMHeadModel(GPT2Config())`` that carries OUR seeded weights (the pretrained 'gpt2' needs network) and OUR seeded
byte-level BPE vocabulary as the tokenizer object (same reason).  Also a single-feature call (the str-returning path) and
``VieCap.compute_perplexity`` (entrypoint.py:155-172) of the generated sentences.
    python tools/oracle/gen_golden_viecap.py"""
import json
import os
import sys

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_cases as gc  # noqa: E402
import refshim  # noqa: E402
from patchioner_amd.viecap import ByteLevelBPE  # noqa: E402

torch.set_grad_enabled(False)
refshim.load()
import importlib  # noqa: E402

clipcap = importlib.import_module("refsrc.viecap.ClipCap")
search = importlib.import_module("refsrc.viecap.search")
retrieval = importlib.import_module("refsrc.viecap.retrieval_categories")
utils = importlib.import_module("refsrc.viecap.utils")
from transformers import GPT2Config, GPT2LMHeadModel  # noqa: E402

c = gc.VIECAP
w, (vocab, merges), ents, emb, x = gc.viecap_case()
tok = ByteLevelBPE(vocab, merges)

mapping = clipcap.MappingNetwork(10, c["C"], 10, 768, 8, 8)
missing, unexpected = mapping.load_state_dict({k[len("mapping_network."):]: v for k, v in w.items() if k.startswith("mapping_network.")}, strict=True)
gpt = GPT2LMHeadModel(GPT2Config(n_layer=c["gpt_layers"], attn_implementation="eager")).eval()
missing, unexpected = gpt.load_state_dict({k[len("gpt."):]: v for k, v in w.items() if k.startswith("gpt.")}, strict=False)
assert not unexpected and all(".attn.bias" in m or ".attn.masked_bias" in m for m in missing), (missing, unexpected)

recorded = []
_decode = tok.decode
tok.decode = lambda ids: (recorded.append(list(ids)), _decode(ids))[1]      # greedy_search only returns strings: keep the ids it decodes


def viecap_forward(image_features):
    """VieCap.forward, entrypoint.py:98-153 (hard prompt, soft prompt first, greedy search)."""
    pad_id = tok.pad_token_id if tok.pad_token_id is not None else 0
    image_features /= image_features.norm(2, dim=-1, keepdim=True)
    cont = mapping(image_features).view(-1, 10, 768)
    logits = retrieval.image_text_simiarlity(emb.clone(), temperature=c["temperature"], images_features=image_features)
    all_tokens, detected_all = [], []
    for i in range(image_features.shape[0]):
        detected, _ = retrieval.top_k_categories(ents, logits[i:i + 1], c["top_k"], c["threshold"])
        detected_all.append(detected[0])
        all_tokens.append(utils.compose_discrete_prompts(tok, detected[0]))
    discrete_tokens = pad_sequence(all_tokens, batch_first=True, padding_value=pad_id)
    discrete = gpt.transformer.wte(discrete_tokens)
    embeddings = torch.cat((cont, discrete), dim=1)
    sentences = search.greedy_search(embeddings=embeddings, tokenizer=tok, model=gpt)
    return cont, logits, detected_all, discrete_tokens, sentences


out = {}
recorded.clear()
cont, logits, detected, dtok, sentences = viecap_forward(x.clone())
out["cont"] = cont.numpy()
out["entity_probs"] = logits.numpy()
out["prompt_tokens"] = dtok.numpy().astype(np.int32)
out["decoded_ids"] = np.array([r + [-1] * (64 - len(r)) for r in recorded], dtype=np.int32)      # cut at the first full stop
meta = {"detected": detected, "sentences": sentences}
recorded.clear()
single = viecap_forward(x[1:2].clone())
assert isinstance(single[4], str)
meta["single_sentence"] = single[4]
out["single_decoded_ids"] = np.array(recorded[0], dtype=np.int32)
# VieCap.compute_perplexity (entrypoint.py:155-172) on the sentences above: the method uses nothing of ``self``
entry = importlib.import_module("refsrc.viecap.entrypoint")


class _Enc(dict):
    def to(self, device):
        return self

    def __getattr__(self, k):
        return self[k]


class _CallableTok:
    """the HF call convention compute_perplexity uses: tokenizer(sentence, return_tensors='pt') -> input_ids, attention_mask"""

    def __call__(self, sentence, return_tensors="pt"):
        ids = torch.tensor([tok.encode(sentence)], dtype=torch.long)
        return _Enc(input_ids=ids, attention_mask=torch.ones_like(ids))


ppl = entry.VieCap.compute_perplexity(None, sentences, tokenizer=_CallableTok(), model=gpt, device="cpu")
out["perplexity"] = np.array(ppl, dtype=np.float64)
meta["perplexity_sentences"] = sentences
print("perplexities", ppl)
out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
path = os.path.join(ROOT, "tests", "golden", "viecap.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KB")
print(detected)
print([len(r) for r in out["decoded_ids"]], sentences[:2])
