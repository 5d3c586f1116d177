"""tests/golden/viecap_beam.npz: the REFERENCE's ``beam_search`` (P/src/viecap/search.py:193-285) called as ``VieCap.forward``
calls it when ``using_greedy_search`` is false (entrypoint.py:143-148: one call per image, beam_width 5), on the prompts of
the seeded ViECap case (hard prompt + soft prompt first, as tools/oracle/gen_golden_viecap.py builds them) and the same seeded
GPT-2 / byte-level BPE stand-ins.  Two runs per image: the reference's own end-of-sentence strings (".", " ."), which seeded
weights never emit, and two strings whose last token the seeded model does emit, so that the stop / length bookkeeping of
search.py:251-278 is exercised.  Stored per call: the ids every beam decodes to (the function only returns strings: the
tokenizer's decode is recorded) and the sentences in the order returned.
    python tools/oracle/gen_golden_viecap_beam.py"""
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
mapping.load_state_dict({k[len("mapping_network."):]: v for k, v in w.items() if k.startswith("mapping_network.")}, strict=True)
gpt = GPT2LMHeadModel(GPT2Config(n_layer=c["gpt_layers"], attn_implementation="eager")).eval()
missing, unexpected = gpt.load_state_dict({k[len("gpt."):]: v for k, v in w.items() if k.startswith("gpt.")}, strict=False)
assert not unexpected and all(".attn.bias" in m or ".attn.masked_bias" in m for m in missing), (missing, unexpected)

recorded = []
_decode = tok.decode
tok.decode = lambda ids: (recorded.append([int(t) for t in ids]), _decode([int(t) for t in ids]))[1]


def prompts(image_features):
    """entrypoint.py:108-135 (hard prompt, soft prompt first)"""
    pad_id = tok.pad_token_id if tok.pad_token_id is not None else 0
    image_features /= image_features.norm(2, dim=-1, keepdim=True)
    cont = mapping(image_features).view(-1, 10, 768)
    logits = retrieval.image_text_simiarlity(emb.clone(), temperature=c["temperature"], images_features=image_features)
    all_tokens = []
    for i in range(image_features.shape[0]):
        detected, _ = retrieval.top_k_categories(ents, logits[i:i + 1], c["top_k"], c["threshold"])
        all_tokens.append(utils.compose_discrete_prompts(tok, detected[0]))
    discrete_tokens = pad_sequence(all_tokens, batch_first=True, padding_value=pad_id)
    return torch.cat((cont, gpt.transformer.wte(discrete_tokens)), dim=1)


NIMG = 3
embeddings = prompts(x[:NIMG].clone())
out, meta = {}, {"calls": []}


def run(label, eos):
    beams_all = []
    for i in range(NIMG):
        recorded.clear()
        sentences = search.beam_search(embeddings=embeddings[i:i + 1], tokenizer=tok, beam_width=5, model=gpt, end_of_sentences=list(eos))
        beams = [list(r) for r in recorded]            # in beam order (before the final sort)
        L = max(len(b) for b in beams)
        out["%s_%d_ids" % (label, i)] = np.array([b + [-1] * (L - len(b)) for b in beams], dtype=np.int32)
        meta["calls"].append({"label": label, "image": i, "sentences": sentences, "lengths": [len(b) for b in beams]})
        print(label, i, [len(b) for b in beams], repr(sentences[0][:50]))
        beams_all.append(beams)
    return beams_all


default = run("default", [".", " ."])
# two "end of sentence" strings whose last token the seeded model DOES emit inside its beams (positions 6.. of images 0 and 1), so
# that beams stop at different steps
def pick(beams, start):
    for pos in range(start, 40):
        for b in beams:
            t = b[pos]
            d = _decode([t])
            if "\ufffd" not in d and tok.encode(d)[-1] == t:
                return d
    raise SystemExit("no round-tripping token found")


eos_strings = [pick(default[0], 6), pick(default[1], 12)]
meta["eos_strings"] = eos_strings
run("seeded_eos", eos_strings)
out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
path = os.path.join(ROOT, "tests", "golden", "viecap_beam.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KB")
