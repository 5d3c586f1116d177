"""ViECap head of the reference (``VieCap``, P/src/viecap/entrypoint.py:16-153), host side.

``Patchioner.caption_tokens`` hands the region features to ``self.viecap.forward`` (P/src/model.py:1394-1398).  Everything
numerical is a C-ABI call (mapping network, entity logits, prompt assembly, KV-cached greedy search: viecap.hip /
decoder.hip); what stays on the host is what is host work in the reference too: top-k / threshold over the 80-odd entity
probabilities, composing the hard-prompt STRING, tokenising it, cutting the generated ids at the first full stop and
detokenising.

There is no network on the target, so nothing is fetched from the HuggingFace hub: the GPT-2 tokenizer comes from local
``vocab.json`` / ``merges.txt`` (``viecap.tokenizer_path``) or an object with ``encode`` / ``decode`` / ``pad_token_id``
(``viecap.tokenizer``); the checkpoint (``mapping_network.*`` + ``gpt.*``) from ``viecap.weight_path`` or a state dict
(``viecap.weights``); the entity vocabulary from the reference's own files under ``files_path`` or
``viecap.entities_text`` / ``viecap.texts_embeddings``.
"""
from __future__ import annotations

import json
import os
import pickle
from argparse import Namespace
from typing import Dict, List, Optional, Sequence

import torch


# ------------------------------------------------------------------------------------------------ GPT-2 byte-level BPE
def _bytes_to_unicode() -> Dict[int, str]:
    """The printable stand-ins GPT-2's byte-level BPE uses for the 256 byte values (public algorithm)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


class ByteLevelBPE:
    """GPT-2's tokenizer (``AutoTokenizer.from_pretrained('gpt2')`` in the reference, entrypoint.py:40) over a local
    vocabulary: ``vocab`` token string -> id, ``merges`` ranked pairs.  ``encode`` / ``decode`` as the reference uses them
    (utils.py:72, search.py:141,176-190); ``pad_token_id`` is None for GPT-2, so the reference pads with id 0."""

    PATTERN = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""

    def __init__(self, vocab: Dict[str, int], merges: Sequence[Sequence[str]]):
        import regex
        self.encoder = dict(vocab)
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.ranks = {tuple(m): r for r, m in enumerate(merges)}
        self.b2u = _bytes_to_unicode()
        self.u2b = {u: b for b, u in self.b2u.items()}
        self.pat = regex.compile(self.PATTERN)
        self.cache: Dict[str, List[str]] = {}
        self.pad_token_id = None

    @classmethod
    def from_files(cls, path: str) -> "ByteLevelBPE":
        with open(os.path.join(path, "vocab.json"), encoding="utf-8") as f:
            vocab = json.load(f)
        with open(os.path.join(path, "merges.txt"), encoding="utf-8") as f:
            lines = [ln.rstrip("\n") for ln in f if ln.strip() and not ln.startswith("#version")]
        return cls(vocab, [tuple(ln.split(" ")) for ln in lines])

    def __len__(self):
        return len(self.encoder)

    def _bpe(self, word: str) -> List[str]:
        if word in self.cache:
            return self.cache[word]
        parts = list(word)
        while len(parts) > 1:
            best, at = None, -1
            for i in range(len(parts) - 1):
                r = self.ranks.get((parts[i], parts[i + 1]))
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            a, b = parts[at], parts[at + 1]
            out, i = [], 0
            while i < len(parts):                 # merge every occurrence of the best-ranked pair, left to right
                if i < len(parts) - 1 and parts[i] == a and parts[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(parts[i])
                    i += 1
            parts = out
        self.cache[word] = parts
        return parts

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for piece in self.pat.findall(text):
            word = "".join(self.b2u[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(word))
        return ids

    def decode(self, ids: Sequence[int]) -> str:
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.u2b[c] for c in text).decode("utf-8", errors="replace")


# ------------------------------------------------------------------------------------------------ prompt composition
def compose_discrete_prompt_text(entities: Sequence[str]) -> str:
    """compose_discrete_prompts (P/src/viecap/utils.py:55-74), the string before tokenisation."""
    if len(entities) == 0:
        return "There are something in image."
    return "There are" + "".join(" " + e + "," for e in entities)[:-1] + " in image."


def top_k_entities(texts: Sequence[str], probs_row: Sequence[float], top_k: int, threshold: float) -> List[str]:
    """top_k_categories (P/src/viecap/retrieval_categories.py:97-116) for one image: torch.topk order, cut at the first
    probability below the threshold."""
    vals, idx = torch.topk(torch.as_tensor(probs_row), k=top_k, dim=-1)
    out = []
    for v, i in zip(vals.tolist(), idx.tolist()):
        if v < threshold:
            break
        out.append(texts[i])
    return out


DEFAULTS = {  # VieCap.defaults (entrypoint.py:61-80)
    "language_model": "gpt2", "continuous_prompt_length": 10, "clip_project_length": 10, "temperature": 0.01, "top_k": 3,
    "threshold": 0.2, "disable_all_entities": False, "name_of_entities_text": "vinvl_vgoi_entities", "prompt_ensemble": False,
    "weight_path": "/raid/datasets/viecap_files/checkpoints/train_coco/coco_prefix-0014.pt", "files_path": "/raid/datasets/viecap_files/",
    "using_hard_prompt": False, "soft_prompt_first": False, "only_hard_prompt": False, "using_greedy_search": False,
    "beam_width": 5, "text_prompt": None,
}


# name_of_entities_text -> (vocabulary file, its reader, stem of the pickled [K, C] embeddings)  (entrypoint.py:186-219;
# readers: load_annotations.py:78-150).  Every reader lower-cases / strips, optionally keeps single words only, and sorts.
def _read_vg(path):
    with open(path, "rb") as f:
        return list(pickle.load(f)["objects"]["joint"])


def _read_json_list(path):
    with open(path) as f:
        return list(json.load(f))


def _read_open_images(path):
    import csv
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    out = []
    for r in rows:                                   # load_annotations.py:110-115: "Bat (Animal)" -> "bat"
        e = r["DisplayName"].lower().strip()
        if e[-1] == ")":
            e = e[:e.find("(")].strip()
        out.append(e)
    return out


def _read_vinvl_vg(path):
    with open(path) as f:
        return list(json.load(f)["object_count"])


_VOCABULARIES = {
    "visual_genome_entities": ("all_objects_attributes_relationships.pickle", _read_vg, "visual_genome_embedding"),
    "coco_entities": ("coco_categories.json", _read_json_list, "coco_embeddings"),
    "open_image_entities": ("oidv7-class-descriptions-boxable.csv", _read_open_images, "open_image_embeddings"),
    "vinvl_vg_entities": ("VG-SGG-dicts-vgoi6-clipped.json", _read_vinvl_vg, "vg_embeddings"),
    "vinvl_vgoi_entities": ("vgcocooiobjects_v1_class2ind.json", _read_json_list, "vgoi_embeddings"),   # dict {name: index}: its keys
}


def _entity_files(args: Namespace, suffix: str):
    """get_viecap_texts_embeddings (entrypoint.py:179-223): the entity names of the selected vocabulary (default
    ``vinvl_vgoi_entities``) and their pickled text embeddings ``<stem>_<suffix>[_with_ensemble].pickle``.  The directory is
    ``files_path/annotations/vocabulary`` or, when that is missing, the ``vocabulary`` directory next to this file (the
    reference's fallback, entrypoint.py:183-185).  The embeddings must exist: computing them needs the CLIP text tower."""
    suffix = suffix.replace("/", "")
    vocab_dir = os.path.join(args.files_path, "annotations/vocabulary")
    if not os.path.exists(vocab_dir):
        vocab_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocabulary")
    if not os.path.exists(vocab_dir):
        raise FileNotFoundError("entity vocabulary directory not found under %r nor next to viecap.py (pass viecap.entities_text / "
                                "viecap.texts_embeddings or point viecap.files_path at the reference's files)" % args.files_path)
    if args.name_of_entities_text not in _VOCABULARIES:
        raise ValueError("The entities text should be input correctly! (%r)" % (args.name_of_entities_text,))      # entrypoint.py:220-222
    fname, reader, stem = _VOCABULARIES[args.name_of_entities_text]
    ents = [e.lower().strip() for e in reader(os.path.join(vocab_dir, fname))]
    if args.disable_all_entities:
        ents = [e for e in ents if len(e.split()) == 1]
    ents.sort()
    name = "%s_%s%s.pickle" % (stem, suffix, "_with_ensemble" if args.prompt_ensemble else "")
    path = os.path.join(vocab_dir, name)
    if not os.path.exists(path):
        raise FileNotFoundError("%r not found: the entity text embeddings are computed with the CLIP text tower (out of scope "
                                "here); provide the pickle or pass viecap.texts_embeddings" % path)
    with open(path, "rb") as f:
        emb = pickle.load(f)
    return ents, torch.as_tensor(emb).float()


class VieCapHead:
    """Mirror of ``VieCap`` (P/src/viecap/entrypoint.py:16): ``forward(image_features, compute_scores)`` -> captions."""

    def __init__(self, args: dict, engine, clip_name: Optional[str]):
        args_dict = dict(args)
        for k, v in DEFAULTS.items():
            args_dict.setdefault(k, v)
        self.args = a = Namespace(**args_dict)
        self.engine = engine
        self.device = engine.device
        self.clip_hidden_size = args_dict.get("clip_hidden_size") or (640 if "RN" in (clip_name or "") else 512)
        if "gpt" not in a.language_model:
            raise NotImplementedError("ViECap with an OPT language model (opt_search): outside the hot-path scope")
        if not a.using_greedy_search and not 1 <= int(a.beam_width) <= 8:
            raise ValueError("viecap.beam_width must be 1..8 (pio_beam_select)")
        if args_dict.get("entities_text") is not None:
            self.entities_text = list(args_dict["entities_text"])
            emb = torch.as_tensor(args_dict["texts_embeddings"]).float()
        else:
            self.entities_text, emb = _entity_files(a, args_dict.get("suffix") or (clip_name or ""))
        if emb.shape != (len(self.entities_text), self.clip_hidden_size):
            raise ValueError("entity embeddings are %s, expected [%d, %d]" % (tuple(emb.shape), len(self.entities_text), self.clip_hidden_size))
        tok = args_dict.get("tokenizer")
        if tok is None:
            path = args_dict.get("tokenizer_path")
            if not path:
                raise FileNotFoundError("viecap.tokenizer_path (a directory with GPT-2's vocab.json and merges.txt) is required: "
                                        "AutoTokenizer.from_pretrained('gpt2') needs network, which this target does not have")
            tok = ByteLevelBPE.from_files(path)
        self.tokenizer = tok
        engine.viecap_set_entities(emb)
        # eos ids as greedy_search takes them (search.py:141): the LAST token of "." and of " ."
        self.eos = [self.tokenizer.encode(e)[-1] for e in (".", " .")]

    def forward(self, image_features: torch.Tensor, compute_scores: bool = False):
        """entrypoint.py:98-153.  image_features [N, clip_hidden_size] is L2-normalised IN PLACE (line 108)."""
        a, eng = self.args, self.engine
        pad_id = self.tokenizer.pad_token_id if self.tokenizer.pad_token_id is not None else 0
        x = image_features
        if not isinstance(x, torch.Tensor):
            x = torch.tensor(x, dtype=torch.float)
        xd = x.to(device=eng.device, dtype=torch.float32).contiguous()
        cont = eng.viecap_mapping(xd)                                   # normalises xd in place
        if isinstance(image_features, torch.Tensor) and xd.data_ptr() != image_features.data_ptr():
            image_features.copy_(xd)
        N = xd.shape[0]
        tokens = None
        if a.using_hard_prompt:
            probs = eng.viecap_entity_logits(xd, a.temperature).cpu()
            rows = []
            for i in range(N):
                ents = top_k_entities(self.entities_text, probs[i], a.top_k, a.threshold)
                rows.append(self.tokenizer.encode(compose_discrete_prompt_text(ents)))
            L = max(len(r) for r in rows)
            tokens = torch.full((N, L), pad_id, dtype=torch.int32)       # pad_sequence(batch_first=True, padding_value=pad_id)
            for i, r in enumerate(rows):
                tokens[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
            self.last_prompt_tokens = tokens
        if a.using_hard_prompt and a.only_hard_prompt:
            cont = None                                                  # entrypoint.py:130-131: the word embeddings alone
        soft_first = bool(a.soft_prompt_first) or tokens is None
        if not a.using_greedy_search:
            # entrypoint.py:143-148: one beam_search call per element of the batch, the best beam's sentence of each
            prompts = eng.viecap_build_prompt(cont, tokens, soft_first=soft_first)
            out = [self.beam_search(prompts[i:i + 1], int(a.beam_width))[0] for i in range(N)]
            self.last_ids = None                  # no [N, 64] id tensor: the beams of an image have their own lengths
            if compute_scores:
                return out, self.compute_perplexity(out)
            return out
        ids = eng.viecap_decode(cont, tokens, soft_first=soft_first, steps=64)
        self.last_ids = ids
        rows = ids.cpu().tolist()
        if N == 1:
            # search.py:172-181: a single caption stops at its first full stop and comes back as a str, not a list
            r = rows[0]
            out = self.tokenizer.decode(r)
            for i, t in enumerate(r):
                if t in self.eos:
                    out = self.tokenizer.decode(r[:i + 1])
                    break
        else:
            out = []
            for r in rows:
                i = len(r) - 1
                for j, t in enumerate(r):
                    if t in self.eos:
                        i = j
                        break
                out.append(self.tokenizer.decode(r[:i + 1]))
        if compute_scores:
            return out, self.compute_perplexity(out)
        return out

    def beam_search(self, embeddings: torch.Tensor, beam_width: int = 5, max_len: int = 64, end_of_sentences=(".", " .")) -> List[str]:
        """``beam_search`` (search.py:193-285) for ONE prompt [1, P, E]: every beam's sentence, best first.  The language model and
        each selection run on the device (Engine.lm_prefill / lm_advance / beam_select, beams = rows with their own KV caches);
        what is left here is the reference's bookkeeping -- token lists, lengths, stop flags, fp32 score arithmetic -- line by line.
        ``self.last_beams``: (token ids, length, score) per beam in the order of the last selection."""
        import numpy as np
        eng = self.engine
        W = int(beam_width)
        P = int(embeddings.shape[1])
        V = eng.cfg.dec_vocab
        eos = [self.tokenizer.encode(e)[-1] for e in end_of_sentences]
        assert len(eos) == 2                                                    # search.py:276
        lp = eng.lm_prefill(embeddings.expand(W, -1, -1).contiguous())          # the prompt, once per beam
        val, idx = eng.beam_select(lp)                                          # :247-249: scores, next_tokens = logits.topk(W)
        scores = val.numpy().astype(np.float32)
        next_tokens = idx.numpy().astype(np.int64)
        tokens = [[int(t)] for t in next_tokens]
        seq_lengths = np.ones(W, dtype=np.float32)
        is_stopped = np.zeros(W, dtype=bool)
        is_stopped = is_stopped | (next_tokens == eos[0]) | (next_tokens == eos[1])
        src = None
        for i in range(1, max_len):
            if is_stopped.all():
                break
            lp = eng.lm_advance(torch.from_numpy(next_tokens.astype(np.int32)), None if src is None else torch.from_numpy(src.astype(np.int32)),
                                pos=P + i - 1)
            val, idx = eng.beam_select(lp, torch.from_numpy(scores), torch.from_numpy(seq_lengths), torch.from_numpy(is_stopped.astype(np.int32)))
            seq_lengths[~is_stopped] += 1                                       # :257
            avg, flat = val.numpy().astype(np.float32), idx.numpy().astype(np.int64)
            src = flat // V                                                     # :261 next_tokens_source
            seq_lengths = seq_lengths[src]
            next_tokens = flat % V
            tokens = [tokens[int(b)] + [int(t)] for b, t in zip(src, next_tokens)]
            scores = (avg * seq_lengths).astype(np.float32)                     # :269
            is_stopped = is_stopped[src]
            is_stopped = is_stopped | (next_tokens == eos[0]) | (next_tokens == eos[1])
        scores = (scores / seq_lengths).astype(np.float32)
        texts = [self.tokenizer.decode(t[:int(n)]) for t, n in zip(tokens, seq_lengths)]
        order = torch.from_numpy(scores).argsort(descending=True).tolist()
        self.last_beams = [(tokens[b][:int(seq_lengths[b])], float(seq_lengths[b]), float(scores[b])) for b in range(W)]
        self.last_beam_order = order
        return [texts[b] for b in order]

    def compute_perplexity(self, sentences) -> List[float]:
        """entrypoint.py:155-172: every caption is tokenised again and scored by the language model with labels = inputs:
        perplexity = exp(mean over the L - 1 next-token losses).  ``for sentence in sentences`` over a single caption (a str,
        from the N == 1 branch of the search) walks its CHARACTERS in the reference; kept."""
        rows = [self.tokenizer.encode(s) for s in sentences]
        if not rows:
            return []
        nll = self.engine.lm_score(rows).cpu()
        return [float(torch.exp(nll[i] / (len(r) - 1))) if len(r) > 1 else float("nan") for i, r in enumerate(rows)]


def load_viecap_weights(cfg: dict) -> Dict[str, torch.Tensor]:
    sd = cfg.get("weights")
    if sd is None:
        path = cfg.get("weight_path", DEFAULTS["weight_path"])
        if not os.path.exists(path):
            raise FileNotFoundError("ViECap checkpoint %r not found (no HuggingFace download on this target)" % (path,))
        sd = torch.load(path, map_location="cpu")
    return sd
