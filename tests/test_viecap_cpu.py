"""ViECap head, CPU side: the oracle's restatement (oracle/patchioner_oracle.py: ViECapOracle) against the REFERENCE's own
pieces run by tools/oracle/gen_golden_viecap.py (tests/golden/viecap.npz), and the host logic of the mirror
(patchioner_amd/viecap.py: byte-level BPE, prompt text, top-k / threshold, sentence cut)."""
import json

import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import patchioner_oracle as O
from patchioner_amd import viecap as V

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def case():
    w, (vocab, merges), ents, emb, x = gc.viecap_case()
    return w, V.ByteLevelBPE(vocab, merges), ents, emb, x


def test_oracle_matches_the_reference_pieces(golden, case):
    g = golden("viecap")
    meta = json.loads(bytes(g["meta_json"]).decode())
    w, tok, ents, emb, x = case
    c = gc.VIECAP
    orc = O.ViECapOracle(w, tok, ents, emb, temperature=c["temperature"], top_k=c["top_k"], threshold=c["threshold"],
                         using_hard_prompt=True, soft_prompt_first=True)
    xin = x.clone()
    caps = orc.forward(xin)
    np.testing.assert_allclose(xin.numpy(), (x / x.norm(dim=-1, keepdim=True)).numpy(), rtol=1e-6, atol=1e-7)   # in place
    np.testing.assert_allclose(orc.last["cont"].numpy(), g["cont"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(orc.last["entity_probs"].numpy(), g["entity_probs"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(orc.last["prompt_tokens"].numpy(), g["prompt_tokens"])
    assert [orc.detect(orc.last["entity_probs"][i]) for i in range(x.shape[0])] == meta["detected"]
    ids, ref = orc.last["ids"].numpy(), g["decoded_ids"]          # the reference's ids end at the first full stop (-1 after it)
    eos = tok.encode(".")[-1]
    for r in range(ids.shape[0]):
        n = int((ref[r] >= 0).sum())
        assert np.array_equal(ids[r, :n], ref[r, :n]), "row %d, min top-2 margin %.2e" % (r, float(orc.last["margins"][r].min()))
        assert n == 64 or (ids[r, n - 1] == eos and eos not in ids[r, :n - 1])
    assert (ref < 0).any(), "the fixture should exercise the cut at the first full stop"
    assert caps == meta["sentences"]
    one = orc.forward(x[1:2].clone())
    assert isinstance(one, str) and one == meta["single_sentence"]
    # VieCap.compute_perplexity of the reference on its own sentences (entrypoint.py:155-172)
    np.testing.assert_allclose(orc.compute_perplexity(meta["perplexity_sentences"]), g["perplexity"], rtol=2e-4)
    assert np.isnan(orc.compute_perplexity(["c"])[0]) if len(tok.encode("c")) == 1 else True      # one token: mean of nothing


def test_oracle_beam_search_matches_the_reference(golden, case):
    """tests/golden/viecap_beam.npz: the reference's beam_search (search.py:193-285) on three prompts, with its own end-of-sentence
    strings and with two the seeded model does emit (beams stop at different steps): every beam's ids and the returned order."""
    g = golden("viecap_beam")
    meta = json.loads(bytes(g["meta_json"]).decode())
    w, tok, ents, emb, x = case
    c = gc.VIECAP
    orc = O.ViECapOracle(w, tok, ents, emb, temperature=c["temperature"], top_k=c["top_k"], threshold=c["threshold"],
                         using_hard_prompt=True, soft_prompt_first=True)
    prompts = orc.prompt_embeddings(x[:3].clone())
    stops = 0
    for call in meta["calls"]:
        eos = (".", " .") if call["label"] == "default" else tuple(meta["eos_strings"])
        i = call["image"]
        got = orc.beam_search(prompts[i:i + 1], beam_width=5, end_of_sentences=eos)
        ref = g["%s_%d_ids" % (call["label"], i)]
        worst = min(orc.last_beam["margins"])
        for b in range(5):
            n = int((ref[b] >= 0).sum())
            assert orc.last_beam["ids"][b] == ref[b, :n].tolist(), (call["label"], i, b, "smallest selection margin %.2e" % worst)
        assert got == call["sentences"], (call["label"], i)
        stops += sum(l < 64 for l in call["lengths"])
    assert stops >= 5, "the fixture should exercise beams that stop early"


def test_mirror_host_logic_matches_the_oracle(case):
    w, tok, ents, emb, x = case
    assert V.compose_discrete_prompt_text([]) == O.ViECapOracle.prompt_text([]) == "There are something in image."
    for e in (["dog"], ["traffic light", "bear"], ["a", "b", "c"]):
        assert V.compose_discrete_prompt_text(e) == O.ViECapOracle.prompt_text(e)
    s = "There are person, traffic light in image."
    ids = tok.encode(s)
    assert tok.decode(ids) == s and all(0 <= i < len(tok) for i in ids)
    assert tok.decode(tok.encode("café 山 .")) == "café 山 ."
    probs = torch.tensor([0.05, 0.45, 0.1, 0.4])
    assert V.top_k_entities(["a", "b", "c", "d"], probs, 3, 0.4) == ["b", "d"]
    assert V.top_k_entities(["a", "b", "c", "d"], probs, 3, 0.5) == []
    assert V.top_k_entities(["a", "b", "c", "d"], probs, 1, 0.0) == ["b"]


def test_config_errors_are_loud(case):
    w, tok, ents, emb, _ = case

    class FakeEngine:
        device = torch.device("cpu")

        def viecap_set_entities(self, e):
            self.n = e.shape[0]

    base = dict(clip_hidden_size=768, entities_text=ents, texts_embeddings=emb, tokenizer=tok, using_greedy_search=True)
    V.VieCapHead(dict(base), FakeEngine(), "ViT-B/16")
    V.VieCapHead(dict(base, using_greedy_search=False), FakeEngine(), "ViT-B/16")                # beam search (width 5)
    with pytest.raises(ValueError):
        V.VieCapHead(dict(base, using_greedy_search=False, beam_width=9), FakeEngine(), "ViT-B/16")
    with pytest.raises(NotImplementedError):
        V.VieCapHead(dict(base, language_model="facebook/opt-1.3b"), FakeEngine(), "ViT-B/16")
    with pytest.raises(FileNotFoundError):
        V.VieCapHead({k: v for k, v in base.items() if k != "tokenizer"}, FakeEngine(), "ViT-B/16")
    with pytest.raises(ValueError):
        V.VieCapHead(dict(base, texts_embeddings=emb[:, :512]), FakeEngine(), "ViT-B/16")


def test_entity_vocabularies_default_vinvl_vgoi_and_the_others(tmp_path):
    """get_viecap_texts_embeddings (P/src/viecap/entrypoint.py:179-223): every vocabulary the reference reads, starting with its
    DEFAULT ``vinvl_vgoi_entities`` (a {name: index} json + vgoi_embeddings_<suffix>.pickle); names lower-cased, stripped,
    sorted, single words only with disable_all_entities; a missing embeddings pickle or an unknown name fails loudly."""
    import csv
    import json
    import pickle
    from argparse import Namespace
    import pytest
    import torch
    from patchioner_amd.viecap import _entity_files
    vd = tmp_path / "annotations" / "vocabulary"
    vd.mkdir(parents=True)
    json.dump({"Traffic Light": 0, " zebra": 1, "apple": 2}, open(vd / "vgcocooiobjects_v1_class2ind.json", "w"))
    json.dump(["Dog", "hot dog", "cat "], open(vd / "coco_categories.json", "w"))
    json.dump({"object_count": {"Tree": 5, "street sign": 2}}, open(vd / "VG-SGG-dicts-vgoi6-clipped.json", "w"))
    pickle.dump({"objects": {"joint": {"Lamp", "tea pot"}}}, open(vd / "all_objects_attributes_relationships.pickle", "wb"))
    with open(vd / "oidv7-class-descriptions-boxable.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["LabelName", "DisplayName"])
        w.writerow(["/m/1", "Bat (Animal)"])
        w.writerow(["/m/2", "Ball"])
    for stem, k in (("vgoi_embeddings", 3), ("coco_embeddings", 3), ("vg_embeddings", 2), ("visual_genome_embedding", 2), ("open_image_embeddings", 2)):
        pickle.dump(torch.arange(k * 4, dtype=torch.float32).view(k, 4), open(vd / ("%s_ViT-B16.pickle" % stem), "wb"))
    pickle.dump(torch.ones(3, 4), open(vd / "vgoi_embeddings_ViT-B16_with_ensemble.pickle", "wb"))

    def args(name, **kw):
        return Namespace(files_path=str(tmp_path), name_of_entities_text=name, disable_all_entities=False, prompt_ensemble=False, **kw)
    ents, emb = _entity_files(args("vinvl_vgoi_entities"), "ViT-B/16")
    assert ents == ["apple", "traffic light", "zebra"] and emb.shape == (3, 4)
    a = args("vinvl_vgoi_entities")
    a.disable_all_entities, a.prompt_ensemble = True, True
    ents, emb = _entity_files(a, "ViT-B/16")
    assert ents == ["apple", "zebra"] and bool((emb == 1).all())
    assert _entity_files(args("coco_entities"), "ViT-B/16")[0] == ["cat", "dog", "hot dog"]
    assert _entity_files(args("vinvl_vg_entities"), "ViT-B/16")[0] == ["street sign", "tree"]
    assert _entity_files(args("visual_genome_entities"), "ViT-B/16")[0] == ["lamp", "tea pot"]
    assert _entity_files(args("open_image_entities"), "ViT-B/16")[0] == ["ball", "bat"]
    with pytest.raises(FileNotFoundError):
        _entity_files(args("coco_entities"), "RN50x4")          # no embeddings pickle for that CLIP: needs the text tower
    with pytest.raises(ValueError):
        _entity_files(args("imagenet_entities"), "ViT-B/16")


def test_sharded_helpers_take_the_id_width_from_the_head():
    from types import SimpleNamespace
    import pytest
    import torch
    from patchioner_amd import dist as pdist
    assert pdist._id_columns(SimpleNamespace(viecap=None, calculate_argmax_text=False)) == 30
    assert pdist._id_columns(SimpleNamespace(viecap=object(), calculate_argmax_text=False)) == 64
    with pytest.raises(ValueError):
        pdist._id_columns(SimpleNamespace(viecap=None, calculate_argmax_text=True))
    with pytest.raises(RuntimeError):
        pdist._last_ids(SimpleNamespace(last_ids=torch.zeros(2, 30)), 64)
