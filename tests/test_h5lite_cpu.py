"""patchioner_amd/h5lite.py against a REAL HDF5 file (tests/golden/bank_tiny.h5, written by libhdf5 1.10 through
tools/oracle/gen_h5_fixture.py with the calls h5py makes for the reference's bank files, im2txtprojection.py:543-555):
float32 [M, D] embeddings and variable-length UTF-8 caption strings; plus the loud failures."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from patchioner_amd import h5lite

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "bank_tiny.h5")


def test_reads_the_reference_bank_layout():
    emb, texts = gc.h5_bank_case()
    assert h5lite.dataset_names(FIXTURE) == ["coco-embeddings", "coco-text"]
    d = h5lite.read_datasets(FIXTURE)
    assert d["coco-embeddings"].dtype == np.float32 and np.array_equal(d["coco-embeddings"], emb)
    assert [b.decode("utf-8") for b in d["coco-text"]] == texts          # bytes, as h5py returns them (:372 .decode()s)
    only = h5lite.read_datasets(FIXTURE, names=("coco-text",))
    assert list(only) == ["coco-text"]


def test_model_loader_returns_bank_and_texts_and_the_oracle_drops_zero_rows():
    from oracle import patchioner_oracle as O
    from patchioner_amd.model import load_memory_bank
    emb, texts = gc.h5_bank_case()
    bank, t = load_memory_bank(FIXTURE, want_texts=True)
    assert torch.equal(bank, torch.from_numpy(emb)) and [x.decode() for x in t] == texts
    assert torch.equal(load_memory_bank(FIXTURE), bank)
    kept = O.load_bank_rows(bank)
    assert kept.shape[0] == emb.shape[0] - len(gc.H5BANK["zero_rows"])


def test_argmax_text_oracle_indexes_the_unfiltered_texts():
    """im2txtprojection.py:343-345 + :371-375: the arg-max runs over the rows KEPT at load, the text list is not filtered,
    so a hit past a dropped row reads the caption of an earlier row -- restated as is."""
    from oracle import patchioner_oracle as O
    emb, texts = gc.h5_bank_case()
    bank = O.load_bank_rows(torch.from_numpy(emb))
    tb = [t.encode() for t in texts]
    q = torch.from_numpy(emb[[3, 10, 30]]).clone()         # rows 3, 10, 30 of the file = kept rows 3, 9, 28
    caps, sims = O.project(q, bank, return_argmax_text=True, return_n_best_sims=2, text_dataset=tb)
    assert caps == [texts[3], texts[9], texts[28]]
    assert all(abs(s[0] - 1.0) < 1e-5 and s[1] < 0.5 for s in sims)
    assert torch.allclose(q.norm(dim=-1), torch.ones(3), atol=1e-6)      # normalised in place


def test_loud_failures(tmp_path):
    p = tmp_path / "not.h5"
    p.write_bytes(b"this is not an HDF5 file" * 100)
    with pytest.raises(ValueError):
        h5lite.read_datasets(str(p))
    raw = bytearray(open(FIXTURE, "rb").read())
    raw[8] = 2                                              # superblock version 2 (libver='latest' files)
    q = tmp_path / "v2.h5"
    q.write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError):
        h5lite.read_datasets(str(q))
    t = tmp_path / "trunc.h5"
    t.write_bytes(bytes(raw[:4096]).replace(b"\x02", b"\x00", 0))
    with pytest.raises((ValueError, NotImplementedError)):
        h5lite.read_datasets(str(t))
