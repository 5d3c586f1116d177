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


def _libhdf5():
    import ctypes
    for cand in ("/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so", "libhdf5.so", "libhdf5_serial.so"):
        try:
            return ctypes.CDLL(cand)
        except OSError:
            continue
    return None


def test_writer_round_trip_and_many_heap_collections(tmp_path):
    """h5lite.write_bank -> h5lite.read_datasets: ragged / empty / non-ASCII captions, zero rows, and more strings than one
    global-heap collection holds (16-bit object indices)."""
    g = np.random.default_rng(5)
    M = h5lite.GCOL_MAX_OBJECTS * 2 + 77
    emb = g.standard_normal((M, 8)).astype(np.float32)
    emb[[0, 5]] = 0
    texts = ["caption %d %s" % (i, "x" * (i % 37)) for i in range(M)]
    texts[3], texts[4] = "", "un café à la crème ☕"
    p = str(tmp_path / "big.h5")
    h5lite.write_bank(p, "vg_captions", emb, texts)
    d = h5lite.read_datasets(p)
    assert sorted(d) == ["vg_captions-embeddings", "vg_captions-text"]
    assert np.array_equal(d["vg_captions-embeddings"], emb)
    assert [b.decode("utf-8") for b in d["vg_captions-text"]] == texts
    with pytest.raises(ValueError):
        h5lite.write_bank(p, "x", emb, texts[:-1])


def test_written_bank_is_read_by_the_hdf5_library(tmp_path):
    """the file h5lite.write_bank produces, opened with libhdf5 itself (the library under h5py, i.e. under the reference's
    ``h5py.File(path)[name][:]``, im2txtprojection.py:398-401): both datasets, values and strings identical.  Skipped where
    no libhdf5 exists (the GPU box); the build container has one."""
    import ctypes
    h = _libhdf5()
    if h is None:
        pytest.skip("no libhdf5 on this machine")
    emb, texts = gc.h5_bank_case()
    p = str(tmp_path / "w.h5")
    h5lite.write_bank(p, "coco", emb, texts)
    hid = ctypes.c_int64
    h.H5open()
    for fn, res, args in (("H5Fopen", hid, [ctypes.c_char_p, ctypes.c_uint, hid]), ("H5Dopen2", hid, [hid, ctypes.c_char_p, hid]),
                          ("H5Dread", ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]), ("H5Dget_space", hid, [hid]),
                          ("H5Sget_simple_extent_dims", ctypes.c_int, [hid, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
                          ("H5Tcopy", hid, [hid]), ("H5Tset_size", ctypes.c_int, [hid, ctypes.c_size_t]),
                          ("H5Tset_cset", ctypes.c_int, [hid, ctypes.c_int]), ("H5Dclose", ctypes.c_int, [hid]),
                          ("H5Fclose", ctypes.c_int, [hid]), ("H5Dget_type", hid, [hid]), ("H5Tis_variable_str", ctypes.c_int, [hid]),
                          ("H5Tget_cset", ctypes.c_int, [hid])):
        getattr(h, fn).restype, getattr(h, fn).argtypes = res, args
    f = h.H5Fopen(p.encode(), 0, 0)
    assert f >= 0, "libhdf5 refuses the file"
    d = h.H5Dopen2(f, b"coco-embeddings", 0)
    assert d >= 0
    dims = (ctypes.c_uint64 * 2)()
    assert h.H5Sget_simple_extent_dims(h.H5Dget_space(d), dims, None) == 2 and tuple(dims) == emb.shape
    got = np.empty_like(emb)
    f32 = hid.in_dll(h, "H5T_NATIVE_FLOAT_g").value
    assert h.H5Dread(d, f32, 0, 0, 0, got.ctypes.data_as(ctypes.c_void_p)) >= 0 and np.array_equal(got, emb)
    h.H5Dclose(d)
    d = h.H5Dopen2(f, b"coco-text", 0)
    assert d >= 0
    ft = h.H5Dget_type(d)
    assert h.H5Tis_variable_str(ft) > 0 and h.H5Tget_cset(ft) == 1           # variable-length, UTF-8: h5py.string_dtype('utf-8')
    st = h.H5Tcopy(hid.in_dll(h, "H5T_C_S1_g").value)
    h.H5Tset_size(st, ctypes.c_size_t(-1).value)
    h.H5Tset_cset(st, 1)
    ptrs = (ctypes.c_char_p * len(texts))()
    assert h.H5Dread(d, st, 0, 0, 0, ctypes.cast(ptrs, ctypes.c_void_p)) >= 0
    assert [(x or b"").decode("utf-8") for x in ptrs] == texts
    h.H5Dclose(d)
    assert h.H5Fclose(f) >= 0


def test_bank_file_names_follow_the_reference_rule():
    """im2txtprojection.py:100-170, :234, :289-299 -- the names the released banks carry"""
    from patchioner_amd.bank import memory_bank_filename
    assert memory_bank_filename("coco_captions", None, 591753) == ("coco_captions_text_embeddings-B16-ViT-B.16-591753.h5", "coco_captions")
    assert memory_bank_filename("vg_captions", "ViT-B/16", 500000, linear_talk2dino=True)[0] == \
        "vg_captions_text_embeddings-B16-linear-ViT-B.16-500000.h5"
    assert memory_bank_filename("coco_captions", None, 1000, use_talk2dino=False)[0] == "clip-coco_captions_text_embeddings-ViT-B.32-1000.h5"
    assert memory_bank_filename("my_bank", "ViT-L/14", 10, talk2dino_attn_type="cls")[0] == "my_bank_text_embeddings_cls-B16-ViT-L.14-10.h5"
    assert memory_bank_filename("coco_captions", "ViT-B/32", 7, use_talk2dino=False, use_open_clip=True)[0] == \
        "clip-coco_captions_text_embeddings-open_clip-ViT-B.32-7.h5"
