"""Writes tests/golden/bank_tiny.h5 with the HDF5 LIBRARY itself (libhdf5 1.10 through ctypes -- the calls h5py makes for
``create_dataset(name, shape, dtype='float32')`` / ``dtype=h5py.string_dtype('utf-8')`` and ``dataset[...] = data``,
P/src/decap/im2txtprojection/im2txtprojection.py:543-555), so that the dependency-free reader patchioner_amd/h5lite.py
is tested against a real HDF5 file, not against a writer of our own.  h5py is not installed in the build container; a
conda tree there ships libhdf5.so (no source of the reference is involved).

    python tools/oracle/gen_h5_fixture.py [/path/to/libhdf5.so]
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else "/opt/conda/lib/libhdf5.so.103"
    h = ctypes.CDLL(lib)
    hid = ctypes.c_int64
    h.H5open()
    for fn, res, args in (("H5Fcreate", hid, [ctypes.c_char_p, ctypes.c_uint, hid, hid]),
                          ("H5Screate_simple", hid, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
                          ("H5Dcreate2", hid, [hid, ctypes.c_char_p, hid, hid, hid, hid, hid]),
                          ("H5Dwrite", ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]),
                          ("H5Tcopy", hid, [hid]), ("H5Tset_size", ctypes.c_int, [hid, ctypes.c_size_t]),
                          ("H5Tset_cset", ctypes.c_int, [hid, ctypes.c_int]), ("H5Dclose", ctypes.c_int, [hid]),
                          ("H5Sclose", ctypes.c_int, [hid]), ("H5Tclose", ctypes.c_int, [hid]), ("H5Fclose", ctypes.c_int, [hid])):
        getattr(h, fn).restype, getattr(h, fn).argtypes = res, args
    f32le = hid.in_dll(h, "H5T_IEEE_F32LE_g").value
    c_s1 = hid.in_dll(h, "H5T_C_S1_g").value
    emb, texts = gc.h5_bank_case()
    name = gc.H5BANK["name"]
    out = os.path.join(ROOT, "tests", "golden", "bank_tiny.h5")
    f = h.H5Fcreate(out.encode(), 2, 0, 0)                      # H5F_ACC_TRUNC, default property lists
    assert f >= 0
    dims = (ctypes.c_uint64 * 2)(*emb.shape)
    sp = h.H5Screate_simple(2, dims, None)
    d = h.H5Dcreate2(f, ("%s-embeddings" % name).encode(), f32le, sp, 0, 0, 0)
    assert d >= 0 and h.H5Dwrite(d, f32le, 0, 0, 0, emb.ctypes.data_as(ctypes.c_void_p)) >= 0
    h.H5Dclose(d); h.H5Sclose(sp)
    st = h.H5Tcopy(c_s1)
    assert h.H5Tset_size(st, ctypes.c_size_t(-1).value) >= 0 and h.H5Tset_cset(st, 1) >= 0     # H5T_VARIABLE, H5T_CSET_UTF8
    dims1 = (ctypes.c_uint64 * 1)(len(texts))
    sp = h.H5Screate_simple(1, dims1, None)
    d = h.H5Dcreate2(f, ("%s-text" % name).encode(), st, sp, 0, 0, 0)
    enc = [t.encode("utf-8") for t in texts]
    ptrs = (ctypes.c_char_p * len(enc))(*enc)
    assert d >= 0 and h.H5Dwrite(d, st, 0, 0, 0, ctypes.cast(ptrs, ctypes.c_void_p)) >= 0
    h.H5Dclose(d); h.H5Sclose(sp); h.H5Tclose(st)
    assert h.H5Fclose(f) >= 0
    print("wrote %s (%d bytes)" % (out, os.path.getsize(out)))


if __name__ == "__main__":
    main()
