"""A dependency-free reader AND writer for the memory-bank files of the reference (HDF5 as h5py writes it by default).

``Im2TxtProjector._build_support_memory`` (P/src/decap/im2txtprojection/im2txtprojection.py:543-555) creates, in the
root group, ``<name>-embeddings`` = float32 [M, D] and ``<name>-text`` = [M] variable-length UTF-8 strings
(``h5py.string_dtype``), both with the library's default (contiguous) layout, and ``_load_support_memory`` (:387-407)
reads them back whole.  h5py is not installed on the target image, so this module parses exactly that subset of the HDF5
file format (version-0/1 superblock, old-style groups = v1 B-tree + local heap + symbol nodes, version-1 object headers
with continuation blocks, dataspace / datatype / data-layout messages, contiguous or compact storage, global heap
collections for the variable-length strings).  Anything else (chunked / filtered datasets, new-style groups, superblock
2+) raises ``NotImplementedError`` with the reason, never a silent wrong read.

``write_bank`` is the other direction (the tail of ``_build_support_memory``, :543-555): the same two datasets in the same
layout, laid out as the library itself lays out a fresh file (version-0 superblock, one symbol node, contiguous data, global
heap collections for the strings), so that h5py / libhdf5 -- and therefore the reference -- open what it writes.

Format reference: the public "HDF5 File Format Specification Version 2.0/3.0" (III.A superblock, III.A.1 B-trees, III.C
symbol-table nodes, III.D local heaps, III.E global heaps, IV.A object headers and messages).
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple, Union

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class _File:
    def __init__(self, path: str):
        self.f = open(path, "rb")
        self.base = 0
        self.O = self.L = 8

    def read(self, addr: int, n: int) -> bytes:
        self.f.seek(self.base + addr)
        b = self.f.read(n)
        if len(b) != n:
            raise ValueError("truncated HDF5 file: wanted %d bytes at %d" % (n, addr))
        return b

    def uint(self, b: bytes, off: int, size: int) -> int:
        return int.from_bytes(b[off:off + size], "little")

    def close(self):
        self.f.close()


def _superblock(fh: _File) -> Tuple[int, int]:
    """-> (B-tree address, local heap address) of the root group."""
    pos = 0
    while True:
        fh.f.seek(pos)
        if fh.f.read(8) == SIGNATURE:
            break
        pos = 512 if pos == 0 else pos * 2
        if pos > (1 << 26):
            raise ValueError("not an HDF5 file (no superblock signature)")
    fh.f.seek(pos + 8)
    b = fh.f.read(120)
    version = b[0]
    if version > 1:
        raise NotImplementedError("HDF5 superblock version %d (new-style file, e.g. libver='latest'): this reader handles the "
                                  "h5py default (version 0); convert the bank to .npy" % version)
    fh.O, fh.L = b[5], b[6]
    if fh.O != 8 or fh.L != 8:
        raise NotImplementedError("HDF5 offsets / lengths of %d / %d bytes (expected 8 / 8)" % (fh.O, fh.L))
    off = 16 + (4 if version == 1 else 0)       # version 1 adds indexed-storage K + reserved
    base = fh.uint(b, off, 8)
    fh.base = base
    root = off + 4 * 8                            # base, free-space, end-of-file, driver-info addresses
    # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
    ohdr = fh.uint(b, root + 8, 8)
    cache = fh.uint(b, root + 16, 4)
    if cache == 1:
        return fh.uint(b, root + 24, 8), fh.uint(b, root + 32, 8)
    msgs = _object_header(fh, ohdr)
    for t, data in msgs:
        if t == 0x0011:
            return fh.uint(data, 0, 8), fh.uint(data, 8, 8)
    raise NotImplementedError("root group without a symbol-table message (new-style group)")


def _object_header(fh: _File, addr: int) -> List[Tuple[int, bytes]]:
    h = fh.read(addr, 16)
    if h[:4] == b"OHDR":
        raise NotImplementedError("version-2 object header (file written with libver='latest')")
    if h[0] != 1:
        raise ValueError("unsupported object header version %d at %d" % (h[0], addr))
    nmsg = fh.uint(h, 2, 2)
    size = fh.uint(h, 8, 4)
    blocks = [(addr + 16, size)]
    out: List[Tuple[int, bytes]] = []
    while blocks and len(out) < nmsg:
        a, n = blocks.pop(0)
        buf = fh.read(a, n)
        p = 0
        while p + 8 <= n and len(out) < nmsg:
            t, sz = fh.uint(buf, p, 2), fh.uint(buf, p + 2, 2)
            data = buf[p + 8:p + 8 + sz]
            p += 8 + sz
            if t == 0x0010:                       # continuation: (offset, length) of another block of messages
                blocks.append((fh.uint(data, 0, 8), fh.uint(data, 8, 8)))
            out.append((t, data))
    return out


def _local_heap(fh: _File, addr: int) -> bytes:
    h = fh.read(addr, 32)
    if h[:4] != b"HEAP":
        raise ValueError("bad local heap signature at %d" % addr)
    size, data_addr = fh.uint(h, 8, 8), fh.uint(h, 24, 8)
    return fh.read(data_addr, size)


def _group_entries(fh: _File, btree: int, heap: bytes) -> Dict[str, int]:
    """name -> object header address, walking the group's v1 B-tree down to its symbol nodes."""
    out: Dict[str, int] = {}

    def node(addr: int):
        h = fh.read(addr, 24)
        if h[:4] == b"SNOD":
            n = fh.uint(h, 6, 2)
            body = fh.read(addr + 8, n * 40)
            for i in range(n):
                e = body[i * 40:(i + 1) * 40]
                name_off, ohdr = fh.uint(e, 0, 8), fh.uint(e, 8, 8)
                end = heap.index(b"\0", name_off)
                out[heap[name_off:end].decode("utf-8")] = ohdr
            return
        if h[:4] != b"TREE" or h[4] != 0:
            raise ValueError("bad group B-tree node at %d" % addr)
        used = fh.uint(h, 6, 2)
        body = fh.read(addr + 24, (2 * used + 1) * 8)
        for i in range(used):
            node(fh.uint(body, (2 * i + 1) * 8, 8))

    node(btree)
    return out


def _datatype(fh: _File, d: bytes):
    """-> ('float' | 'int' | 'uint' | 'str' | 'vlen_str', element size in the file)"""
    cls, ver = d[0] & 0x0F, d[0] >> 4
    bits = d[1] | (d[2] << 8) | (d[3] << 16)
    size = fh.uint(d, 4, 4)
    if cls in (0, 1) and (bits & 1):
        raise NotImplementedError("big-endian HDF5 numbers")
    if cls == 1:
        return "float", size
    if cls == 0:
        return ("int" if bits & 0x08 else "uint"), size
    if cls == 3:
        return "str", size
    if cls == 9:
        if (bits & 0x0F) != 1:
            raise NotImplementedError("variable-length sequences (only variable-length strings are read)")
        return "vlen_str", size
    raise NotImplementedError("HDF5 datatype class %d (version %d)" % (cls, ver))


def _dataset(fh: _File, ohdr: int):
    shape = kind = esize = layout = None
    for t, data in _object_header(fh, ohdr):
        if t == 0x0001:
            ver, rank = data[0], data[1]
            start = 8 if ver == 1 else 4
            shape = tuple(fh.uint(data, start + 8 * i, 8) for i in range(rank))
        elif t == 0x0003:
            kind, esize = _datatype(fh, data)
        elif t == 0x0008:
            ver = data[0]
            if ver != 3:
                raise NotImplementedError("data layout message version %d" % ver)
            cls = data[1]
            if cls == 1:
                layout = ("contiguous", fh.uint(data, 2, 8), fh.uint(data, 10, 8))
            elif cls == 0:
                n = fh.uint(data, 2, 2)
                layout = ("compact", data[4:4 + n])
            else:
                raise NotImplementedError("chunked HDF5 dataset (the reference writes contiguous ones); convert the bank to .npy")
        elif t == 0x000B:
            raise NotImplementedError("filtered (compressed) HDF5 dataset")
    if shape is None or kind is None or layout is None:
        return None                                # a group or a committed datatype, not a dataset
    count = int(np.prod(shape)) if shape else 1
    if layout[0] == "contiguous":
        addr = layout[1]
        if addr != UNDEF and kind in ("float", "int", "uint"):
            # straight from the file into the array (the COCO bank is 1.8 GB: no intermediate bytes object)
            dt = {"float": "<f%d", "int": "<i%d", "uint": "<u%d"}[kind] % esize
            fh.f.seek(fh.base + addr)
            arr = np.fromfile(fh.f, dtype=dt, count=count)
            if arr.size != count:
                raise ValueError("truncated HDF5 file: dataset of %d elements, %d read" % (count, arr.size))
            return arr.reshape(shape)
        raw = b"\0" * (count * esize) if addr == UNDEF else fh.read(addr, count * esize)      # never written: fill value 0
    else:
        raw = layout[1]
    if kind == "float":
        return np.frombuffer(raw, dtype={2: "<f2", 4: "<f4", 8: "<f8"}[esize], count=count).reshape(shape).copy()
    if kind in ("int", "uint"):
        return np.frombuffer(raw, dtype="<%s%d" % ("i" if kind == "int" else "u", esize), count=count).reshape(shape).copy()
    if kind == "str":
        return [raw[i * esize:(i + 1) * esize].split(b"\0")[0] for i in range(count)]
    # variable-length strings: (length u32, heap collection address, object index u32) per element
    heaps: Dict[int, Dict[int, bytes]] = {}
    out: List[bytes] = []
    for i in range(count):
        n, coll, idx = struct.unpack_from("<IQI", raw, i * 16)
        if coll == 0 or coll == UNDEF or n == 0:
            out.append(b"")
            continue
        if coll not in heaps:
            heaps[coll] = _global_heap(fh, coll)
        out.append(heaps[coll][idx][:n])
    return out


def _global_heap(fh: _File, addr: int) -> Dict[int, bytes]:
    h = fh.read(addr, 16)
    if h[:4] != b"GCOL":
        raise ValueError("bad global heap signature at %d" % addr)
    size = fh.uint(h, 8, 8)
    buf = fh.read(addr, size)
    objs: Dict[int, bytes] = {}
    p = 16
    while p + 16 <= size:
        idx, osize = fh.uint(buf, p, 2), fh.uint(buf, p + 8, 8)
        if idx == 0:
            break                                 # free space closes the collection
        objs[idx] = buf[p + 16:p + 16 + osize]
        p += 16 + ((osize + 7) // 8) * 8
    return objs


def read_datasets(path: str, names=None) -> Dict[str, Union[np.ndarray, List[bytes]]]:
    """Every dataset of the root group (or only ``names``): numeric ones as ndarrays, strings as lists of bytes
    (what ``h5py`` returns for ``dataset[:]``: callers ``.decode()`` them, im2txtprojection.py:372)."""
    fh = _File(path)
    try:
        btree, heap_addr = _superblock(fh)
        entries = _group_entries(fh, btree, _local_heap(fh, heap_addr))
        out = {}
        for name, ohdr in entries.items():
            if names is not None and name not in names:
                continue
            d = _dataset(fh, ohdr)
            if d is not None:
                out[name] = d
        return out
    finally:
        fh.close()


def dataset_names(path: str) -> List[str]:
    fh = _File(path)
    try:
        btree, heap_addr = _superblock(fh)
        return sorted(_group_entries(fh, btree, _local_heap(fh, heap_addr)))
    finally:
        fh.close()


# ------------------------------------------------------------------------------------------------------------------
# writer


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype: int, data: bytes) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), 0) + data


def _ohdr_v1(msgs: List[bytes]) -> bytes:
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


_F32LE = bytes.fromhex("11201f00" "04000000" "0000" "2000" "17" "08" "00" "17" "7f000000")
# variable-length UTF-8 string, null-terminated, over a 1-byte base element: the bytes libhdf5 emits for
# H5Tcopy(H5T_C_S1) + H5Tset_size(H5T_VARIABLE) + H5Tset_cset(UTF8), which is what h5py.string_dtype('utf-8') is
_VLEN_UTF8 = bytes.fromhex("19010100" "10000000" "10000000" "01000000" "00000800")
GCOL_MAX_OBJECTS = 8192          # object indices are 16 bits; collections stay around a megabyte
GCOL_MIN_SIZE = 4096


def _dataset_header(shape, dtype_msg: bytes, fill: bytes, addr: int, nbytes: int) -> bytes:
    rank = len(shape)
    space = struct.pack("<BBB5x", 1, rank, 1) + b"".join(struct.pack("<Q", int(d)) for d in shape) * 2     # dims, max dims
    layout = struct.pack("<BBQQ", 3, 1, addr, nbytes)
    return _ohdr_v1([_msg(0x0001, space), _msg(0x0003, dtype_msg), _msg(0x0005, fill), _msg(0x0008, layout)])


def write_bank(path: str, name: str, embeddings: np.ndarray, texts) -> None:
    """``<name>-embeddings`` float32 [M, D] and ``<name>-text`` [M] variable-length UTF-8 strings in the root group of a new
    HDF5 file -- the file ``Im2TxtProjector._build_support_memory`` leaves behind (im2txtprojection.py:543-555) and
    ``_load_support_memory`` (:387-407) / ``read_datasets`` read back."""
    emb = np.ascontiguousarray(embeddings, dtype="<f4")
    if emb.ndim != 2:
        raise ValueError("embeddings must be [M, D]")
    enc = [t if isinstance(t, bytes) else str(t).encode("utf-8") for t in texts]
    if len(enc) != emb.shape[0]:
        raise ValueError("len(texts) = %d != len(embeddings) = %d" % (len(enc), emb.shape[0]))   # the reference's assert (:537)
    names = sorted([("%s-embeddings" % name).encode(), ("%s-text" % name).encode()])
    e_name, t_name = ("%s-embeddings" % name).encode(), ("%s-text" % name).encode()

    # local heap data: the empty name at 0, then the link names, then one free block
    heap = bytearray(8)
    name_off = {}
    for n in names:
        name_off[n] = len(heap)
        heap += _pad8(n + b"\0")
    free_at = len(heap)
    heap += struct.pack("<QQ", 1, 32) + b"\0" * 16                  # next = 1 (end of list), size of this block

    K_LEAF, K_NODE = 4, 16
    btree_size = 24 + (2 * K_NODE + 1) * 8 + 2 * K_NODE * 8
    snod_size = 8 + 2 * K_LEAF * 40
    hdr_size = 16 + 4 * 8 + 40 + 24 + 24 + 24                     # v1 prefix, dataspace (rank<=2 padded below), type, fill, layout

    a_root = 96
    root_hdr_len = 16 + 8 + 16
    a_btree = a_root + root_hdr_len
    a_heap = a_btree + btree_size
    a_heapdata = a_heap + 32
    a_snod = a_heapdata + len(heap)
    a_e_hdr = a_snod + snod_size
    e_hdr = _dataset_header(emb.shape, _F32LE, bytes.fromhex("0202020100000000"), 0, 0)
    a_t_hdr = a_e_hdr + len(e_hdr)
    t_hdr = _dataset_header((len(enc),), _VLEN_UTF8, bytes.fromhex("0202000100000000"), 0, 0)
    a_e_data = (a_t_hdr + len(t_hdr) + 7) // 8 * 8
    a_t_data = a_e_data + emb.nbytes
    a_gcol = a_t_data + 16 * len(enc)
    del hdr_size

    # global heap collections for the strings + the 16-byte references that point into them
    refs = bytearray(16 * len(enc))
    colls: List[bytes] = []
    addr = a_gcol
    i = 0
    while i < len(enc):
        body = bytearray()
        idx = 1
        while i < len(enc) and idx <= GCOL_MAX_OBJECTS:
            b = enc[i]
            if b:                                                   # an empty string is the null reference, as libhdf5 writes it
                struct.pack_into("<IQI", refs, 16 * i, len(b), addr, idx)
                body += struct.pack("<HH4xQ", idx, 1, len(b)) + _pad8(b)
                idx += 1
            i += 1
        size = 16 + len(body)
        total = max(GCOL_MIN_SIZE, size + 16)
        body += struct.pack("<HH4xQ", 0, 0, total - size) + b"\0" * (total - size - 16)      # object 0: the free space
        colls.append(b"GCOL" + struct.pack("<B3xQ", 1, total) + bytes(body))
        addr += total
    eof = addr

    e_hdr = _dataset_header(emb.shape, _F32LE, bytes.fromhex("0202020100000000"), a_e_data, emb.nbytes)
    t_hdr = _dataset_header((len(enc),), _VLEN_UTF8, bytes.fromhex("0202000100000000"), a_t_data, 16 * len(enc))

    sb = SIGNATURE + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, K_LEAF, K_NODE, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQI4xQQ", 0, a_root, 1, a_btree, a_heap)    # root entry: cached B-tree / heap addresses
    assert len(sb) == 96
    root = _ohdr_v1([_msg(0x0011, struct.pack("<QQ", a_btree, a_heap))])
    assert len(root) == root_hdr_len
    btree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, a_snod, name_off[names[-1]])
    btree += b"\0" * (btree_size - len(btree))
    lheap = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_at, a_heapdata)
    hdr_of = {e_name: a_e_hdr, t_name: a_t_hdr}
    snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for n in names:
        snod += struct.pack("<QQI4x16x", name_off[n], hdr_of[n], 0)
    snod += b"\0" * (snod_size - len(snod))

    with open(path, "wb") as f:
        f.write(sb + root + btree + lheap + bytes(heap) + snod + e_hdr + t_hdr)
        f.write(b"\0" * (a_e_data - f.tell()))
        emb.tofile(f)
        f.write(bytes(refs))
        for c in colls:
            f.write(c)
        assert f.tell() == eof
