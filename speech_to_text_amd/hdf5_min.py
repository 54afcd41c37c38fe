"""A minimal, dependency-free HDF5 writer/reader for the reference's feature files.

The reference stores one file per utterance (/root/reference/src/preprocess.py:138-144):

    with h5py.File(f'mfcc/{speaker_id}_{chapter_id}_{utterance_id}.hdf5', 'w') as file:
        dataset = file.create_dataset('mfccs', data=mfcc_frames)     # float32 (F, n_mfcc), contiguous
        dataset.attrs['speaker_id' | 'chapter_id' | 'utterance_id' | 'sample_rate'] = int
        dataset.attrs['transcript'] = str                            # variable-length UTF-8

and reads them back with ``file['mfccs'][:]`` and ``.attrs[...]`` (/root/reference/src/trainer.py:110-114).
h5py is used when it is importable; this module writes/reads the same schema in plain Python so
the engine still produces real HDF5 files on machines without h5py/libhdf5.  Files follow the
HDF5 File Format Specification version 1 structures that h5py's default (``libver='earliest'``)
produces: version-0 superblock, old-style group (B-tree + local heap + symbol-table node),
version-1 object headers, contiguous layout, version-1 attribute messages, variable-length
strings in a global heap collection.
"""
from __future__ import annotations

import struct
from typing import Dict, Tuple, Union

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"

AttrValue = Union[int, str]


def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


# ---- datatype / dataspace encodings ------------------------------------------------------------
def _dt_f32() -> bytes:
    # class 1 (floating point), version 1; little-endian, implied-msb mantissa, sign at bit 31
    return struct.pack("<B3BI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _dt_i64() -> bytes:
    # class 0 (fixed point), version 1; little-endian, signed
    return struct.pack("<B3BI", 0x10, 0x08, 0, 0, 8) + struct.pack("<HH", 0, 64)


def _dt_vlen_utf8() -> bytes:
    # class 9 (variable length), version 1; type = string, null-terminated, UTF-8.  libhdf5 gives a
    # variable-length C string the base type unsigned 8-bit integer (H5Tset_size(.., H5T_VARIABLE)).
    base = struct.pack("<B3BI", 0x10, 0x00, 0, 0, 1) + struct.pack("<HH", 0, 8)
    return struct.pack("<B3BI", 0x19, 0x01, 0x01, 0, 16) + base


def _ds_simple(shape: Tuple[int, ...]) -> bytes:
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)


def _message(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _attribute(name: str, dt: bytes, ds: bytes, data: bytes) -> bytes:
    nm = name.encode("utf-8") + b"\x00"
    body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data
    return _message(0x000C, body)


def _object_header(messages: bytes, n_messages: int) -> bytes:
    return struct.pack("<BxHII4x", 1, n_messages, 1, len(messages)) + messages


# ---- writer -----------------------------------------------------------------------------------
def write_features(path, data: np.ndarray, attrs: Dict[str, AttrValue], name: str = "mfccs") -> None:
    """Write one dataset ``name`` (float32, C-contiguous, any rank >= 1) with scalar attributes
    (Python int -> int64, str -> variable-length UTF-8) to a new HDF5 file at ``path``."""
    data = np.ascontiguousarray(data, dtype="<f4")
    nm = name.encode("utf-8") + b"\x00"

    # fixed-size pieces first, to know the addresses
    SB, ROOT_OH, BTREE, HEAP_HDR, SNOD = 96, 40, 544, 32, 328
    heap_data = _pad8(b"\x00" * 8 + nm)
    heap_data += struct.pack("<QQ", 1, 16)                       # one free block closing the segment
    a_root = SB
    a_btree = a_root + ROOT_OH
    a_heap = a_btree + BTREE
    a_heap_data = a_heap + HEAP_HDR
    a_snod = a_heap_data + len(heap_data)
    a_dset = a_snod + SNOD

    # global heap for variable-length strings
    strings = [(k, v.encode("utf-8")) for k, v in attrs.items() if isinstance(v, str)]
    gheap_objs = b""
    str_index = {}
    for i, (k, raw) in enumerate(strings, start=1):
        str_index[k] = (i, len(raw))
        gheap_objs += struct.pack("<HH4xQ", i, 1, len(raw)) + _pad8(raw)

    def dataset_header(a_gheap: int, a_data: int) -> bytes:
        msgs = [
            _message(0x0001, _ds_simple(data.shape)),
            _message(0x0003, _dt_f32(), flags=1),                # constant message
            _message(0x0005, struct.pack("<BBBBI", 2, 2, 2, 1, 0)),   # fill value: default, alloc late
            _message(0x0008, struct.pack("<BBQQ", 3, 1, a_data if data.nbytes else UNDEF, data.nbytes)),
        ]
        for k, v in attrs.items():
            if isinstance(v, str):
                idx, ln = str_index[k]
                msgs.append(_attribute(k, _dt_vlen_utf8(), _ds_simple(()), struct.pack("<IQI", ln, a_gheap, idx)))
            elif isinstance(v, (int, np.integer)) and not isinstance(v, bool):
                msgs.append(_attribute(k, _dt_i64(), _ds_simple(()), struct.pack("<q", int(v))))
            else:
                raise TypeError(f"attribute {k!r}: only int and str are supported, got {type(v).__name__}")
        return _object_header(b"".join(msgs), len(msgs))

    dset_len = len(dataset_header(0, 0))
    a_gheap = a_dset + dset_len
    gheap = b""
    if strings:
        size = max(4096, (16 + len(gheap_objs) + 16 + 7) // 8 * 8)
        free = size - 16 - len(gheap_objs)
        gheap = b"GCOL" + struct.pack("<B3xQ", 1, size) + gheap_objs
        gheap += struct.pack("<HH4xQ", 0, 0, free) + b"\x00" * (free - 16)
    a_data = a_gheap + len(gheap)
    dset = dataset_header(a_gheap, a_data)
    eof = a_data + data.nbytes

    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQI4xQQ", 0, a_root, 1, a_btree, a_heap)     # root symbol-table entry
    assert len(sb) == SB

    root = _object_header(_message(0x0011, struct.pack("<QQ", a_btree, a_heap)), 1)
    assert len(root) == ROOT_OH

    off_name = 8
    btree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF)
    btree += struct.pack("<QQQ", 0, a_snod, off_name)
    btree += b"\x00" * (BTREE - len(btree))

    heap = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), len(heap_data) - 16, a_heap_data)
    assert len(heap) == HEAP_HDR

    snod = b"SNOD" + struct.pack("<BxH", 1, 1)
    snod += struct.pack("<QQI4x16x", off_name, a_dset, 0)
    snod += b"\x00" * (SNOD - len(snod))

    with open(path, "wb") as f:
        for part in (sb, root, btree, heap, heap_data, snod, dset, gheap):
            f.write(part)
        f.write(data.tobytes())


# ---- reader -----------------------------------------------------------------------------------
class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        if buf[:8] != SIGNATURE or buf[8] != 0 or buf[13] != 8 or buf[14] != 8:
            raise ValueError("not a version-0-superblock HDF5 file with 8-byte offsets")
        self.root_btree, self.root_heap = struct.unpack_from("<QQ", buf, 24 + 32 + 24)

    def _heap_name(self, heap_addr: int, off: int) -> str:
        seg = struct.unpack_from("<Q", self.b, heap_addr + 24)[0]
        end = self.b.index(b"\x00", seg + off)
        return self.b[seg + off:end].decode("utf-8")

    def _group_entries(self, btree: int, heap: int):
        assert self.b[btree:btree + 4] == b"TREE"
        level, used = struct.unpack_from("<BH", self.b, btree + 5)
        p = btree + 24
        for i in range(used):
            child = struct.unpack_from("<Q", self.b, p + 8 + 16 * i)[0]
            if level > 0:
                yield from self._group_entries(child, heap)
                continue
            assert self.b[child:child + 4] == b"SNOD"
            n = struct.unpack_from("<H", self.b, child + 6)[0]
            for j in range(n):
                off, oh = struct.unpack_from("<QQ", self.b, child + 8 + 40 * j)
                yield self._heap_name(heap, off), oh

    def _messages(self, oh: int):
        ver, nmsg, _, size = struct.unpack_from("<BxHII", self.b, oh)
        if ver != 1:
            raise ValueError("only version-1 object headers are supported")
        blocks = [(oh + 16, size)]
        while blocks:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end:
                mtype, msize, _ = struct.unpack_from("<HHB", self.b, p)
                body = self.b[p + 8:p + 8 + msize]
                if mtype == 0x0010:          # continuation
                    blocks.append(struct.unpack_from("<QQ", body))
                else:
                    yield mtype, body
                p += 8 + msize

    def _decode(self, dt: bytes, raw: bytes):
        cls = dt[0] & 0x0F
        size = struct.unpack_from("<I", dt, 4)[0]
        if cls == 0:
            signed = bool(dt[1] & 0x08)
            return int.from_bytes(raw[:size], "little", signed=signed)
        if cls == 1:
            return float(np.frombuffer(raw[:size], dtype=f"<f{size}")[0])
        if cls == 3:
            return raw[:size].split(b"\x00")[0].decode("utf-8")
        if cls == 9:
            ln, addr, idx = struct.unpack_from("<IQI", raw)
            assert self.b[addr:addr + 4] == b"GCOL"
            p = addr + 16
            while True:
                i, _, osz = struct.unpack_from("<HH4xQ", self.b, p)
                if i == idx:
                    return self.b[p + 16:p + 16 + ln].decode("utf-8")
                if i == 0:
                    raise ValueError("global heap object not found")
                p += 16 + (osz + 7) // 8 * 8
        raise ValueError(f"unsupported attribute datatype class {cls}")

    def dataset(self, name: str):
        for nm, oh in self._group_entries(self.root_btree, self.root_heap):
            if nm != name:
                continue
            shape, dtype, addr, attrs = None, None, None, {}
            for mtype, body in self._messages(oh):
                if mtype == 0x0001:
                    rank = body[1]
                    shape = struct.unpack_from(f"<{rank}Q", body, 8) if rank else ()
                elif mtype == 0x0003:
                    if (body[0] & 0x0F) != 1 or body[1] & 1:
                        raise ValueError("dataset is not little-endian floating point")
                    dtype = np.dtype(f"<f{struct.unpack_from('<I', body, 4)[0]}")
                elif mtype == 0x0008:
                    if body[0] != 3 or body[1] != 1:
                        raise ValueError("only contiguous version-3 layouts are supported")
                    addr = struct.unpack_from("<Q", body, 2)[0]
                elif mtype == 0x000C:
                    _, nsz, dsz, ssz = struct.unpack_from("<BxHHH", body)
                    p = 8
                    aname = body[p:p + nsz].split(b"\x00")[0].decode("utf-8")
                    p += (nsz + 7) // 8 * 8
                    adt = body[p:p + dsz]
                    p += (dsz + 7) // 8 * 8 + (ssz + 7) // 8 * 8
                    attrs[aname] = self._decode(adt, body[p:])
            n = int(np.prod(shape)) if shape else 1
            arr = np.frombuffer(self.b, dtype=dtype, count=n, offset=addr).reshape(shape) if n else \
                np.zeros(shape, dtype=dtype)
            return arr.copy(), attrs
        raise KeyError(name)


def read_features(path, name: str = "mfccs"):
    """Returns (array, attrs) -- what ``file['mfccs'][:]`` and ``dict(file['mfccs'].attrs)`` give
    (/root/reference/src/trainer.py:110-114, /root/reference/src/preprocess.py:172-176)."""
    with open(path, "rb") as f:
        return _Reader(f.read()).dataset(name)
