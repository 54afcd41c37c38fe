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

import os
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


_DT_F32, _DT_I64, _DT_VLEN_UTF8, _DS_SCALAR = _dt_f32(), _dt_i64(), _dt_vlen_utf8(), _ds_simple(())
_MSG_FILL = _message(0x0005, struct.pack("<BBBBI", 2, 2, 2, 1, 0))


# ---- writer -----------------------------------------------------------------------------------
GCOL_MAX_OBJECTS = 60000          # object indices in a global heap collection are 16-bit
GCOL_MAX_BYTES = 1 << 26


class _StringHeap:
    """Variable-length strings go to global heap collections placed after the object headers."""

    def __init__(self):
        self.collections = [[]]       # lists of utf-8 byte strings
        self.sizes = [16]

    def add(self, raw: bytes):
        need = 16 + (len(raw) + 7) // 8 * 8
        if len(self.collections[-1]) >= GCOL_MAX_OBJECTS or self.sizes[-1] + need + 16 > GCOL_MAX_BYTES:
            self.collections.append([])
            self.sizes.append(16)
        self.collections[-1].append(raw)
        self.sizes[-1] += need
        return len(self.collections) - 1, len(self.collections[-1]), len(raw)    # (collection, index, length)

    def collection_bytes(self, c: int) -> int:
        return max(4096, (self.sizes[c] + 16 + 7) // 8 * 8)

    def addresses(self, start: int):
        out, a = [], start
        for c in range(len(self.collections)):
            out.append(a)
            a += self.collection_bytes(c)
        return out, a

    def encode(self, c: int) -> bytes:
        size = self.collection_bytes(c)
        body = b"".join(struct.pack("<HH4xQ", i, 1, len(raw)) + _pad8(raw)
                        for i, raw in enumerate(self.collections[c], start=1))
        free = size - 16 - len(body)
        return b"GCOL" + struct.pack("<B3xQ", 1, size) + body + struct.pack("<HH4xQ", 0, 0, free) + b"\x00" * (free - 16)

    @property
    def empty(self) -> bool:
        return not self.collections[0]


def write_datasets(path, datasets) -> None:
    """Write a new HDF5 file whose root group holds ``datasets``: an iterable of
    ``(name, data, attrs)`` with at most eight entries.  ``data`` is a float32 or int64 array
    (stored C-contiguous) or a list of str (a 1-D variable-length UTF-8 dataset); ``attrs`` maps
    names to Python ints (int64 scalars) or strs (variable-length UTF-8 scalars)."""
    datasets = sorted(((n, d, dict(a or {})) for n, d, a in datasets), key=lambda t: t[0].encode("utf-8"))
    if not 1 <= len(datasets) <= 8:
        raise ValueError("between one and eight datasets per file")
    strings = _StringHeap()
    prepared = []
    for name, data, attrs in datasets:
        if isinstance(data, (list, tuple)) and all(isinstance(x, str) for x in data):
            refs = [strings.add(x.encode("utf-8")) for x in data]
            prepared.append(dict(name=name, kind="str", shape=(len(refs),), refs=refs, nbytes=16 * len(refs), attrs=attrs))
        else:
            arr = np.asarray(data)
            if arr.dtype.kind == "f":
                arr, kind = np.ascontiguousarray(arr, dtype="<f4"), "f32"
            elif arr.dtype.kind in "iu":
                arr, kind = np.ascontiguousarray(arr, dtype="<i8"), "i64"
            else:
                raise TypeError(f"dataset {name!r}: unsupported dtype {arr.dtype}")
            prepared.append(dict(name=name, kind=kind, shape=arr.shape, arr=arr, nbytes=arr.nbytes, attrs=attrs))
        aref = {}
        for k, v in attrs.items():
            if isinstance(v, str):
                aref[k] = strings.add(v.encode("utf-8"))
            elif not (isinstance(v, (int, np.integer)) and not isinstance(v, bool)):
                raise TypeError(f"attribute {k!r}: only int and str are supported, got {type(v).__name__}")
        prepared[-1]["aref"] = aref

    def dataset_header(d):
        """The dataset's object header with zeroed addresses, and where they go: the offset of
        the data address and (offset, collection) of every global-heap address."""
        dt = {"f32": _DT_F32, "i64": _DT_I64, "str": _DT_VLEN_UTF8}[d["kind"]]
        msgs = [
            _message(0x0001, _ds_simple(d["shape"])),
            _message(0x0003, dt, flags=1),                           # constant message
            _MSG_FILL,                                               # fill value: default, alloc late
        ]
        pos = 16 + sum(len(m) for m in msgs)                         # 16: object header prefix
        data_at = pos + 8 + 2                                        # message header, version, class
        msgs.append(_message(0x0008, struct.pack("<BBQQ", 3, 1, 0, d["nbytes"])))
        pos += len(msgs[-1])
        heap_at = []
        for k, v in d["attrs"].items():
            if isinstance(v, str):
                c, idx, ln = d["aref"][k]
                m = _attribute(k, _DT_VLEN_UTF8, _DS_SCALAR, struct.pack("<IQI", ln, 0, idx))
                heap_at.append((pos + len(m) - 16 + 4, c))           # the 16 data bytes close the message
            else:
                m = _attribute(k, _DT_I64, _DS_SCALAR, struct.pack("<q", int(v)))
            msgs.append(m)
            pos += len(m)
        return bytearray(_object_header(b"".join(msgs), len(msgs))), data_at, heap_at

    # fixed-size pieces first, to know the addresses
    SB, ROOT_OH, BTREE, HEAP_HDR, SNOD = 96, 40, 544, 32, 328
    heap_data, name_off = b"\x00" * 8, []
    for d in prepared:
        name_off.append(len(heap_data))
        heap_data += _pad8(d["name"].encode("utf-8") + b"\x00")
    heap_data += struct.pack("<QQ", 1, 16)                           # one free block closing the segment
    a_root = SB
    a_btree = a_root + ROOT_OH
    a_heap = a_btree + BTREE
    a_heap_data = a_heap + HEAP_HDR
    a_snod = a_heap_data + len(heap_data)
    a = a_snod + SNOD
    a_dset, headers = [], []
    for d in prepared:
        a_dset.append(a)
        headers.append(dataset_header(d))
        a += len(headers[-1][0])
    gaddr, a = strings.addresses(a) if not strings.empty else ([], a)
    a_data = []
    for d in prepared:
        a_data.append(a)
        a += (d["nbytes"] + 7) // 8 * 8
    eof = a

    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQI4xQQ", 0, a_root, 1, a_btree, a_heap)     # root symbol-table entry
    assert len(sb) == SB
    root = _object_header(_message(0x0011, struct.pack("<QQ", a_btree, a_heap)), 1)
    assert len(root) == ROOT_OH
    btree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF)
    btree += struct.pack("<QQQ", 0, a_snod, name_off[-1])            # keys: "" and the largest name
    btree += b"\x00" * (BTREE - len(btree))
    heap = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), len(heap_data) - 16, a_heap_data)
    assert len(heap) == HEAP_HDR
    snod = b"SNOD" + struct.pack("<BxH", 1, len(prepared))
    for off, ad in zip(name_off, a_dset):
        snod += struct.pack("<QQI4x16x", off, ad, 0)
    snod += b"\x00" * (SNOD - len(snod))

    # Written under a temporary name and renamed when complete.  The reference's reader opens every
    # file under the feature directory (/root/reference/src/trainer.py:106-110), so the temporary
    # lives in a sibling "<directory>.partial" on the same file system, where no reader walks; it is
    # removed if the write fails.
    tmp, side = _staging_path(path)
    try:
        with open(tmp, "wb") as f:
            for part in (sb, root, btree, heap, heap_data, snod):
                f.write(part)
            for d, ad, (hdr, data_at, heap_at) in zip(prepared, a_data, headers):
                struct.pack_into("<Q", hdr, data_at, ad if d["nbytes"] else UNDEF)
                for at, c in heap_at:
                    struct.pack_into("<Q", hdr, at, gaddr[c])
                f.write(hdr)
            for c in range(len(gaddr)):
                f.write(strings.encode(c))
            for d in prepared:
                if d["kind"] == "str":
                    raw = b"".join(struct.pack("<IQI", ln, gaddr[c], idx) for c, idx, ln in d["refs"])
                else:
                    raw = d["arr"].tobytes()
                f.write(raw + b"\x00" * (-len(raw) % 8))
        os.replace(tmp, path)
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise
    finally:
        if side is not None:
            try:
                os.rmdir(side)          # only when this writer left it empty
            except OSError:
                pass


def _staging_path(path):
    """(temporary file, side directory or None) for a file that will be renamed to ``path``."""
    path = os.fspath(path)
    d, base = os.path.split(path)
    d = d.rstrip("/") if len(d) > 1 else d
    if d in ("", "/"):
        return path + ".tmp", None
    side = d + ".partial"
    os.makedirs(side, exist_ok=True)
    return os.path.join(side, f"{base}.{os.getpid()}"), side


def write_features_native(path, data: np.ndarray, attrs: Dict[str, AttrValue], name: str = "mfccs") -> None:
    """write_features for a 2-D dataset through the native writer of libmfcc_hip.so
    (mfcc_host_write_feature_file): the same bytes, without the interpreter -- and without its
    lock, so files can be written from several threads at once."""
    import ctypes as C

    from . import _lib

    arr = np.ascontiguousarray(data, dtype="<f4")
    if arr.ndim != 2:
        raise ValueError("the native writer takes 2-D datasets")
    n = len(attrs)
    names = (C.c_char_p * n)(*[k.encode("utf-8") for k in attrs])
    is_str = (C.c_int32 * n)(*[1 if isinstance(v, str) else 0 for v in attrs.values()])
    for k, v in attrs.items():
        if not isinstance(v, str) and not (isinstance(v, (int, np.integer)) and not isinstance(v, bool)):
            raise TypeError(f"attribute {k!r}: only int and str are supported, got {type(v).__name__}")
    ints = (C.c_int64 * n)(*[0 if isinstance(v, str) else int(v) for v in attrs.values()])
    strs = (C.c_char_p * n)(*[v.encode("utf-8") if isinstance(v, str) else None for v in attrs.values()])
    rc = _lib.lib().mfcc_host_write_feature_file(str(path).encode("utf-8"), name.encode("utf-8"), arr.ctypes.data,
                                                 arr.shape[0], arr.shape[1], n, names, is_str, ints, strs)
    if rc != 0:
        raise OSError(_lib.lib().mfcc_host_file_error().decode("utf-8", "replace"))


def write_many_features_native(paths, arrays, attr_rows, name: str = "mfccs") -> None:
    """One native call for many files: file i holds the 2-D float32 array ``arrays[i]`` with the
    attributes ``attr_rows[i]`` (dicts with the same keys and value kinds for every file).  The
    interpreter is only needed to line the arguments up; the files are written without it."""
    import ctypes as C

    from . import _lib

    n = len(paths)
    if n == 0:
        return
    keys = list(attr_rows[0].keys())
    kinds = [isinstance(attr_rows[0][k], str) for k in keys]
    arrs = [np.ascontiguousarray(a, dtype="<f4") for a in arrays]
    cols = arrs[0].shape[1]
    if any(a.ndim != 2 or a.shape[1] != cols for a in arrs):
        raise ValueError("the native writer takes 2-D datasets of one width")
    na = len(keys)
    c_paths = (C.c_char_p * n)(*[str(p).encode("utf-8") for p in paths])
    c_data = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    c_rows = (C.c_int64 * n)(*[a.shape[0] for a in arrs])
    c_names = (C.c_char_p * na)(*[k.encode("utf-8") for k in keys])
    c_kinds = (C.c_int32 * na)(*[1 if s else 0 for s in kinds])
    ints, strs = [], []
    for row in attr_rows:
        if list(row.keys()) != keys:
            raise ValueError("every file must carry the same attributes")
        for k, is_str in zip(keys, kinds):
            v = row[k]
            if isinstance(v, str) != is_str:
                raise TypeError(f"attribute {k!r} changes kind between files")
            ints.append(0 if is_str else int(v))
            strs.append(v.encode("utf-8") if is_str else None)
    c_ints = (C.c_int64 * (n * na))(*ints)
    c_strs = (C.c_char_p * (n * na))(*strs)
    rc = _lib.lib().mfcc_host_write_feature_files(n, c_paths, name.encode("utf-8"), c_data, c_rows, cols, na, c_names,
                                                  c_kinds, c_ints, c_strs)
    if rc != 0:
        raise OSError(_lib.lib().mfcc_host_file_error().decode("utf-8", "replace"))


def write_feature_batch_native(paths, base: np.ndarray, row_offsets, int_names, int_rows, str_name: str,
                               strings, name: str = "mfccs", n_threads: int = 1) -> None:
    """The per-utterance files of one GPU batch in one native call, straight from the batch's host
    buffer: file i (``paths[i]``) holds rows ``row_offsets[i] : row_offsets[i + 1]`` of the 2-D
    float32 array ``base`` as dataset ``name``, the int64 attributes ``int_names`` with the values
    ``int_rows[i]`` and then the string attribute ``str_name`` = ``strings[i]`` -- the order of
    /root/reference/src/preprocess.py:140-144 when ``int_names`` is (speaker_id, chapter_id,
    utterance_id, sample_rate) and ``str_name`` is transcript.  ``n_threads`` host threads create
    the files.  Nothing is sliced or copied per utterance in the interpreter."""
    import ctypes as C

    from . import _lib

    n = len(paths)
    if n == 0:
        return
    if base.dtype != np.float32 or base.ndim != 2 or not base.flags.c_contiguous:
        raise ValueError("the batch buffer must be a C-contiguous 2-D float32 array")
    fo = np.ascontiguousarray(row_offsets, dtype=np.int64)
    ints = np.ascontiguousarray(int_rows, dtype=np.int64)
    ni = len(int_names)
    if fo.shape != (n + 1,) or ints.shape != (n, ni) or len(strings) != n:
        raise ValueError("one path, one row range, one row of integers and one string per file")
    if fo[0] < 0 or fo[-1] > base.shape[0] or (np.diff(fo) < 0).any():
        raise ValueError("row offsets outside the batch buffer")
    cols = int(base.shape[1])
    na = ni + 1
    ptrs = (base.ctypes.data + fo[:-1] * (cols * 4)).astype(np.uint64)
    rows = np.ascontiguousarray(np.diff(fo))
    all_ints = np.zeros((n, na), dtype=np.int64)
    all_ints[:, :ni] = ints
    strs = [None] * (n * na)
    strs[ni::na] = [s.encode("utf-8") for s in strings]
    c_paths = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    c_strs = (C.c_char_p * (n * na))(*strs)
    c_names = (C.c_char_p * na)(*[k.encode("utf-8") for k in list(int_names) + [str_name]])
    c_kinds = (C.c_int32 * na)(*([0] * ni + [1]))
    rc = _lib.lib().mfcc_host_write_feature_files_mt(n, c_paths, name.encode("utf-8"), ptrs.ctypes.data, rows.ctypes.data,
                                                     cols, na, c_names, c_kinds, all_ints.ctypes.data, c_strs,
                                                     int(n_threads))
    if rc != 0:
        raise OSError(_lib.lib().mfcc_host_file_error().decode("utf-8", "replace"))


def write_features(path, data: np.ndarray, attrs: Dict[str, AttrValue], name: str = "mfccs") -> None:
    """Write one dataset ``name`` (float32, C-contiguous, any rank >= 1) with scalar attributes
    (Python int -> int64, str -> variable-length UTF-8) to a new HDF5 file at ``path``."""
    write_datasets(path, [(name, np.ascontiguousarray(data, dtype="<f4"), attrs)])


# ---- reader -----------------------------------------------------------------------------------
class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        if buf[:8] != SIGNATURE or buf[8] != 0 or buf[13] != 8 or buf[14] != 8:
            raise ValueError("not a version-0-superblock HDF5 file with 8-byte offsets")
        self.root_btree, self.root_heap = struct.unpack_from("<QQ", buf, 24 + 32 + 24)
        self._gcol = {}

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
            index = self._gcol.get(addr)
            if index is None:                 # object index -> data offset, one pass per collection
                assert self.b[addr:addr + 4] == b"GCOL"
                index, p = {}, addr + 16
                end = addr + struct.unpack_from("<Q", self.b, addr + 8)[0]
                while p + 16 <= end:
                    i, _, osz = struct.unpack_from("<HH4xQ", self.b, p)
                    if i == 0:
                        break
                    index[i] = p + 16
                    p += 16 + (osz + 7) // 8 * 8
                self._gcol[addr] = index
            if idx not in index:
                raise ValueError("global heap object not found")
            return self.b[index[idx]:index[idx] + ln].decode("utf-8")
        raise ValueError(f"unsupported attribute datatype class {cls}")

    def names(self):
        return [nm for nm, _ in self._group_entries(self.root_btree, self.root_heap)]

    def dataset(self, name: str):
        for nm, oh in self._group_entries(self.root_btree, self.root_heap):
            if nm != name:
                continue
            shape, dtype, addr, attrs, vlen = None, None, None, {}, None
            for mtype, body in self._messages(oh):
                if mtype == 0x0001:
                    rank = body[1]
                    shape = struct.unpack_from(f"<{rank}Q", body, 8) if rank else ()
                elif mtype == 0x0003:
                    cls, size = body[0] & 0x0F, struct.unpack_from("<I", body, 4)[0]
                    if cls == 9:
                        vlen = bytes(body)
                    elif body[1] & 1 or cls not in (0, 1):
                        raise ValueError("dataset is not little-endian integer, floating point or vlen string")
                    else:
                        dtype = np.dtype(f"<f{size}") if cls == 1 else np.dtype(f"<{'i' if body[1] & 0x08 else 'u'}{size}")
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
            if vlen is not None:
                return [self._decode(vlen, self.b[addr + 16 * i:addr + 16 * i + 16]) for i in range(n)], attrs
            arr = np.frombuffer(self.b, dtype=dtype, count=n, offset=addr).reshape(shape) if n else \
                np.zeros(shape, dtype=dtype)
            return arr.copy(), attrs
        raise KeyError(name)


def read_dataset(path, name: str):
    """(values, attrs) of any dataset written by write_datasets: a numpy array, or a list of str."""
    with open(path, "rb") as f:
        return _Reader(f.read()).dataset(name)


def list_datasets(path):
    with open(path, "rb") as f:
        return _Reader(f.read()).names()


def read_features(path, name: str = "mfccs"):
    """Returns (array, attrs) -- what ``file['mfccs'][:]`` and ``dict(file['mfccs'].attrs)`` give
    (/root/reference/src/trainer.py:110-114, /root/reference/src/preprocess.py:172-176)."""
    with open(path, "rb") as f:
        return _Reader(f.read()).dataset(name)
