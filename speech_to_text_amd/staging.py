"""Host -> device staging of a ragged PCM batch.

The reference hands every utterance to librosa as a host array
(/root/reference/src/preprocess.py:132-134); the GPU path wants one concatenated device buffer.
Packing 10 hours of float32 audio (2.3 GB) with one ``numpy`` copy per utterance into fresh
pageable memory and uploading that costs several times the PCIe transfer itself, so the batch is
streamed instead: a few threads pack 64 MiB chunks into a reusable pinned double buffer while the
previous chunk is on its way to the device on a copy stream.

``mode``:
    'f32'         float32 waveforms, uploaded as they are
    'i16'         int16 PCM arrays, uploaded as they are
    'f32_to_i16'  float32 waveforms that are expected to lie on the int16 grid (value * 32768 an
                  integer in [-32768, 32767], what a 16-bit FLAC decodes to): converted while
                  packing, half the upload; if any sample is off the grid the batch is uploaded
                  again as float32
"""
from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

# Chunk size and packing threads were swept on an MI355X host (tools/sweep_stage.py, 16 CPUs for
# the process): 64 MiB chunks packed by 6 threads keep the PCIe link busy (52 GB/s of float32);
# smaller chunks or more threads lose to hand-over costs and to contention with the other stages.
CHUNK_BYTES = int(os.environ.get("MFCC_STAGE_CHUNK_MIB", "64")) << 20
N_BUFFERS = 3


def host_threads(limit: int = 16) -> int:
    """Host threads for packing and file writing: the CPUs this process may use (its affinity
    mask, and OMP_NUM_THREADS where a scheduler sets it as the job's share of a larger machine),
    one left for the interpreter, at most ``limit``."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 2
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    if share > 0:
        n = min(n, share)
    return max(1, min(limit, n - 1))


N_THREADS = int(os.environ.get("MFCC_STAGE_THREADS", "0")) or host_threads(6)


class _Stager:
    def __init__(self, device: torch.device):
        self.device = device
        self.bufs = [torch.empty(CHUNK_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(N_BUFFERS)]
        self.events = [None] * N_BUFFERS
        self.turn = 0                  # next staging buffer (they rotate across calls)
        self.stream = torch.cuda.Stream(device)
        self.pool = ThreadPoolExecutor(max_workers=N_THREADS)
        self.lock = threading.Lock()


_STAGERS = {}
_STAGERS_LOCK = threading.Lock()


def _stager(device: torch.device) -> _Stager:
    with _STAGERS_LOCK:
        st = _STAGERS.get(device.index)
        if st is None:
            st = _STAGERS[device.index] = _Stager(device)
        return st


_MODES = {"f32": 0, "i16": 1, "f32_to_i16": 2}


def _pack_range(chans: List[np.ndarray], so: np.ndarray, dst: np.ndarray, c0: int, a: int, b: int, to_i16: bool) -> bool:
    """Copies global elements [a, b) of the concatenation into dst[a - c0 : b - c0] (numpy path, for
    sources the native packer does not take: non-contiguous or of another dtype).  Returns False
    if a float sample is off the int16 grid (to_i16 only)."""
    u = int(np.searchsorted(so, a, side="right")) - 1
    ok = True
    while a < b:
        while so[u + 1] <= a:
            u += 1
        e = min(b, int(so[u + 1]))
        src = chans[u][a - int(so[u]): e - int(so[u])]
        out = dst[a - c0: e - c0]
        if to_i16:
            scaled = np.asarray(src, dtype=np.float32) * np.float32(32768.0)              # exact: a power of two
            r = np.rint(scaled)
            good = bool(np.array_equal(r, scaled) and (r.size == 0 or (r.min() >= -32768.0 and r.max() <= 32767.0)))
            if good:
                np.copyto(out, r, casting="unsafe")
            else:
                ok = False
                break
        else:
            np.copyto(out, src)
        a = e
    return ok


def _source_pointers(chans: Sequence[np.ndarray], np_src_dtype) -> "np.ndarray | None":
    """Addresses of the utterances for the native packer, or None if one of them is not a
    C-contiguous array of the expected dtype."""
    ptrs = np.empty(len(chans), dtype=np.uint64)
    for i, c in enumerate(chans):
        if not isinstance(c, np.ndarray) or c.dtype != np_src_dtype or not c.flags.c_contiguous:
            return None
        ptrs[i] = c.__array_interface__["data"][0]
    return ptrs


def upload_ragged(chans: Sequence[np.ndarray], so: np.ndarray, device: torch.device, mode: str = "f32",
                  join: bool = True):
    """chans: 1-D host arrays (float32, or int16 for mode 'i16'); so: int64 offsets of their
    concatenation.  Returns (device tensor of so[-1] elements, 'f32' or 'i16' as uploaded) -- and,
    with ``join=False``, a third value: the event after which the tensor is complete.  With
    ``join=True`` the caller's current stream waits for the upload; with ``join=False`` nothing
    waits, so a staging thread can upload batch k + 1 while batch k is being computed, and whoever
    uses the tensor makes its stream wait for the event.

    The batch is cut into chunks of a few tens of MiB: the host threads pack chunk c into one of
    the pinned staging buffers (one native call per thread and chunk, mfcc_host_pack_range) while
    chunk c - 1 is on its way to the device on the copy stream.  The buffers rotate across calls,
    so consecutive batches stream through without a gap."""
    total = int(so[-1])
    to_i16 = mode == "f32_to_i16"
    np_dtype = np.int16 if mode in ("i16", "f32_to_i16") else np.float32
    t_dtype = torch.int16 if np_dtype == np.int16 else torch.float32
    kind = "i16" if np_dtype == np.int16 else "f32"
    if total == 0:
        pcm = torch.zeros(1, dtype=t_dtype, device=device)
        return (pcm, kind) if join else (pcm, kind, None)
    st = _stager(device)
    esize = np.dtype(np_dtype).itemsize
    chunk = CHUNK_BYTES // esize
    so = np.ascontiguousarray(so, dtype=np.int64)
    ptrs = _source_pointers(chans, np.int16 if mode == "i16" else np.float32)
    pack_native = _lib.lib().mfcc_host_pack_range if ptrs is not None else None
    n_utts = len(chans)
    ev = None
    with st.lock, torch.cuda.device(device):
        cur = torch.cuda.current_stream(device)
        pcm = torch.empty(total, dtype=t_dtype, device=device)
        pcm.record_stream(st.stream)
        # The allocator may hand out memory that work already queued on the current stream still
        # reads (a freed earlier batch): the copies start after that work.  This delays only the
        # copy stream, not the caller.
        st.stream.wait_stream(cur)
        ok = True
        for c0 in range(0, total, chunk):
            c1 = min(total, c0 + chunk)
            which = st.turn
            st.turn = (st.turn + 1) % N_BUFFERS
            if st.events[which] is not None:
                st.events[which].synchronize()           # the buffer's previous upload has left it
            host = st.bufs[which].numpy().view(np_dtype)[: c1 - c0]
            step = max(1 << 16, -(-(c1 - c0) // N_THREADS))
            if pack_native is not None:
                dst = host.ctypes.data
                jobs = [st.pool.submit(pack_native, ptrs.ctypes.data, so.ctypes.data, n_utts, dst, c0, a,
                                       min(c1, a + step), _MODES[mode]) for a in range(c0, c1, step)]
                rcs = [j.result() for j in jobs]
                if any(rc < 0 for rc in rcs):
                    raise RuntimeError("mfcc_host_pack_range: bad arguments")
                ok = all(rc == 1 for rc in rcs) and ok
            else:
                jobs = [st.pool.submit(_pack_range, chans, so, host, c0, a, min(c1, a + step), to_i16)
                        for a in range(c0, c1, step)]
                ok = all([j.result() for j in jobs]) and ok
            if not ok:
                break
            with torch.cuda.stream(st.stream):
                pcm[c0:c1].copy_(st.bufs[which].view(t_dtype)[: c1 - c0], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(st.stream)
            st.events[which] = ev
        if join:
            cur.wait_stream(st.stream)
    if not ok:                                           # off the int16 grid: plain float32 upload
        return upload_ragged(chans, so, device, "f32", join=join)
    return (pcm, kind) if join else (pcm, kind, ev)
