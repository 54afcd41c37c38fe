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

CHUNK_BYTES = 64 << 20
N_THREADS = max(1, min(8, (os.cpu_count() or 2) - 1))


class _Stager:
    def __init__(self, device: torch.device):
        self.device = device
        self.bufs = [torch.empty(CHUNK_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        self.events = [None, None]
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


def _pack_range(chans: List[np.ndarray], so: np.ndarray, dst: np.ndarray, c0: int, a: int, b: int, to_i16: bool) -> bool:
    """Copies global elements [a, b) of the concatenation into dst[a - c0 : b - c0].  Returns
    False if a float sample is off the int16 grid (to_i16 only)."""
    u = int(np.searchsorted(so, a, side="right")) - 1
    ok = True
    while a < b:
        while so[u + 1] <= a:
            u += 1
        e = min(b, int(so[u + 1]))
        src = chans[u][a - int(so[u]): e - int(so[u])]
        out = dst[a - c0: e - c0]
        if to_i16:
            if src.dtype == np.float32 and src.flags.c_contiguous:
                good = _lib.lib().mfcc_host_pcm16_from_float(src.ctypes.data, src.size, out.ctypes.data) == 1
            else:
                scaled = src * np.float32(32768.0)              # exact: a power of two
                r = np.rint(scaled)
                good = bool(np.array_equal(r, scaled) and (r.size == 0 or (r.min() >= -32768.0 and r.max() <= 32767.0)))
                if good:
                    np.copyto(out, r, casting="unsafe")
            if not good:
                ok = False
                break
        else:
            np.copyto(out, src)
        a = e
    return ok


def upload_ragged(chans: Sequence[np.ndarray], so: np.ndarray, device: torch.device, mode: str = "f32") -> Tuple[torch.Tensor, str]:
    """chans: 1-D host arrays (float32, or int16 for mode 'i16'); so: int64 offsets of their
    concatenation.  Returns (device tensor of so[-1] elements, 'f32' or 'i16' as uploaded)."""
    total = int(so[-1])
    to_i16 = mode == "f32_to_i16"
    np_dtype = np.int16 if mode in ("i16", "f32_to_i16") else np.float32
    t_dtype = torch.int16 if np_dtype == np.int16 else torch.float32
    if total == 0:
        return torch.zeros(1, dtype=t_dtype, device=device), ("i16" if np_dtype == np.int16 else "f32")
    st = _stager(device)
    esize = np.dtype(np_dtype).itemsize
    chunk = CHUNK_BYTES // esize
    with st.lock, torch.cuda.device(device):
        cur = torch.cuda.current_stream(device)
        pcm = torch.empty(total, dtype=t_dtype, device=device)
        pcm.record_stream(st.stream)
        st.stream.wait_stream(cur)
        ok = True
        for k, c0 in enumerate(range(0, total, chunk)):
            c1 = min(total, c0 + chunk)
            which = k & 1
            if st.events[which] is not None:
                st.events[which].synchronize()           # the buffer's previous upload has left it
            host = st.bufs[which].numpy().view(np_dtype)[: c1 - c0]
            step = -(-(c1 - c0) // N_THREADS)
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
        cur.wait_stream(st.stream)
    if not ok:                                           # off the int16 grid: plain float32 upload
        return upload_ragged(chans, so, device, "f32")
    return pcm, ("i16" if np_dtype == np.int16 else "f32")
