"""Consecutive ragged batches through the GPU with every host stage overlapped.

The reference's loop handles one utterance at a time: decode, ``librosa.feature.mfcc``, write one
file (/root/reference/src/preprocess.py:155-167).  With the arithmetic on the GPU the wall clock
of that loop is host work -- packing the batch for upload, the PCIe transfers, creating the files
-- so ``FeatureStream`` cuts a batch's life into three calls that different threads make:

    stage(waves)     packs the utterances into pinned staging buffers and starts their upload on a
                     copy stream; returns as soon as the last chunk is packed (staging.py)
    compute(staged)  makes the compute stream wait for the upload, runs the kernels and starts the
                     download of the (total_frames, n_mfcc) features into a pinned host slot
    fetch(computed)  waits for that download; the features of utterance i are rows
                     ``fo[i] : fo[i + 1]`` of the returned array, which lives in the slot until
                     ``release(computed)``

While batch k is being computed, batch k + 1 is being packed and uploaded and batch k - 1's files
are being written from its slot.
"""
from __future__ import annotations

import queue
import threading
import warnings
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import staging
from .config import MfccConfig
from .feature import ParameterError, get_plan


class _Staged:
    __slots__ = ("pcm", "so", "event", "lens")

    def __init__(self, pcm, so, event, lens):
        self.pcm, self.so, self.event, self.lens = pcm, so, event, lens


class _Computed:
    __slots__ = ("slot", "fo", "event", "flag", "n_utts")

    def __init__(self, slot, fo, event, flag, n_utts):
        self.slot, self.fo, self.event, self.flag, self.n_utts = slot, fo, event, flag, n_utts


class FeatureStream:
    """``config``: the MFCC configuration; ``try_int16``: float waveforms that lie on the int16 grid
    are converted to 16-bit PCM while they are packed (half the upload, bit-identical features);
    with ``"auto"`` that is tried until a batch turns out to be off the grid (its first chunk
    tells) and not again afterwards;
    ``check_finite``: librosa's ``valid_audio`` finiteness check, run on the device;
    ``slots``: pinned host buffers for finished batches (one being filled, one being written,
    one spare)."""

    def __init__(self, config: MfccConfig, device=None, *, try_int16=False, check_finite: bool = True,
                 slots: int = 3):
        self.plan = get_plan(config, device)
        self.config = config
        self.device = self.plan.device
        self.auto_int16 = try_int16 == "auto"
        self.uploads = {"i16": 0, "f32": 0}      # batches uploaded as 16-bit PCM / as float32
        self.try_int16 = bool(try_int16)
        self.check_finite = bool(check_finite)
        self._free: "queue.Queue" = queue.Queue()
        for _ in range(max(2, int(slots))):
            self._free.put(None)             # allocated (and grown) on first use
        self._lock = threading.Lock()

    # -- stage: any thread ---------------------------------------------------------------------
    def stage(self, waves: Sequence[np.ndarray]) -> _Staged:
        chans = [np.asarray(w).reshape(-1) for w in waves]
        lens = np.fromiter((c.shape[0] for c in chans), dtype=np.int64, count=len(chans))
        so = np.zeros(len(chans) + 1, dtype=np.int64)
        np.cumsum(lens, out=so[1:])
        pcm16 = len(chans) > 0 and all(c.dtype == np.int16 for c in chans)
        if not pcm16:
            for c in chans:
                if not np.issubdtype(c.dtype, np.floating):
                    raise ParameterError("Audio data must be floating-point")
            chans = [c if c.dtype == np.float32 else c.astype(np.float32) for c in chans]
        mode = "i16" if pcm16 else ("f32_to_i16" if self.try_int16 else "f32")
        pcm, kind, ev = staging.upload_ragged(chans, so, self.device, mode, join=False)
        if mode == "f32_to_i16" and kind != "i16" and self.auto_int16:
            self.try_int16 = False           # this source is not 16-bit audio: stop trying
        self.uploads[kind] += 1
        return _Staged(pcm, so, ev, lens)

    # -- compute: the thread that owns the compute stream ---------------------------------------
    def compute(self, st: _Staged) -> _Computed:
        cfg = self.config
        if cfg.center and len(st.lens) and int(st.lens.min()) < cfg.n_fft:
            warnings.warn(f"n_fft={cfg.n_fft} is too large for input signal of length={int(st.lens.min())}", stacklevel=2)
        if cfg.center and cfg.pad_mode == "reflect" and len(st.lens) and int(st.lens.min()) == 0:
            raise ParameterError("can't extend an empty signal with pad_mode='reflect'")
        fo = self.plan.frame_offsets(st.so)
        F = int(fo[-1])
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if st.event is not None:
                cur.wait_event(st.event)
            st.pcm.record_stream(cur)
            flag = None
            if self.check_finite and st.pcm.dtype == torch.float32:
                flag = torch.isfinite(st.pcm).all()
            d_so = torch.from_numpy(st.so).to(self.device, non_blocking=True)
            d_fo = torch.from_numpy(fo).to(self.device, non_blocking=True)
            out = self.plan.execute(st.pcm, d_so, d_fo, F)
            slot = self._free.get()                     # blocks while every slot is still being written
            need = F + 1                                # the last row takes the finiteness flag
            if slot is None or slot.shape[0] < need:
                slot = torch.empty((max(need, int(need * 1.25)), cfg.n_mfcc), dtype=torch.float32, pin_memory=True)
            slot[:F].copy_(out, non_blocking=True)
            host_flag = None
            if flag is not None:
                host_flag = slot[F, :1]
                host_flag.copy_(flag.to(torch.float32).reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cur)
        return _Computed(slot, fo, ev, host_flag, len(st.lens))

    # -- fetch / release: any thread --------------------------------------------------------------
    def fetch(self, comp: _Computed) -> Tuple[np.ndarray, np.ndarray]:
        comp.event.synchronize()
        if comp.flag is not None and float(comp.flag[0]) != 1.0:
            self.release(comp)
            raise ParameterError("Audio buffer is not finite everywhere")      # librosa.util.valid_audio
        F = int(comp.fo[-1])
        return comp.slot.numpy()[:F], comp.fo

    def release(self, comp: _Computed) -> None:
        if comp.slot is not None:
            self._free.put(comp.slot)
            comp.slot = None


def features_of_batches(batches, config: MfccConfig, device=None, *, try_int16=False,
                        check_finite: bool = True, ahead: int = 2, timeline: Optional[dict] = None):
    """Generator over ``(tag, features (total_frames, n_mfcc), frame_offsets, release)`` for an iterable
    of ``(tag, waves)``: a staging thread packs and uploads up to ``ahead`` batches in advance, the
    calling thread computes batch k + 1 before it hands out batch k.  ``features`` lives in a
    pinned slot until ``release()`` is called (from any thread).  ``timeline``: a dict that receives
    the (start, end) wall-clock pairs of every stage / compute / fetch call and, under 'uploads', how many
    batches went to the device as 16-bit PCM and how many as float32."""
    import time

    fs = FeatureStream(config, device, try_int16=try_int16, check_finite=check_finite, slots=ahead + 2)
    if timeline is not None:
        timeline.update({"stage": [], "compute": [], "fetch": [], "uploads": fs.uploads})
    staged: "queue.Queue" = queue.Queue(maxsize=max(1, int(ahead)))
    stop = threading.Event()
    _END = object()

    def put(x) -> bool:
        while not stop.is_set():
            try:
                staged.put(x, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def stager():
        try:
            for tag, waves in batches:
                t0 = time.perf_counter()
                st = fs.stage(waves)
                if timeline is not None:
                    timeline["stage"].append((t0, time.perf_counter()))
                if stop.is_set() or not put((tag, st)):
                    return
                del st
        except BaseException as e:              # reaches the consumer, like a worker's exception does
            put(e)
            return
        put(_END)

    th = threading.Thread(target=stager, name="mfcc-stager", daemon=True)
    th.start()
    prev: Optional[Tuple[object, _Computed]] = None
    try:
        while True:
            got = staged.get()
            if isinstance(got, BaseException):
                raise got
            done = got is _END
            cur = None
            if not done:
                tag, st = got
                t0 = time.perf_counter()
                cur = (tag, fs.compute(st))
                if timeline is not None:
                    timeline["compute"].append((t0, time.perf_counter()))
                del st, got
            if prev is not None:
                ptag, comp = prev
                t0 = time.perf_counter()
                host, fo = fs.fetch(comp)
                if timeline is not None:
                    timeline["fetch"].append((t0, time.perf_counter()))
                yield ptag, host, fo, (lambda c=comp: fs.release(c))
            prev = cur
            if done:
                break
    finally:
        stop.set()
        th.join(timeout=5.0)


def mfcc_of_batches(batches, *, n_fft: int, hop_length: int, n_mfcc: int, device=None, try_int16: bool = False) -> List:
    """Convenience for tests: the per-utterance (F, n_mfcc) arrays of every batch, copied out."""
    from .feature import make_config

    cfg = make_config(n_fft=n_fft, hop_length=hop_length, n_mfcc=n_mfcc)
    res = []
    for tag, host, fo, release in features_of_batches(batches, cfg, device, try_int16=try_int16):
        res.append((tag, [host[int(a):int(b)].copy() for a, b in zip(fo[:-1], fo[1:])]))
        release()
    return res
