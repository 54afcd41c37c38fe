"""Host-side mirror of the librosa calls the reference makes, running on libmfcc_hip.

``mfcc`` keeps the keyword signature of ``librosa.feature.mfcc`` as called at
/root/reference/src/preprocess.py:134 (``y, n_fft, hop_length, n_mfcc`` plus the librosa
defaults); ``mfcc_batch`` is the ragged-batch form the GPU wants; ``melspectrogram`` and
``spectrogram`` are the intermediate taps (/root/reference/src/preprocess.py:111-113).

PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import collections
import ctypes as C
import threading
import warnings
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config import MfccConfig, ParameterError, make_config


def _device(device=None) -> torch.device:
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("speech_to_text_amd needs a ROCm GPU: there is no CPU fallback")
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


class MfccPlan:
    """Device tables for one MfccConfig on one GPU (``mfcc_plan_create``)."""

    def __init__(self, config: MfccConfig, device=None):
        self.config = config
        self.device = _device(device)
        self._handle = C.c_void_p()
        cfg_c, keep = config.to_c()
        self._cfg_c, self._keep = cfg_c, keep
        with torch.cuda.device(self.device):
            rc = _lib.lib().mfcc_plan_create(C.byref(cfg_c), C.byref(self._handle))
        if rc == _lib.MFCC_E_INVALID:     # the cases librosa reports as ParameterError
            raise ParameterError(_lib.last_error())
        _lib.check(rc)
        # one workspace per stream (include/mfcc_hip.h: a plan may be shared by several streams,
        # each with its own workspace); a few streams at most are kept
        self._ws: "collections.OrderedDict[int, torch.Tensor]" = collections.OrderedDict()
        self._ws_lock = threading.Lock()

    def close(self):
        if self._handle:
            _lib.lib().mfcc_plan_destroy(self._handle)
            self._handle = C.c_void_p()
            self._ws.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_profiling(self, enable: bool = True) -> None:
        _lib.check(_lib.lib().mfcc_plan_set_profiling(self._handle, 1 if enable else 0))

    def last_timings(self):
        """(front-end kernel ms, dB+DCT kernel ms) of the last execute, from HIP events."""
        a, b = C.c_float(), C.c_float()
        _lib.check(_lib.lib().mfcc_plan_last_timings(self._handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timing_history(self, max_n: int = 256):
        """Per-kernel times (ms) of the most recent executes since profiling was enabled, oldest
        first: two float32 arrays (front end, dB+DCT).  Waits for the last execute only."""
        a = (C.c_float * max_n)()
        b = (C.c_float * max_n)()
        n = _lib.lib().mfcc_plan_timing_history(self._handle, int(max_n), a, b)
        if n < 0:
            _lib.check(n)
        return np.array(a[:n], dtype=np.float32), np.array(b[:n], dtype=np.float32)

    def dct_rows(self, logmel: torch.Tensor) -> torch.Tensor:
        """DCT (the plan's type and norm), first n_mfcc coefficients and lifter of ``(rows, n_mels)``
        dB-scaled mel rows on this plan's device: ``(rows, n_mfcc)``.  Asynchronous on the current
        stream."""
        assert logmel.dtype == torch.float32 and logmel.is_contiguous() and logmel.device == self.device
        assert logmel.ndim == 2 and logmel.shape[1] == self.config.n_mels
        out = torch.empty((logmel.shape[0], self.config.n_mfcc), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mfcc_dct(self._handle, logmel.data_ptr(), int(logmel.shape[0]), out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
        return out

    @property
    def front_kernel(self) -> str:
        """Which front-end kernel the plan runs: 'fft<n_fft>' for the in-register FFT kernels (every multiple of 16 from 128 to 1024, and 2048) or 'dense_dft'."""
        return _lib.lib().mfcc_plan_front_kernel(self._handle).decode("ascii")

    # ------------------------------------------------------------------------------------
    def num_frames(self, n_samples: int) -> int:
        F = _lib.lib().mfcc_num_frames(C.byref(self._cfg_c), int(n_samples))
        if F < 0:
            raise ParameterError(_lib.last_error())
        return int(F)

    def frame_offsets(self, sample_offsets: np.ndarray) -> np.ndarray:
        so = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        fo = np.zeros_like(so)
        rc = _lib.lib().mfcc_batch_layout(C.byref(self._cfg_c), so.ctypes.data_as(C.POINTER(C.c_int64)),
                                          len(so) - 1, fo.ctypes.data_as(C.POINTER(C.c_int64)))
        if rc == _lib.MFCC_E_INVALID:
            raise ParameterError(_lib.last_error())
        _lib.check(rc)
        return fo

    MAX_STREAM_WORKSPACES = 4

    def workspace(self, total_frames: int, n_utts: int) -> torch.Tensor:
        """The workspace of the current stream, grown to fit the batch.  Executes on different
        streams get different workspaces, so they may overlap; two threads that share one stream
        must serialise their calls themselves (as for any stream-ordered buffer)."""
        need = int(_lib.lib().mfcc_workspace_bytes(self._handle, int(total_frames), int(n_utts)))
        key = int(torch.cuda.current_stream(self.device).cuda_stream)
        with self._ws_lock:
            ws = self._ws.pop(key, None)
            if ws is None or ws.numel() < need:
                ws = None
                ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws                       # most recently used last
            while len(self._ws) > self.MAX_STREAM_WORKSPACES:
                self._ws.popitem(last=False)
            return ws

    # ------------------------------------------------------------------------------------
    def execute(self, pcm: torch.Tensor, sample_offsets: torch.Tensor, frame_offsets: torch.Tensor,
                total_frames: int, *, group: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, melspec_out: Optional[torch.Tensor] = None,
                workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Run the whole pipeline on device-resident buffers; asynchronous on the current
        stream.  Returns the (total_frames, n_mfcc) float32 tensor.  ``pcm`` is float32, or int16
        PCM (value / 32768, the waveform torchaudio decodes LibriSpeech FLAC into)."""
        n_utts = sample_offsets.numel() - 1
        assert pcm.dtype in (torch.float32, torch.int16) and pcm.is_contiguous() and pcm.device == self.device
        fn = _lib.lib().mfcc_execute if pcm.dtype == torch.float32 else _lib.lib().mfcc_execute_pcm16
        assert sample_offsets.dtype == torch.int64 and frame_offsets.dtype == torch.int64
        if out is None:
            out = torch.empty((total_frames, self.config.n_mfcc), dtype=torch.float32, device=self.device)
        ws = workspace if workspace is not None else self.workspace(total_frames, n_utts)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(fn(
                self._handle, pcm.data_ptr(), sample_offsets.data_ptr(), frame_offsets.data_ptr(),
                group.data_ptr() if group is not None else None, n_utts, int(total_frames),
                out.data_ptr(), melspec_out.data_ptr() if melspec_out is not None else None,
                ws.data_ptr(), ws.numel(), stream))
        return out

    def execute_spectrogram(self, pcm, sample_offsets, frame_offsets, total_frames, *, out=None,
                            workspace=None) -> torch.Tensor:
        n_utts = sample_offsets.numel() - 1
        n_bins = 1 + self.config.n_fft // 2
        if out is None:
            out = torch.empty((total_frames, n_bins), dtype=torch.float32, device=self.device)
        ws = workspace if workspace is not None else self.workspace(total_frames, n_utts)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(_lib.lib().mfcc_execute_spectrogram(
                self._handle, pcm.data_ptr(), sample_offsets.data_ptr(), frame_offsets.data_ptr(),
                n_utts, int(total_frames), out.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        return out


_PLANS: "collections.OrderedDict" = collections.OrderedDict()
_PLANS_LOCK = threading.Lock()
MAX_CACHED_PLANS = 16


def get_plan(config: MfccConfig, device=None) -> MfccPlan:
    """Plans are cached per (config, device): librosa rebuilds its filterbank on every call
    (/root/reference/src/preprocess.py:134), this engine builds its tables once.  The cache keeps
    the MAX_CACHED_PLANS most recently used plans; an evicted plan frees its device tables when
    its last user drops it."""
    dev = _device(device)
    key = (config, dev.index)
    with _PLANS_LOCK:
        plan = _PLANS.pop(key, None)
        if plan is None:
            plan = MfccPlan(config, dev)
        _PLANS[key] = plan
        while len(_PLANS) > MAX_CACHED_PLANS:
            _PLANS.popitem(last=False)
        return plan


# --------------------------------------------------------------------------------------------
def _as_float_array(y):
    if isinstance(y, torch.Tensor):
        if not y.dtype.is_floating_point:
            raise ParameterError("Audio data must be floating-point")
        return y
    if not isinstance(y, np.ndarray):
        raise ParameterError("Audio data must be of type numpy.ndarray")
    if not np.issubdtype(y.dtype, np.floating):
        raise ParameterError("Audio data must be floating-point")
    return y


def _valid_audio(y) -> None:
    """librosa.util.valid_audio: floating point and finite everywhere."""
    if y.ndim == 0:
        raise ParameterError(f"Audio data must be at least one-dimensional, given y.shape={tuple(y.shape)}")
    finite = bool(torch.isfinite(y).all()) if isinstance(y, torch.Tensor) else bool(np.isfinite(y).all())
    if not finite:
        raise ParameterError("Audio buffer is not finite everywhere")


def _upload_ragged(chans: Sequence, device, dtype=np.float32, try_int16: bool = False) -> (torch.Tensor, np.ndarray):
    """Concatenates the utterances on the device.  Host arrays go through the pinned, chunked,
    multi-threaded path of staging.py; with try_int16 float waveforms on the int16 grid are
    uploaded as 16-bit PCM."""
    from . import staging

    lens = np.array([int(c.shape[-1]) for c in chans], dtype=np.int64)
    so = np.zeros(len(chans) + 1, dtype=np.int64)
    np.cumsum(lens, out=so[1:])
    total = int(so[-1])
    if dtype == np.int16:
        host = [np.asarray(c).reshape(-1) for c in chans]
        return staging.upload_ragged(host, so, device, "i16")[0], so
    if all(isinstance(c, torch.Tensor) and c.is_cuda for c in chans) and len(chans) > 0:
        pcm = torch.cat([c.reshape(-1).to(device=device, dtype=torch.float32) for c in chans]) if total else \
            torch.zeros(1, dtype=torch.float32, device=device)
        return pcm, so
    host = [(c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else np.asarray(c)).reshape(-1) for c in chans]
    pcm, _ = staging.upload_ragged(host, so, device, "f32_to_i16" if try_int16 else "f32")
    return pcm, so


def _run_batch(chans: Sequence, config: MfccConfig, *, groups: Optional[np.ndarray], device,
               what: str = "mfcc", try_int16: bool = False, check_finite: bool = False):
    """chans: 1-D arrays.  Returns (device tensor (total_frames, width), frame_offsets)."""
    plan = get_plan(config, device)
    for c in chans:
        if config.center and config.n_fft > c.shape[-1]:
            warnings.warn(f"n_fft={config.n_fft} is too large for input signal of length={c.shape[-1]}",
                          stacklevel=3)
        if config.center and config.pad_mode == "reflect" and c.shape[-1] == 0:
            raise ParameterError("can't extend an empty signal with pad_mode='reflect'")
    pcm16 = len(chans) > 0 and all(isinstance(c, np.ndarray) and c.dtype == np.int16 for c in chans)
    pcm, so = _upload_ragged(chans, plan.device, np.int16 if pcm16 else np.float32, try_int16=try_int16)
    if check_finite and pcm.dtype == torch.float32 and not bool(torch.isfinite(pcm).all()):
        raise ParameterError("Audio buffer is not finite everywhere")      # librosa.util.valid_audio
    fo = plan.frame_offsets(so)
    total_frames = int(fo[-1])
    d_so = torch.from_numpy(so).to(plan.device)
    d_fo = torch.from_numpy(fo).to(plan.device)
    if what == "spectrogram":
        out = plan.execute_spectrogram(pcm, d_so, d_fo, total_frames)
        return out, fo
    d_grp = None
    if groups is not None:
        g = np.ascontiguousarray(groups, dtype=np.int32)
        if g.shape != (len(chans),) or (len(g) and (g.min() < 0 or g.max() >= len(chans))):
            raise ParameterError("groups must hold one value in [0, n_utterances) per utterance")
        d_grp = torch.from_numpy(g).to(plan.device)
    if what == "melspectrogram":
        mel = torch.empty((total_frames, config.n_mels), dtype=torch.float32, device=plan.device)
        plan.execute(pcm, d_so, d_fo, total_frames, group=d_grp, melspec_out=mel)
        return mel, fo
    out = plan.execute(pcm, d_so, d_fo, total_frames, group=d_grp)
    return out, fo


def _split_kwargs(sr, n_mfcc, dct_type, norm, lifter, kwargs) -> MfccConfig:
    kw = dict(kwargs)
    if "norm" in kw:  # melspectrogram's **kwargs reach filters.mel, whose `norm` is the mel norm
        raise ParameterError("pass the mel filterbank norm as mel_norm=...")
    return make_config(sr=sr, n_mfcc=n_mfcc, dct_type=dct_type, norm=norm, lifter=lifter, **kw)


def _channels_of(y):
    """(..., T) -> list of 1-D channel views, leading shape."""
    lead = tuple(y.shape[:-1])
    flat = y.reshape(-1, y.shape[-1]) if y.ndim > 1 else y.reshape(1, -1)
    return [flat[i] for i in range(flat.shape[0])], lead


def mfcc(*, y=None, sr=22050, S=None, n_mfcc=20, dct_type=2, norm="ortho", lifter=0, device=None,
         validate=True, **kwargs) -> np.ndarray:
    """Drop-in for ``librosa.feature.mfcc(y=..., ...)``: returns float32 of shape
    ``y.shape[:-1] + (n_mfcc, n_frames)``.

    ``kwargs`` are melspectrogram / stft / filters.mel keywords (n_fft, hop_length, win_length,
    window, center, pad_mode, power, n_mels, fmin, fmax, htk; the mel norm is ``mel_norm``).
    As in librosa, power_to_db runs with ref=1.0, amin=1e-10, top_db=80.0 and the top_db maximum
    is taken over the whole array passed in (all channels).
    """
    if S is not None:
        # librosa's other spelling: S is the dB-scaled mel spectrogram, (..., n_mels, n_frames);
        # the melspectrogram keywords are not used
        return _mfcc_of_logmel(S, sr, n_mfcc, dct_type, norm, lifter, device)
    if y is None:
        raise ParameterError("y must be given")
    y = _as_float_array(y)
    if validate:
        _valid_audio(y)
    config = _split_kwargs(sr, n_mfcc, dct_type, norm, lifter, kwargs)
    chans, lead = _channels_of(y)
    out, fo = _run_batch(chans, config, groups=np.zeros(len(chans), dtype=np.int32), device=device)
    F = int(fo[1] - fo[0]) if len(chans) else 0
    res = out.reshape(len(chans), F, config.n_mfcc).cpu().numpy()
    return res.transpose(0, 2, 1).reshape(lead + (config.n_mfcc, F))


def _mfcc_of_logmel(S, sr, n_mfcc, dct_type, norm, lifter, device) -> np.ndarray:
    """``scipy.fftpack.dct(S, axis=-2, type, norm)[..., :n_mfcc, :]`` and the lifter, on the GPU.
    As in librosa, asking for more coefficients than S has bands returns one row per band.  The
    result is float32 whatever the dtype of S (librosa keeps a float64 S in float64).  Only the
    DCT and lifter tables of the plan are used; its front end is the cheapest one to build (a
    256-point transform), whose tables never see data here."""
    S = np.asarray(S)
    if S.ndim < 2:
        raise ParameterError("S must have shape (..., n_mels, n_frames)")
    n_mels, F = int(S.shape[-2]), int(S.shape[-1])
    lead = tuple(S.shape[:-2])
    if lifter < 0:
        raise ParameterError("MFCC lifter must be a non-negative number")
    n_mfcc = min(int(n_mfcc), n_mels)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # empty-filter warnings of a filterbank nobody uses
        config = make_config(sr=sr, n_mfcc=n_mfcc, dct_type=dct_type, norm=norm, lifter=lifter, n_mels=n_mels,
                             n_fft=256, hop_length=64)
    plan = get_plan(config, _device(device))
    rows = int(np.prod(lead, dtype=np.int64)) * F
    x = torch.from_numpy(np.ascontiguousarray(np.swapaxes(S, -1, -2), dtype=np.float32)).to(plan.device)
    out = plan.dct_rows(x.reshape(rows, n_mels))
    res = out.reshape(lead + (F, config.n_mfcc)).cpu().numpy()
    return np.ascontiguousarray(np.swapaxes(res, -1, -2))


def melspectrogram(*, y=None, sr=22050, device=None, validate=True, **kwargs) -> np.ndarray:
    """``librosa.feature.melspectrogram(y=...)``: ``y.shape[:-1] + (n_mels, n_frames)``."""
    y = _as_float_array(y)
    if validate:
        _valid_audio(y)
    # no DCT here: keep the (unused) n_mfcc within [1, n_mels] so small filterbanks are accepted
    kwargs.setdefault("n_mfcc", min(20, int(kwargs.get("n_mels", 128))))
    config = make_config(sr=sr, **kwargs)
    chans, lead = _channels_of(y)
    out, fo = _run_batch(chans, config, groups=None, device=device, what="melspectrogram")
    F = int(fo[1] - fo[0]) if len(chans) else 0
    res = out.reshape(len(chans), F, config.n_mels).cpu().numpy()
    return res.transpose(0, 2, 1).reshape(lead + (config.n_mels, F))


def spectrogram(*, y=None, n_fft=2048, hop_length=512, power=1.0, device=None, validate=True,
                **kwargs) -> np.ndarray:
    """``np.abs(librosa.stft(y, ...)) ** power``: ``y.shape[:-1] + (1 + n_fft//2, n_frames)``
    (/root/reference/src/preprocess.py:111-113 uses power=1)."""
    y = _as_float_array(y)
    if validate:
        _valid_audio(y)
    config = make_config(n_fft=n_fft, hop_length=hop_length, power=power, **kwargs)
    chans, lead = _channels_of(y)
    out, fo = _run_batch(chans, config, groups=None, device=device, what="spectrogram")
    F = int(fo[1] - fo[0]) if len(chans) else 0
    nb = 1 + config.n_fft // 2
    res = out.reshape(len(chans), F, nb).cpu().numpy()
    return res.transpose(0, 2, 1).reshape(lead + (nb, F))


def _to_db(S, amplitude: bool, ref, amin: float, top_db, device):
    if amin <= 0:
        raise ParameterError("amin must be strictly positive")
    if top_db is not None and top_db < 0:
        raise ParameterError("top_db must be non-negative")
    ref_is_max = callable(ref)
    if ref_is_max and ref is not np.max and getattr(ref, "__name__", "") not in ("max", "amax"):
        raise NotImplementedError("the only callable reference supported is np.max")
    dev = _device(device if device is not None else (S.device if isinstance(S, torch.Tensor) and S.is_cuda else None))
    as_tensor = isinstance(S, torch.Tensor)
    if not as_tensor and np.iscomplexobj(S):
        warnings.warn("power_to_db was called on complex input so phase information will be discarded.",
                      stacklevel=3)
        S = np.abs(S)
    x = (S if as_tensor else torch.from_numpy(np.ascontiguousarray(S, dtype=np.float32))).to(device=dev, dtype=torch.float32)
    x = x.contiguous()
    out = torch.empty_like(x)
    ws = torch.empty(256, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mfcc_to_db(x.data_ptr(), x.numel(), 1 if amplitude else 0,
                                         0.0 if ref_is_max else float(ref), 1 if ref_is_max else 0, float(amin),
                                         -1.0 if top_db is None else float(top_db), out.data_ptr(), ws.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream))
    return out if as_tensor else out.cpu().numpy()


def power_to_db(S, *, ref=1.0, amin=1e-10, top_db=80.0, device=None):
    """``librosa.power_to_db``: ``10 log10(max(amin, S)) - 10 log10(max(amin, ref))``, clamped at the
    array's maximum minus ``top_db``.  ``S``: numpy array or device tensor; same kind comes back."""
    return _to_db(S, False, ref, amin, top_db, device)


def amplitude_to_db(S, *, ref=1.0, amin=1e-5, top_db=80.0, device=None):
    """``librosa.amplitude_to_db`` as the reference applies it to ``np.abs(stft)``
    (/root/reference/src/preprocess.py:113-114): ``power_to_db(S**2, ref**2, amin**2)``."""
    return _to_db(S, True, ref, amin, top_db, device)


def delta_batch(feats: torch.Tensor, frame_offsets, *, width: int = 9, order: int = 1) -> torch.Tensor:
    """Delta features of a ragged batch: ``feats`` is the concatenated ``(total_frames, n_coef)``
    device tensor ``MfccPlan.execute`` returns, ``frame_offsets`` its host offsets.  Every
    utterance is differentiated on its own, as ``librosa.feature.delta(m, width=width, order=order,
    mode='interp')`` would on its ``(n_coef, F)`` array.  (An extension: the reference trains on
    the plain MFCCs.)"""
    fo = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous() and feats.ndim == 2
    assert int(fo[-1]) == feats.shape[0]
    out = torch.empty_like(feats)
    d_fo = torch.from_numpy(fo).to(feats.device)
    with torch.cuda.device(feats.device):
        rc = _lib.lib().mfcc_delta(feats.data_ptr(), d_fo.data_ptr(), fo.ctypes.data_as(C.POINTER(C.c_int64)),
                                   len(fo) - 1, feats.shape[1], int(width), int(order), out.data_ptr(),
                                   torch.cuda.current_stream(feats.device).cuda_stream)
    if rc == _lib.MFCC_E_INVALID:
        raise ParameterError(_lib.last_error())
    _lib.check(rc)
    return out


def delta(data, *, width: int = 9, order: int = 1, axis: int = -1, device=None):
    """``librosa.feature.delta(data, width=9, order=1, axis=-1, mode='interp')`` for one array
    (time along ``axis``); numpy in, numpy out."""
    arr = np.moveaxis(np.asarray(data, dtype=np.float32), axis, 0)
    F = arr.shape[0]
    flat = np.ascontiguousarray(arr.reshape(F, -1))
    dev = _device(device)
    out = delta_batch(torch.from_numpy(flat).to(dev), np.array([0, F], dtype=np.int64), width=width, order=order)
    return np.moveaxis(out.cpu().numpy().reshape(arr.shape), 0, axis)


def mfcc_batch(ys: Sequence, *, sr=22050, n_mfcc=20, dct_type=2, norm="ortho", lifter=0, device=None,
               validate=True, return_tensor=False, int16_if_exact=False, **kwargs) -> List:
    """Ragged batch: ``ys`` is a sequence of mono utterances, each ``(T_i,)`` or ``(1, T_i)``
    (the waveform layout of a LibriSpeech item, /root/reference/src/preprocess.py:132).

    If every utterance is a numpy int16 array the batch is taken as 16-bit PCM (sample value
    ``y / 32768``, what torchaudio decodes LibriSpeech FLAC into): half the upload, same result
    as the float32 waveform.  (``mfcc`` itself stays strict, like librosa: floating point only.)
    With ``int16_if_exact`` float waveforms are converted to that form while they are staged for
    upload whenever all of them lie on the int16 grid (bit-identical output, half the upload).
    The finiteness check of ``librosa.util.valid_audio`` runs on the device, on the uploaded batch.

    Returns one ``(n_frames_i, n_mfcc)`` float32 C-contiguous array per utterance -- exactly
    ``librosa.feature.mfcc(y=y_i, ...)[0].T`` as written to HDF5 at
    /root/reference/src/preprocess.py:135-139.  Each utterance has its own top_db maximum.
    """
    config = _split_kwargs(sr, n_mfcc, dct_type, norm, lifter, kwargs)
    chans = []
    pcm16 = len(ys) > 0 and all(isinstance(y, np.ndarray) and y.dtype == np.int16 for y in ys)
    for y in ys:
        if not pcm16:
            y = _as_float_array(y)
        if y.ndim == 2 and y.shape[0] == 1:
            y = y[0]
        if y.ndim != 1:
            raise ParameterError(f"mfcc_batch takes mono utterances, got shape {tuple(y.shape)}")
        chans.append(y)
    if not chans:
        return []
    out, fo = _run_batch(chans, config, groups=None, device=device, try_int16=int16_if_exact and not pcm16,
                         check_finite=validate and not pcm16)
    if return_tensor:
        return [out[int(a):int(b)] for a, b in zip(fo[:-1], fo[1:])]
    host = out.cpu().numpy()
    return [host[int(a):int(b)] for a, b in zip(fo[:-1], fo[1:])]
