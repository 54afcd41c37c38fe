"""Frozen description of one MFCC configuration, with librosa's keyword names and defaults.

The reference fixes ``n_fft=2048, hop_length=512, n_mfcc=13`` as instance attributes
(/root/reference/src/preprocess.py:60-62) and leaves everything else at the librosa 0.10.1
defaults of ``feature.mfcc -> feature.melspectrogram -> stft -> filters.mel -> power_to_db``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

from . import _lib


class ParameterError(ValueError):
    """Same role as ``librosa.util.exceptions.ParameterError``."""


_PAD_MODES = {"constant": _lib.PAD_CONSTANT, "reflect": _lib.PAD_REFLECT}


@dataclass(frozen=True)
class MfccConfig:
    # librosa.feature.mfcc
    sr: float = 22050
    n_mfcc: int = 20
    dct_type: int = 2
    norm: Optional[str] = "ortho"
    lifter: float = 0
    # librosa.feature.melspectrogram / stft
    n_fft: int = 2048
    hop_length: int = 512
    win_length: Optional[int] = None
    window: object = "hann"          # 'hann', or an explicit array of win_length samples
    center: bool = True
    pad_mode: str = "constant"
    power: float = 2.0
    # librosa.filters.mel
    n_mels: int = 128
    fmin: float = 0.0
    fmax: Optional[float] = None
    htk: bool = False
    mel_norm: Optional[str] = "slaney"
    # librosa.power_to_db (mfcc always calls it with these defaults)
    ref: float = 1.0
    amin: float = 1e-10
    top_db: Optional[float] = 80.0
    # cache key helper for explicit windows
    _window_key: Tuple = field(default=(), repr=False, compare=True)

    @staticmethod
    def reference_default() -> "MfccConfig":
        """The configuration of /root/reference/src/preprocess.py:134."""
        return MfccConfig(n_fft=2048, hop_length=512, n_mfcc=13)

    # ------------------------------------------------------------------------------------
    def validate(self) -> None:
        if not (isinstance(self.hop_length, (int, np.integer)) and self.hop_length > 0):
            raise ParameterError(f"hop_length={self.hop_length} must be a positive integer")
        if self.pad_mode not in _PAD_MODES:
            raise ParameterError(f"pad_mode='{self.pad_mode}' is not supported (constant, reflect)")
        if self.amin <= 0:
            raise ParameterError("amin must be strictly positive")
        if self.top_db is not None and self.top_db < 0:
            raise ParameterError("top_db must be non-negative")
        if self.lifter < 0:
            raise ParameterError(f"MFCC lifter={self.lifter} must be a non-negative number")
        if self.norm not in (None, "ortho"):
            raise ParameterError(f"Unsupported DCT norm={self.norm}")
        if self.mel_norm not in (None, "slaney"):
            raise ParameterError(f"Unsupported norm={self.mel_norm}")

    def window_array(self) -> Optional[np.ndarray]:
        """float64 window of win_length samples for anything but the built-in periodic Hann."""
        w = self.window
        if isinstance(w, str) and w in ("hann", "hanning"):
            return None
        win_length = self.win_length or self.n_fft
        if callable(w):
            arr = np.asarray(w(win_length), dtype=np.float64)
        elif isinstance(w, (str, tuple)) or np.isscalar(w):
            import scipy.signal

            arr = scipy.signal.get_window(w, win_length, fftbins=True).astype(np.float64)
        else:
            arr = np.asarray(w, dtype=np.float64)
        if arr.shape != (win_length,):
            raise ParameterError(f"Window size mismatch: {arr.shape[0] if arr.ndim else 0} != {win_length}")
        return np.ascontiguousarray(arr)

    def to_c(self):
        """Returns (MfccConfigC, keepalive) -- keepalive holds the arrays the struct points at."""
        self.validate()
        c = _lib.MfccConfigC()
        _lib.lib().mfcc_config_default(C.byref(c))
        c.n_fft = int(self.n_fft)
        c.hop_length = int(self.hop_length)
        c.win_length = int(self.win_length or 0)
        c.center = 1 if self.center else 0
        c.pad_mode = _PAD_MODES[self.pad_mode]
        c.power = float(self.power)
        c.sr = float(self.sr)
        c.n_mels = int(self.n_mels)
        c.fmin = float(self.fmin)
        c.fmax = float(self.fmax) if self.fmax is not None else 0.0
        c.htk = 1 if self.htk else 0
        c.mel_norm = 1 if self.mel_norm == "slaney" else 0
        c.ref = float(abs(self.ref))
        c.amin = float(self.amin)
        c.top_db = float(self.top_db) if self.top_db is not None else -1.0
        c.n_mfcc = int(self.n_mfcc)
        c.dct_type = int(self.dct_type)
        c.dct_norm = 1 if self.norm == "ortho" else 0
        c.lifter = float(self.lifter)
        keep = []
        win = self.window_array()
        if win is not None:
            keep.append(win)
            c.window = win.ctypes.data_as(C.POINTER(C.c_double))
        return c, keep


def make_config(**kw) -> MfccConfig:
    """Build an MfccConfig from librosa-style keywords; explicit window arrays become hashable."""
    w = kw.get("window", "hann")
    if not (isinstance(w, (str, tuple)) or np.isscalar(w) or callable(w)):
        arr = np.ascontiguousarray(np.asarray(w, dtype=np.float64))
        kw["window"] = _HashableArray(arr)
        kw["_window_key"] = (arr.shape, arr.tobytes())
    return MfccConfig(**kw)


class _HashableArray:
    """Wraps an ndarray so a frozen dataclass holding it stays hashable (keyed by _window_key)."""

    def __init__(self, arr):
        self.arr = arr

    def __array__(self, dtype=None, copy=None):
        return self.arr if dtype is None else self.arr.astype(dtype)

    def __eq__(self, other):
        return isinstance(other, _HashableArray) and np.array_equal(self.arr, other.arr)

    def __hash__(self):
        return hash(self.arr.tobytes())

    def __len__(self):
        return len(self.arr)
