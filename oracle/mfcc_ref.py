"""CPU oracle: an op-for-op numpy/scipy restatement of ``librosa.feature.mfcc`` (librosa 0.10.1).

TEST INFRASTRUCTURE ONLY.  Nothing under ``speech_to_text_amd/`` imports this module; it is
used by ``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` as the checker / the timed CPU baseline, never as the product path.

Parity status: **parity unpinned by the reference's own tests** -- the reference
(jvargeaux/speech_to_text) ships no tests, goldens or fixtures, and the arithmetic lives in the
third-party package librosa==0.10.1 (``/root/reference/requirements.txt:15``) over
numpy==1.24.4 (``:23``) and scipy==1.11.2 (``:33``), none of which is vendored and librosa is
not installable offline.  This file restates the published librosa 0.10.1 algorithm for the one
call the reference makes,

    librosa.feature.mfcc(y=samples.numpy(), n_fft=2048, hop_length=512, n_mfcc=13)
        -- /root/reference/src/preprocess.py:134 (and :123)

and is pinned instead by (a) librosa's documented known-answer values for its mel-scale
helpers and filterbank (tests/test_oracle_known_answers.py), (b) independent implementations
of each stage (explicit DFT / cosine matrices) and (c) analytic signals (silence, bin-centred
sinusoid, impulse).

Each function cites the upstream librosa 0.10.1 function it restates; the reference call
sites are ``/root/reference/src/preprocess.py:111`` (stft), ``:114`` (amplitude_to_db) and
``:123``/``:134`` (mfcc).
"""
from __future__ import annotations

import warnings

import numpy as np
import scipy.fftpack
import scipy.signal

# librosa.util.utils.MAX_MEM_BLOCK: the STFT is evaluated in column blocks of at most this
# many bytes.  It does not change results, only the blocking of the rFFT calls.
MAX_MEM_BLOCK = 2 ** 8 * 2 ** 10


class ParameterError(ValueError):
    """librosa.util.exceptions.ParameterError."""


# --------------------------------------------------------------------------------------------
# librosa.core.convert
# --------------------------------------------------------------------------------------------
def hz_to_mel(frequencies, *, htk: bool = False):
    """librosa.core.convert.hz_to_mel (Slaney by default, HTK optional)."""
    frequencies = np.asanyarray(frequencies)
    if htk:
        return 2595.0 * np.log10(1.0 + frequencies / 700.0)
    f_min = 0.0
    f_sp = 200.0 / 3
    mels = (frequencies - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if frequencies.ndim:
        log_t = frequencies >= min_log_hz
        mels[log_t] = min_log_mel + np.log(frequencies[log_t] / min_log_hz) / logstep
    elif frequencies >= min_log_hz:
        mels = min_log_mel + np.log(frequencies / min_log_hz) / logstep
    return mels


def mel_to_hz(mels, *, htk: bool = False):
    """librosa.core.convert.mel_to_hz."""
    mels = np.asanyarray(mels)
    if htk:
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_min = 0.0
    f_sp = 200.0 / 3
    freqs = f_min + f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = mels >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_frequencies(n_mels: int = 128, *, fmin: float = 0.0, fmax: float = 11025.0, htk: bool = False):
    """librosa.core.convert.mel_frequencies."""
    min_mel = hz_to_mel(fmin, htk=htk)
    max_mel = hz_to_mel(fmax, htk=htk)
    mels = np.linspace(min_mel, max_mel, n_mels)
    return mel_to_hz(mels, htk=htk)


def fft_frequencies(*, sr: float = 22050, n_fft: int = 2048):
    """librosa.core.convert.fft_frequencies."""
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)


# --------------------------------------------------------------------------------------------
# librosa.filters
# --------------------------------------------------------------------------------------------
def mel(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney", dtype=np.float32):
    """librosa.filters.mel: triangular filterbank, computed in float64, stored as ``dtype``."""
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=dtype)
    fftfreqs = fft_frequencies(sr=sr, n_fft=n_fft)
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax, htk=htk)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if isinstance(norm, str):
        if norm == "slaney":
            enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
            weights *= enorm[:, np.newaxis]
        else:
            raise ParameterError(f"Unsupported norm={norm}")
    elif norm is not None:
        # librosa.util.normalize(weights, norm=norm, axis=-1)
        if norm == np.inf:
            length = np.max(np.abs(weights), axis=-1, keepdims=True)
        elif norm == 1 or norm == 2:
            length = np.sum(np.abs(weights) ** norm, axis=-1, keepdims=True) ** (1.0 / norm)
        else:
            raise ParameterError(f"Unsupported norm={norm}")
        tiny = np.finfo(weights.dtype).tiny
        length[length < tiny] = 1.0
        weights = weights / length
    if not np.all((mel_f[:-2] == 0) | (weights.max(axis=1) > 0)):
        warnings.warn(
            "Empty filters detected in mel frequency basis. Some channels will produce empty "
            "responses. Try increasing your sampling rate (and fmax) or reducing n_mels.",
            stacklevel=2,
        )
    return weights


def get_window(window, Nx: int, *, fftbins: bool = True):
    """librosa.filters.get_window: scipy window spec, callable, or an explicit array."""
    if callable(window):
        return np.asarray(window(Nx))
    if isinstance(window, (str, tuple)) or np.isscalar(window):
        return scipy.signal.get_window(window, Nx, fftbins=fftbins)
    if isinstance(window, (np.ndarray, list)):
        if len(window) == Nx:
            return np.asarray(window)
        raise ParameterError(f"Window size mismatch: {len(window)} != {Nx}")
    raise ParameterError(f"Invalid window specification: {window!r}")


def pad_center(data, *, size: int, axis: int = -1):
    """librosa.util.pad_center (constant zero padding, left pad = (size - n) // 2)."""
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, int(size - n - lpad))
    if lpad < 0:
        raise ParameterError(f"Target size ({size}) must be at least input size ({n})")
    return np.pad(data, lengths, mode="constant")


# --------------------------------------------------------------------------------------------
# librosa.util.valid_audio / librosa.core.spectrum
# --------------------------------------------------------------------------------------------
def valid_audio(y) -> bool:
    """librosa.util.valid_audio: floating ndarray, non-empty last axis, all finite."""
    if not isinstance(y, np.ndarray):
        raise ParameterError("Audio data must be of type numpy.ndarray")
    if not np.issubdtype(y.dtype, np.floating):
        raise ParameterError("Audio data must be floating-point")
    if y.ndim == 0:
        raise ParameterError(f"Audio data must be at least one-dimensional, given y.shape={y.shape}")
    if not np.isfinite(y).all():
        raise ParameterError("Audio buffer is not finite everywhere")
    return True


def _dtype_r2c(d):
    return {np.dtype(np.float32): np.complex64, np.dtype(np.float64): np.complex128}.get(np.dtype(d), np.complex64)


def n_frames(T: int, n_fft: int, hop_length: int, center: bool = True) -> int:
    """Number of STFT columns librosa produces for a signal of T samples."""
    if center:
        T = T + 2 * (n_fft // 2)
    if T < n_fft:
        raise ParameterError(f"n_fft={n_fft} is too large for input signal of length={T}")
    return 1 + (T - n_fft) // hop_length


def stft(y, *, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True,
         dtype=None, pad_mode="constant"):
    """librosa.core.spectrum.stft (0.10.1): float64 window x input frames -> double-precision
    ``numpy.fft.rfft`` along the frame axis -> stored as complex64 for float32 input.

    Centred analysis follows 0.10's head / tail split literally: when the frames that touch the
    left padding and those that touch the right padding are different frames, librosa pads two
    SLICES of the signal (``y[:...]`` on the left only, ``y[tail_k * hop - n_fft // 2:]`` on the
    right only) instead of the whole signal.  For ``pad_mode='constant'`` that is the same thing.
    For ``'reflect'`` it is the same thing as long as the tail slice holds more than ``n_fft // 2``
    samples.  A tail frame starts ``s`` samples after a multiple of the hop, and its slice holds
    ``n_fft // 2 + (len(y) mod hop_length)`` samples: with ``hop_length >= n_fft / 2`` and a signal length
    that is a multiple of the hop the slice is one sample short, numpy turns round inside it, and
    the very last sample of the last frame is ``y[lo + 1]`` where a reflection of the whole signal
    has ``y[lo - 1]`` (``lo`` = the slice's start).
    """
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = int(win_length // 4)
    elif not (isinstance(hop_length, (int, np.integer)) and hop_length > 0):
        raise ParameterError(f"hop_length={hop_length} must be a positive integer")
    valid_audio(y)
    if pad_mode in ("wrap", "maximum", "mean", "median", "minimum"):
        # librosa 0.10.1 rejects exactly these five numpy.pad modes in stft
        raise ParameterError(f"pad_mode='{pad_mode}' is not supported by librosa.stft")
    if pad_mode not in ("constant", "reflect", "edge", "symmetric", "linear_ramp", "empty"):
        raise ParameterError(f"pad_mode='{pad_mode}' is not a numpy.pad mode")

    fft_window = get_window(window, win_length, fftbins=True)      # float64
    fft_window = pad_center(fft_window, size=n_fft)
    fft_window = fft_window.reshape((-1, 1))                         # broadcast along frames

    def frame(x):
        """librosa.util.frame(x, frame_length=n_fft, hop_length=hop_length) -> (..., n_fft, F)."""
        nf = 1 + (x.shape[-1] - n_fft) // hop_length if x.shape[-1] >= n_fft else 0
        idx = np.arange(n_fft)[:, None] + hop_length * np.arange(nf)[None, :]
        return x[..., idx]

    y_frames_pre = y_frames_post = None
    start = extra = 0
    if center:
        if n_fft > y.shape[-1]:
            warnings.warn(f"n_fft={n_fft} is too large for input signal of length={y.shape[-1]}",
                          stacklevel=2)
        padding = [(0, 0)] * y.ndim
        start_k = int(np.ceil(n_fft // 2 / hop_length))                     # frames that touch the left padding
        tail_k = (y.shape[-1] + n_fft // 2 - n_fft) // hop_length + 1      # first frame that touches the right padding
        if tail_k <= start_k:
            # head and tail overlap: pad the whole signal
            padding[-1] = (n_fft // 2, n_fft // 2)
            y = np.pad(y, padding, mode=pad_mode)
        else:
            start = start_k * hop_length - n_fft // 2           # the middle of the signal starts here
            padding[-1] = (n_fft // 2, 0)
            y_pre = np.pad(y[..., : (start_k - 1) * hop_length - n_fft // 2 + n_fft + 1], padding, mode=pad_mode)
            y_frames_pre = frame(y_pre)[..., :start_k]
            extra = y_frames_pre.shape[-1]
            if tail_k * hop_length - n_fft // 2 + n_fft <= y.shape[-1] + n_fft // 2:
                padding[-1] = (0, n_fft // 2)
                y_post = np.pad(y[..., tail_k * hop_length - n_fft // 2:], padding, mode=pad_mode)
                y_frames_post = frame(y_post)
                extra += y_frames_post.shape[-1]
            else:
                y_frames_post = y_frames_pre[..., :0]
    elif n_fft > y.shape[-1]:
        raise ParameterError(f"n_fft={n_fft} is too large for uncentered analysis of input "
                             f"signal of length={y.shape[-1]}")

    y_frames = frame(y[..., start:])                                # (..., n_fft, middle frames), y's dtype
    if dtype is None:
        dtype = _dtype_r2c(y.dtype)
    shape = list(y_frames.shape)
    shape[-2] = 1 + n_fft // 2
    shape[-1] += extra
    stft_matrix = np.zeros(shape, dtype=dtype, order="F")

    off_start = 0
    if center and extra > 0:
        off_start = y_frames_pre.shape[-1]
        stft_matrix[..., :off_start] = np.fft.rfft(fft_window * y_frames_pre, axis=-2)
        off_end = y_frames_post.shape[-1]
        if off_end > 0:
            stft_matrix[..., -off_end:] = np.fft.rfft(fft_window * y_frames_post, axis=-2)

    n_columns = max(1, int(MAX_MEM_BLOCK // (np.prod(y_frames.shape[:-1]) * y_frames.itemsize)))
    for bl_s in range(0, y_frames.shape[-1], n_columns):
        bl_t = min(bl_s + n_columns, y_frames.shape[-1])
        # float64 window * float32 frames -> float64 product -> double-precision rFFT
        stft_matrix[..., bl_s + off_start:bl_t + off_start] = np.fft.rfft(fft_window * y_frames[..., bl_s:bl_t], axis=-2)
    return stft_matrix


def _spectrogram(*, y, n_fft=2048, hop_length=512, power=1.0, win_length=None, window="hann",
                 center=True, pad_mode="constant"):
    """librosa.core.spectrum._spectrogram for the y-given case: ``np.abs(stft(...)) ** power``."""
    S = np.abs(stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length, center=center,
                    window=window, pad_mode=pad_mode)) ** power
    return S, n_fft


def power_to_db(S, *, ref=1.0, amin=1e-10, top_db=80.0):
    """librosa.core.spectrum.power_to_db.  ``log_spec.max()`` has no axis: it is taken over the
    whole array handed in (one utterance at /root/reference/src/preprocess.py:134)."""
    S = np.asarray(S)
    if amin <= 0:
        raise ParameterError("amin must be strictly positive")
    if np.issubdtype(S.dtype, np.complexfloating):
        warnings.warn("power_to_db was called on complex input so phase information will be "
                      "discarded.", stacklevel=2)
        magnitude = np.abs(S)
    else:
        magnitude = S
    ref_value = ref(magnitude) if callable(ref) else np.abs(ref)
    log_spec = 10.0 * np.log10(np.maximum(amin, magnitude))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        if top_db < 0:
            raise ParameterError("top_db must be non-negative")
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def amplitude_to_db(S, *, ref=1.0, amin=1e-5, top_db=80.0):
    """librosa.core.spectrum.amplitude_to_db == power_to_db(S**2, ref**2, amin**2)
    (call site /root/reference/src/preprocess.py:114)."""
    S = np.asarray(S)
    magnitude = np.abs(S)
    ref_value = ref(magnitude) if callable(ref) else np.abs(ref)
    power = np.square(magnitude, out=magnitude)
    return power_to_db(power, ref=ref_value ** 2, amin=amin ** 2, top_db=top_db)


# --------------------------------------------------------------------------------------------
# librosa.feature.spectral
# --------------------------------------------------------------------------------------------
def melspectrogram(*, y=None, sr=22050, S=None, n_fft=2048, hop_length=512, win_length=None,
                   window="hann", center=True, pad_mode="constant", power=2.0, **kwargs):
    """librosa.feature.melspectrogram: ``einsum('...ft,mf->...mt', |stft|**power, mel_basis)``."""
    if S is None:
        S, n_fft = _spectrogram(y=y, n_fft=n_fft, hop_length=hop_length, power=power,
                                win_length=win_length, window=window, center=center,
                                pad_mode=pad_mode)
    else:
        S = np.asarray(S)
        n_fft = 2 * (S.shape[-2] - 1)
    mel_basis = mel(sr=sr, n_fft=n_fft, **kwargs)
    return np.einsum("...ft,mf->...mt", S, mel_basis, optimize=True)


def mfcc(*, y=None, sr=22050, S=None, n_mfcc=20, dct_type=2, norm="ortho", lifter=0, **kwargs):
    """librosa.feature.mfcc.  ``**kwargs`` go to melspectrogram only; power_to_db always runs
    with its defaults (ref=1.0, amin=1e-10, top_db=80.0)."""
    if S is None:
        S = power_to_db(melspectrogram(y=y, sr=sr, **kwargs))
    M = scipy.fftpack.dct(S, axis=-2, type=dct_type, norm=norm)[..., :n_mfcc, :]
    if lifter > 0:
        LI = np.sin(np.pi * np.arange(1, 1 + n_mfcc, dtype=M.dtype) / lifter)
        LI = LI.reshape([-1 if ax == M.ndim - 2 else 1 for ax in range(M.ndim)])
        M *= 1 + (lifter / 2) * LI
        return M
    elif lifter == 0:
        return M
    raise ParameterError(f"MFCC lifter={lifter} must be a non-negative number")


# --------------------------------------------------------------------------------------------
# The reference's use of it
# --------------------------------------------------------------------------------------------
def process_audio_features(samples, *, n_fft=2048, hop_length=512, n_mfcc=13, **kwargs):
    """What /root/reference/src/preprocess.py:134-136 hands to h5py: ``mfcc(y=(1,T))[0].T``,
    made C-contiguous -- a float32 array of shape (n_frames, n_mfcc)."""
    samples = np.asarray(samples)
    if samples.ndim == 1:
        samples = samples[None, :]
    m = mfcc(y=samples, n_fft=n_fft, hop_length=hop_length, n_mfcc=n_mfcc, **kwargs)
    return np.ascontiguousarray(m[0].T)


# --------------------------------------------------------------------------------------------
# librosa.feature.delta (not called by the reference; checks the delta extension)
# --------------------------------------------------------------------------------------------
def delta(data, *, width=9, order=1, axis=-1, mode="interp", **kwargs):
    """librosa.feature.delta: scipy.signal.savgol_filter(data, width, deriv=order, polyorder=order,
    axis=axis, mode=mode)."""
    import scipy.signal

    data = np.atleast_1d(data)
    if mode == "interp" and width > data.shape[axis]:
        raise ParameterError(f"when mode='interp', width={width} cannot exceed data.shape[axis]={data.shape[axis]}")
    if width < 3 or np.mod(width, 2) != 1:
        raise ParameterError("width must be an odd integer >= 3")
    if order <= 0 or not isinstance(order, (int, np.integer)):
        raise ParameterError("order must be a positive integer")
    kwargs.pop("deriv", None)
    kwargs.setdefault("polyorder", order)
    return scipy.signal.savgol_filter(data, width, deriv=order, axis=axis, mode=mode, **kwargs)
