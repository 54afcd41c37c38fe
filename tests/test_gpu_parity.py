"""GPU parity: libmfcc_hip (through the C ABI) against the CPU oracle on the same inputs.

Tolerances are the north-star's: <= 1e-4 relative on mel power (with the absolute floor an
fp32 FFT needs, stated below) and <= 1e-2 dB absolute on log-mel / MFCC.
"""
import numpy as np
import pytest

from oracle import mfcc_ref as R
from tests.signals import speech_like

pytestmark = pytest.mark.gpu

DB_TOL = 1e-2          # dB, absolute, on MFCC
MEL_RTOL = 1e-4        # relative, on mel power
# An fp32 FFT's error is relative to the frame's energy, not to each bin: bands far below the
# frame peak get an absolute floor of MEL_FLOOR * (largest mel power in the utterance).
MEL_FLOOR = 2e-7


def check_mel(got, want):
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64)
    tol = MEL_RTOL * np.abs(want) + MEL_FLOOR * np.abs(want).max()
    err = np.abs(got - want)
    assert (err <= tol).all(), f"mel power: max err/tol = {(err / np.maximum(tol, 1e-300)).max():.3g}"


def test_reference_default_one_utterance(gpu):
    import speech_to_text_amd as S

    y = speech_like(166960, seed=2)
    want = R.process_audio_features(y)                      # (327, 13)
    got = S.mfcc(y=y[None, :], n_fft=2048, hop_length=512, n_mfcc=13)
    assert got.shape == (1, 13, 327) and got.dtype == np.float32
    np.testing.assert_allclose(got[0].T, want, rtol=0, atol=DB_TOL)


def test_melspectrogram_parity(gpu):
    import speech_to_text_amd as S

    y = speech_like(50000, seed=3)
    want = R.melspectrogram(y=y, n_fft=2048, hop_length=512)
    got = S.melspectrogram(y=y, n_fft=2048, hop_length=512)
    assert got.shape == want.shape
    check_mel(got, want)


def test_spectrogram_parity(gpu):
    import speech_to_text_amd as S

    y = speech_like(30000, seed=4)
    want = np.abs(R.stft(y, n_fft=2048, hop_length=512)) ** 2
    got = S.spectrogram(y=y, n_fft=2048, hop_length=512, power=2.0)
    assert got.shape == want.shape
    tol = 1e-4 * want + 2e-7 * want.max(axis=0, keepdims=True) * 10
    assert (np.abs(got - want) <= tol).all()
