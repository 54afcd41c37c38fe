"""Committed golden vectors (tests/golden/mfcc_golden.npz, made by oracle/make_golden.py from
the librosa-0.10.1 restatement; "parity unpinned" by the reference, which ships no fixtures).
The CPU test pins the oracle; the GPU test checks the HIP path against the same vectors."""
import warnings
from pathlib import Path

import numpy as np
import pytest

from oracle import mfcc_ref as R
from oracle.make_golden import CASES

G = np.load(Path(__file__).resolve().parent / "golden" / "mfcc_golden.npz")


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    key, kw = CASES[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = R.mfcc(y=G[f"y_{key}"], **kw)
    want = G[f"mfcc_{name}"]
    assert got.shape == want.shape and got.dtype == want.dtype == np.float32
    # same code, possibly another numpy/scipy build: allow last-ulp differences of libm / FFT
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_golden(gpu, name):
    import speech_to_text_amd as S

    key, kw = CASES[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = S.mfcc(y=G[f"y_{key}"], **kw)
        mkw = {k: v for k, v in kw.items() if k != "n_mfcc"}
        mel = S.melspectrogram(y=G[f"y_{key}"], **mkw)
    np.testing.assert_allclose(got, G[f"mfcc_{name}"], rtol=0, atol=1e-2)          # dB
    want = G[f"mel_{name}"].astype(np.float64)
    tol = 1e-4 * want + 2e-7 * want.max()
    assert (np.abs(mel - want) <= tol).all()
