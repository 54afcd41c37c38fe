"""Generates tests/golden/mfcc_golden.npz: small input waveforms and the outputs of the CPU
oracle (oracle/mfcc_ref.py) for them.

Provenance: these vectors come from the numpy/scipy RESTATEMENT of librosa 0.10.1, not from
librosa itself (it is not installable offline, and the reference ships no fixtures).  They pin
the oracle against accidental change and give the GPU tests fixed data; they are not an
independent witness of librosa's behaviour.

    python -m oracle.make_golden
"""
from pathlib import Path

import numpy as np

from oracle import mfcc_ref as R

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "mfcc_golden.npz"

CASES = {
    # name: (signal key, oracle kwargs)
    "ref_default": ("speech", dict(n_fft=2048, hop_length=512, n_mfcc=13)),
    "ref_default_reflect": ("speech", dict(n_fft=2048, hop_length=512, n_mfcc=13, pad_mode="reflect")),
    "short": ("short", dict(n_fft=2048, hop_length=512, n_mfcc=13)),
    "asr400": ("speech", dict(sr=16000, n_fft=400, hop_length=160, n_mels=80, n_mfcc=13)),
    "asr512": ("speech", dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13)),
    "quiet_click": ("click", dict(n_fft=2048, hop_length=512, n_mfcc=13)),
}


def signals():
    rng = np.random.default_rng(20241218)
    n = 9000
    t = np.arange(n) / 16000.0
    speech = 0.1 * rng.standard_normal(n) * np.abs(np.sin(2 * np.pi * 1.5 * t)) + 0.3 * np.sin(2 * np.pi * 220 * t)
    speech = np.round(np.clip(speech, -0.999, 0.999) * 32768) / 32768
    click = 1e-4 * rng.standard_normal(n)
    click[4321] = 0.9
    return {"speech": speech.astype(np.float32), "short": speech[:700].astype(np.float32),
            "click": click.astype(np.float32)}


def main():
    import warnings

    sig = signals()
    out = {f"y_{k}": v for k, v in sig.items()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, (key, kw) in CASES.items():
            y = sig[key]
            out[f"mfcc_{name}"] = R.mfcc(y=y, **kw)
            mkw = {k: v for k, v in kw.items() if k != "n_mfcc"}
            out[f"mel_{name}"] = R.melspectrogram(y=y, **mkw)
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes)")


if __name__ == "__main__":
    main()
