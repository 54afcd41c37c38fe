"""Deterministic synthetic audio shared by the tests, the golden generator and bench.py."""
import numpy as np

SR = 16000


def speech_like(n_samples: int, seed: int, int16_grid: bool = False) -> np.ndarray:
    """0.1 N(0,1) |sin(2 pi 1.5 t)| + 0.3 sin(2 pi 220 t) at 16 kHz, float32 in (-1, 1)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / SR
    y = 0.1 * rng.standard_normal(n_samples) * np.abs(np.sin(2 * np.pi * 1.5 * t)) + 0.3 * np.sin(2 * np.pi * 220.0 * t)
    y = np.clip(y, -0.999, 0.999)
    if int16_grid:
        y = np.round(y * 32768.0) / 32768.0
    return y.astype(np.float32)


def ragged_lengths(n_utts: int, lo_s: float, hi_s: float, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (rng.uniform(lo_s, hi_s, size=n_utts) * SR).astype(np.int64)
