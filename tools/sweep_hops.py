#!/usr/bin/env python3
"""Extreme hop lengths (1, 7, 63, 64, 65, 3000) over every front-end kernel, centred and not, both
pad modes, two utterance lengths: each case must match the CPU oracle at 1e-2 dB.

    python tools/sweep_hops.py"""
import itertools
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_to_text_amd as S  # noqa: E402
from oracle import mfcc_ref as R  # noqa: E402
from tests.signals import speech_like  # noqa: E402

warnings.simplefilter("ignore")
bad, n = [], 0
LENGTHS = tuple(int(x) for x in os.environ.get("SWEEP_LENGTHS", "2048,1024,512,400,640,480,320,256,600").split(","))
for n_fft, hop, center, pad, T in itertools.product(LENGTHS,
                                                    (1, 7, 63, 64, 65, 3000), (True, False),
                                                    ("constant", "reflect"), (5000, 2600)):
    if not center and T < n_fft:
        continue
    y = speech_like(T, seed=T + hop)
    kw = dict(n_fft=n_fft, hop_length=hop, center=center, pad_mode=pad, n_mels=40, n_mfcc=13, sr=16000)
    n += 1
    try:
        got, want = S.mfcc(y=y, **kw), R.mfcc(y=y, **kw)
        err = float(np.abs(got - want).max()) if got.shape == want.shape else -1.0
        if err < 0 or err > 1e-2 or not np.isfinite(got).all():
            bad.append((kw, T, got.shape, want.shape, err))
    except Exception as e:  # a plan that cannot be built is a finding too
        bad.append((kw, T, type(e).__name__ + ": " + str(e)[:150]))
print(n, "cases,", len(bad), "problems")
for b in bad[:30]:
    print(b)
sys.exit(1 if bad else 0)
