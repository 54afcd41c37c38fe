#!/usr/bin/env python3
"""Sweep of filterbank shapes over the front-end kernels: every combination of n_fft (2048, 1024,
512, 400, 640, 256), n_mels (1 .. 256), sr (8 .. 44.1 kHz), htk, fmax and hop must build a plan and
match the CPU oracle on one second of speech-like audio at 1e-2 dB (1 680 shapes, about 15 s).

    python tools/sweep_shapes.py                     (SWEEP_LENGTHS=128,160,... for other transform lengths)"""
import os, sys, warnings, itertools
import numpy as np
sys.path.insert(0, os.getcwd())
import speech_to_text_amd as S
from oracle import mfcc_ref as R
from tests.signals import speech_like
warnings.simplefilter("ignore")
LENGTHS = tuple(int(x) for x in os.environ.get("SWEEP_LENGTHS", "2048,1024,512,400,640,256").split(","))
y = speech_like(16000, seed=3)
bad = []
n = 0
for n_fft, n_mels, sr, htk, fmax, hop in itertools.product(
        LENGTHS, (1, 2, 3, 5, 8, 13, 23, 40, 64, 80, 96, 128, 160, 200, 256),
        (8000, 16000, 22050, 44100), (False, True), (None, 0.4), (None, 512)):
    kw = dict(n_fft=n_fft, hop_length=hop or n_fft // 4, n_mels=n_mels, sr=sr, htk=htk, n_mfcc=min(13, n_mels))
    if fmax: kw["fmax"] = fmax * sr
    if hop == 512 and n_fft != 2048: continue
    n += 1
    try:
        got = S.mfcc(y=y, **kw)
        want = R.mfcc(y=y, **kw)
        err = float(np.abs(got - want).max())
        if not np.isfinite(got).all() or err > 1e-2: bad.append((kw, "err %.4f" % err))
    except Exception as e:
        bad.append((kw, type(e).__name__ + ": " + str(e)[:120]))
print(n, "shapes,", len(bad), "problems")
for b in bad[:40]: print(b)
