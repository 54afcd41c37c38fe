#!/usr/bin/env python3
"""A/B timing of builds of libmfcc_hip.so on one box: front-end and dB+DCT kernel times (HIP
events) of each library for one configuration, every library in a fresh child process, the whole
list repeated so that drift between runs shows.

    python tools/lib_ab.py <ref|asr512|asr400|asr480|asr320|asr640|fft1024> <hours> lib1.so[:VAR=VALUE[:VAR=VALUE...]] [lib2.so ...]

A library may be followed by environment settings for its run (e.g. build/ab/x.so:MFCC_HIP_FAST=1)."""

import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
import speech_to_text_amd as S
from bench import make_batch
from tools.bench_configs import run
which = %r
cfg = {"ref": S.MfccConfig.reference_default(),
       "asr512": S.make_config(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13),
       "asr400": S.make_config(sr=16000, n_fft=400, hop_length=160, n_mels=80, n_mfcc=13),
       "fft1024": S.make_config(sr=16000, n_fft=1024, hop_length=256, n_mels=80, n_mfcc=13),
       "asr480": S.make_config(sr=16000, n_fft=480, hop_length=160, n_mels=80, n_mfcc=13),
       "asr640": S.make_config(sr=16000, n_fft=640, hop_length=160, n_mels=80, n_mfcc=13),
       "asr800": S.make_config(sr=16000, n_fft=800, hop_length=160, n_mels=80, n_mfcc=13),
       "asr960": S.make_config(sr=16000, n_fft=960, hop_length=160, n_mels=80, n_mfcc=13),
       "asr320": S.make_config(sr=16000, n_fft=320, hop_length=160, n_mels=80, n_mfcc=13)}[which]
dev = torch.device('cuda', 0)
pcm, so = make_batch(%f, 1234, dev)
r = run(cfg, pcm, so, steps=20, warmup=3)
print(json.dumps({'front_ms': round(r['front_kernel_ms'], 4), 'db_dct_ms': round(r['db_dct_kernel_ms'], 4),
                  'audio_h_per_s': round(r['audio_hours_per_s'], 1)}))
"""


def main():
    which, hours, libs = sys.argv[1], float(sys.argv[2]), sys.argv[3:]
    for rep in range(2):
        for spec in libs:
            lib, *sets = spec.split(":")
            env = dict(os.environ, MFCC_HIP_LIBRARY=str(Path(lib).resolve()))
            env.update(dict(kv.split("=", 1) for kv in sets))
            r = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), which, hours)], env=env,
                               capture_output=True, text=True, timeout=300)
            print(rep, spec, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-600:],
                  flush=True)


if __name__ == "__main__":
    main()
