#!/usr/bin/env python3
"""Dense-DFT kernel time for each tiles-per-workgroup setting (MFCC_HIP_GENERIC_NT), one child
process per setting so every plan is built fresh.  python tools/nt_sweep.py [--hours H]"""
import argparse

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
dev = torch.device('cuda', 0)
pcm, so = make_batch(%f, 1234, dev)
out = {}
for name, cfg in (('nfft400', S.make_config(sr=16000, n_fft=400, hop_length=160, n_mels=80, n_mfcc=13)),
                  ('nfft512', S.make_config(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13))):
    r = run(cfg, pcm, so, steps=5, warmup=2)
    out[name] = {'front_kernel_ms': r['front_kernel_ms'], 'tflops': r['dense_dft_mfma_tflops']}
print(json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=float, default=10.0)
    args = ap.parse_args()
    for nt in (1, 2, 3, 4):
        env = dict(os.environ, MFCC_HIP_GENERIC_NT=str(nt))
        r = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), args.hours)], env=env, capture_output=True,
                           text=True, timeout=200)
        print(nt, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
