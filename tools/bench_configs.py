#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations (bench.py reports configs[1] only).

    python tools/bench_configs.py [--hours H] [--out file.json]

config 3  1000 h: the 10 h resident batch looped 100 times, HIP-event kernel times (no new
          audio is generated per pass; a 1000 h corpus does not fit one GPU, each rank would
          stream its share through resident batches of this size)
config 4  ASR front end: sr 16000, hop 160, 80 mels, 13 MFCC with n_fft = 400 (25 x 16 in-register
          DFT kernel) and with n_fft = 512 / win_length = 400 (the 400-sample window zero-padded
          to a 512-point FFT; in-register FFT kernel), + global CMVN (accumulate, apply)
extra     n_fft = 480 (30 ms) and 320 (20 ms): the 30 x 16 and 20 x 16 in-register kernels; n_fft = 640,
          which no in-register kernel covers: the dense-DFT (MFMA) kernel
config 5  long utterances U(30, 60) s, one of them silent but for a click

Each entry: audio-hours/s from wall time over whole steps (synchronised), per-kernel HIP-event
times, algorithmic bytes and, for the dense-DFT kernel, the fp32 MFMA rate.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import speech_to_text_amd as S  # noqa: E402
from bench import SR, make_batch  # noqa: E402
from speech_to_text_amd import cmvn  # noqa: E402


def run(config, pcm, so, steps, warmup, with_cmvn=False):
    dev = pcm.device
    plan = S.MfccPlan(config, dev)
    plan.set_profiling(True)
    fo = plan.frame_offsets(so)
    F = int(fo[-1])
    d_so = torch.from_numpy(so).to(dev)
    d_fo = torch.from_numpy(fo).to(dev)
    out = torch.empty((F, config.n_mfcc), dtype=torch.float32, device=dev)
    ws = plan.workspace(F, len(so) - 1)
    stats = cmvn.new_stats(config.n_mfcc, dev) if with_cmvn else None

    def step():
        plan.execute(pcm, d_so, d_fo, F, out=out, workspace=ws)
        if with_cmvn:
            stats.zero_()
            cmvn.accumulate(out, stats)
            cmvn.apply(out, stats)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    front, dct = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        a, b = plan.last_timings()
        front.append(a)
        dct.append(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    hours = float(so[-1]) / SR / 3600.0
    n_bins = 1 + config.n_fft // 2
    res = {
        "audio_hours_per_step": hours, "utterances": int(len(so) - 1), "frames": F,
        "ms_per_step": dt * 1e3, "audio_hours_per_s": hours / dt,
        "front_kernel_ms": float(np.mean(front)), "db_dct_kernel_ms": float(np.mean(dct)),
        "algorithmic_bytes_end_to_end": 4.0 * float(so[-1]) + 4.0 * config.n_mfcc * F,
        "algorithmic_bytes_front": 4.0 * float(so[-1]) + 4.0 * config.n_mels * F,
    }
    res["front_GBps"] = res["algorithmic_bytes_front"] / (res["front_kernel_ms"] * 1e-3) / 1e9
    res["front_kernel"] = plan.front_kernel
    if plan.front_kernel == "dense_dft":
        kpad = (config.n_fft + 15) // 16 * 16
        coltiles = (2 * n_bins + 15) // 16
        flops = 2.0 * 16 * kpad * 16 * coltiles * ((F + 15) // 16)     # MFMA work incl. padding
        res["dense_dft_mfma_tflops"] = flops / (res["front_kernel_ms"] * 1e-3) / 1e12
        res["fp32_mfma_peak_tflops"] = 157.3
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=float, default=10.0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    results = {}

    pcm, so = make_batch(args.hours, 1234, dev)
    ref = S.MfccConfig.reference_default()
    r = run(ref, pcm, so, steps=100, warmup=3)
    r["note"] = (f"{args.hours:g} h resident batch looped 100x = {100 * args.hours:g} audio-hours; "
                 "kernel times are HIP events per pass")
    results["config3_1000h_loop"] = r

    asr400 = S.make_config(sr=16000, n_fft=400, hop_length=160, n_mels=80, n_mfcc=13)
    results["config4_nfft400_cmvn"] = run(asr400, pcm, so, steps=10, warmup=2, with_cmvn=True)
    asr512 = S.make_config(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13)
    results["config4_nfft512_win400_cmvn"] = run(asr512, pcm, so, steps=10, warmup=2, with_cmvn=True)
    asr480 = S.make_config(sr=16000, n_fft=480, hop_length=160, n_mels=80, n_mfcc=13)
    results["extra_nfft480"] = run(asr480, pcm, so, steps=10, warmup=2)
    asr320 = S.make_config(sr=16000, n_fft=320, hop_length=160, n_mels=80, n_mfcc=13)
    results["extra_nfft320"] = run(asr320, pcm, so, steps=10, warmup=2)
    asr640 = S.make_config(sr=16000, n_fft=640, hop_length=160, n_mels=80, n_mfcc=13)
    results["extra_nfft640"] = run(asr640, pcm, so, steps=10, warmup=2)
    for n in [n for n in range(128, 1009, 16) if n not in (256, 320, 400, 480, 512, 640)]:      # (those six are timed above)
        cfg = S.make_config(sr=16000, n_fft=n, hop_length=min(160, n // 2), n_mels=80 if n >= 320 else 40, n_mfcc=13)
        results[f"extra_nfft{n}"] = run(cfg, pcm, so, steps=5, warmup=2)
    results["extra_nfft1024_hop256"] = run(S.make_config(sr=16000, n_fft=1024, hop_length=256, n_mels=80, n_mfcc=13),
                                           pcm, so, steps=10, warmup=2)
    asr600 = S.make_config(sr=16000, n_fft=600, hop_length=160, n_mels=80, n_mfcc=13)
    results["extra_nfft600_dense_dft"] = run(asr600, pcm, so, steps=5, warmup=2)
    del pcm

    pcm5, so5 = make_batch(args.hours, 5, dev, lo_s=30.0, hi_s=60.0)
    a, b = int(so5[1]), int(so5[2])                 # utterance 1: -80 dB noise and one click
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    pcm5[a:b] = 1e-4 * torch.randn(b - a, generator=g, device=dev)
    pcm5[(a + b) // 2] = 0.9
    results["config5_long_utterances"] = run(ref, pcm5, so5, steps=20, warmup=3)

    text = json.dumps(results, indent=1)
    print(text)
    if args.out:
        Path(args.out).write_text(text)


if __name__ == "__main__":
    main()
