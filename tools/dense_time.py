#!/usr/bin/env python3
"""Throughput of the dense-DFT fallback kernel (csrc/front_generic.h) at transform lengths that have no
in-register kernel, on a 2-hour batch:  python tools/dense_time.py"""
import sys, time, json
sys.path.insert(0, ".")
import torch, numpy as np
import speech_to_text_amd as S
from bench import make_batch
dev = torch.device("cuda", 0)
pcm, so = make_batch(2.0, 1234, dev)
d_so = torch.from_numpy(so).to(dev)
for n, hop, mels in ((600, 160, 80), (200, 80, 40), (1104, 256, 80), (1200, 480, 80), (1536, 384, 128)):
    plan = S.MfccPlan(S.make_config(sr=16000, n_fft=n, hop_length=hop, n_mels=mels, n_mfcc=13), dev)
    fo = plan.frame_offsets(so); F = int(fo[-1]); d_fo = torch.from_numpy(fo).to(dev)
    for _ in range(2): plan.execute(pcm, d_so, d_fo, F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): plan.execute(pcm, d_so, d_fo, F)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(n, plan.front_kernel, f"{2.0 / dt:.0f} audio-h/s", f"{dt*1e3:.2f} ms", flush=True)
    plan.close()
