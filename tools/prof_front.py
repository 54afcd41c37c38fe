"""Runs the pipeline a few times on a small batch; meant to be run under rocprofv3 --pmc.

    python tools/prof_front.py [hours] [iterations] [ref|asr512|asr400|fft1024|fft600]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import speech_to_text_amd as S
from bench import make_batch

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
which = sys.argv[3] if len(sys.argv) > 3 else "ref"
cfg = {"ref": S.MfccConfig.reference_default(),
       "asr512": S.make_config(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13),
       "asr400": S.make_config(sr=16000, n_fft=400, hop_length=160, n_mels=80, n_mfcc=13),
       "fft1024": S.make_config(sr=16000, n_fft=1024, hop_length=256, n_mels=80, n_mfcc=13),
       "fft600": S.make_config(sr=16000, n_fft=600, hop_length=160, n_mels=80, n_mfcc=13)}[which]
pcm, so = make_batch(hours, 1234, dev)
plan = S.MfccPlan(cfg, dev)
fo = plan.frame_offsets(so)
F = int(fo[-1]); d_so = torch.from_numpy(so).to(dev); d_fo = torch.from_numpy(fo).to(dev)
for _ in range(iters):
    plan.execute(pcm, d_so, d_fo, F)
torch.cuda.synchronize()
print("frames", F, "utts", len(so) - 1)
