"""Runs the pipeline a few times on a small batch; meant to be run under rocprofv3 --pmc."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import speech_to_text_amd as S
from bench import make_batch

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
cfg = S.MfccConfig.reference_default()
pcm, so = make_batch(hours, 1234, dev)
plan = S.MfccPlan(cfg, dev)
fo = plan.frame_offsets(so)
F = int(fo[-1]); d_so = torch.from_numpy(so).to(dev); d_fo = torch.from_numpy(fo).to(dev)
for _ in range(iters):
    plan.execute(pcm, d_so, d_fo, F)
torch.cuda.synchronize()
print("frames", F, "utts", len(so) - 1)
