"""Times every front-end kernel variant on the same batch in one process (interleaved rounds)
and checks each against the CPU oracle on a few utterances.  Development aid."""
import os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import speech_to_text_amd as S
from bench import make_batch
from oracle import mfcc_ref as R

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(7))
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
cfg = S.MfccConfig.reference_default()
pcm, so = make_batch(hours, 1234, dev)
plans = {}
for v in variants:
    os.environ["MFCC_HIP_VARIANT"] = str(v)
    plans[v] = S.MfccPlan(cfg, dev)
    plans[v].set_profiling(True)
fo = plans[variants[0]].frame_offsets(so)
F = int(fo[-1]); d_so = torch.from_numpy(so).to(dev); d_fo = torch.from_numpy(fo).to(dev)
nchk = 3
host = pcm[: int(so[nchk])].cpu().numpy()
want = np.concatenate([R.process_audio_features(host[int(so[i]):int(so[i + 1])]) for i in range(nchk)])
times = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        out = plans[v].execute(pcm, d_so, d_fo, F)
        a, b = plans[v].last_timings()
        if r == 0:
            err = float(np.abs(out[: int(fo[nchk])].cpu().numpy() - want).max())
            print(f"variant {v}: max|diff| vs oracle = {err:.3e} dB", flush=True)
        else:
            times[v].append(a)
ah = so[-1] / 16000 / 3600
for v in variants:
    t = np.array(times[v])
    print(f"variant {v}: front median {np.median(t):.3f} ms  min {t.min():.3f} ms -> {ah / (np.median(t) * 1e-3):.0f} audio-h/s  ({F / np.median(t) * 1e-6:.1f} Mframes/s)")
