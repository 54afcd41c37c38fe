#!/usr/bin/env python3
"""Sweep of the host staging knobs (chunk size, packing threads, writer threads) for the drop-in
step: every setting in a fresh child process, steady-state wall clock of Preprocessor.preprocess
(second of two runs) for 10 h of float32 waveforms in host memory.

    python tools/sweep_stage.py [hours]"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r"""
import sys, time, tempfile, shutil, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import speech_to_text_amd as S
from bench_preprocess import make_items
items, hours = make_items(%f)
if %d:
    items = [(np.rint(it[0] * 32768.0).astype(np.int16),) + tuple(it[1:]) for it in items]
d = tempfile.mkdtemp(prefix="sw_", dir="/dev/shm")
best = None
for rep in range(3):
    pre = S.Preprocessor(items, d + f"/o{rep}", ingest=%r)
    t0 = time.perf_counter(); pre.preprocess(verbose=False, multiprocess=True); dt = time.perf_counter() - t0
    if rep and (best is None or dt < best[0]):
        best = (dt, pre.stats["stage_seconds"])
shutil.rmtree(d)
print(json.dumps({"seconds": round(best[0], 4), "audio_h_per_s": round(hours / best[0], 1),
                  "busy": {k: round(v, 4) for k, v in best[1].items()}}))
"""


CHUNKS = [int(x) for x in os.environ.get('SWEEP_CHUNKS', '16,32,64').split(',')]
THREADS = [int(x) for x in os.environ.get('SWEEP_THREADS', '6,10,15').split(',')]


def main():
    hours = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    for src16, ingest in ((0, "float32"), (0, "int16"), (1, "float32")):
        for chunk in CHUNKS:
            for threads in THREADS:
                env = dict(os.environ, MFCC_STAGE_CHUNK_MIB=str(chunk), MFCC_STAGE_THREADS=str(threads))
                out = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), str(ROOT / "tools"), hours, src16, ingest)],
                                     env=env, capture_output=True, text=True)
                line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
                print(f"source {'int16' if src16 else 'float32'} ingest {ingest:8s} chunk {chunk:2d} MiB threads {threads:2d}: {line}",
                      flush=True)


if __name__ == "__main__":
    main()
