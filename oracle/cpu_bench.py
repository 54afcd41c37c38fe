"""CPU baseline timer for bench.py (test/measurement infrastructure, never on the product path).

Times the numpy/scipy restatement of ``librosa.feature.mfcc`` (oracle/mfcc_ref.py) on utterances
stored in two .npy files, either in one process or in a worker pool shaped like the reference's
``mp.Pool(...).imap_unordered(process_audio, data)`` (/root/reference/src/preprocess.py:156-160).
Runs in its own process so that no GPU state is inherited by the workers.

    python -m oracle.cpu_bench <pcm.npy> <offsets.npy> <n_utts> <procs> <repeats>

prints one JSON line: {"seconds": ..., "hours": ..., "procs": ...}.
"""
from __future__ import annotations

import json
import multiprocessing as mp
import sys
import time

import numpy as np

_PCM = None
_SO = None


def _init(pcm_path, so_path):
    global _PCM, _SO
    _PCM = np.load(pcm_path, mmap_mode="r")
    _SO = np.load(so_path)
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        pass


def _one(i):
    from oracle import mfcc_ref as R

    y = np.asarray(_PCM[int(_SO[i]): int(_SO[i + 1])])
    return R.process_audio_features(y).shape[0]


def main(argv):
    pcm_path, so_path, n_utts, procs, repeats = argv[0], argv[1], int(argv[2]), int(argv[3]), int(argv[4])
    _init(pcm_path, so_path)
    idx = list(range(n_utts)) * repeats
    _one(0)  # imports, plan caches
    if procs <= 1:
        t0 = time.perf_counter()
        for i in idx:
            _one(i)
        dt = time.perf_counter() - t0
    else:
        with mp.Pool(procs, initializer=_init, initargs=(pcm_path, so_path)) as pool:
            list(pool.imap_unordered(_one, range(min(n_utts, 2 * procs))))   # warm the workers
            t0 = time.perf_counter()
            for _ in pool.imap_unordered(_one, idx, chunksize=4):
                pass
            dt = time.perf_counter() - t0
    hours = repeats * float(_SO[n_utts] - _SO[0]) / 16000.0 / 3600.0
    print(json.dumps({"seconds": dt, "hours": hours, "procs": procs}))


if __name__ == "__main__":
    main(sys.argv[1:])
