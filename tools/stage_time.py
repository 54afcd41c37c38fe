"""Time of staging.upload_ragged for one 120 MB batch in its three modes.  python tools/stage_time.py"""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from speech_to_text_amd import staging
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
lens = [int(x) for x in rng.integers(32000, 560000, size=100)]
ys = [(rng.integers(-20000, 20000, size=n).astype(np.float32) / 32768.0) for n in lens]
ys16 = [np.rint(y * 32768).astype(np.int16) for y in ys]
so = np.zeros(len(ys) + 1, dtype=np.int64); np.cumsum(lens, out=so[1:])
print("MB f32", so[-1] * 4 / 1e6)
for mode, ch in (("f32", ys), ("f32_to_i16", ys), ("i16", ys16), ("f32", ys), ("f32_to_i16", ys)):
    staging.upload_ragged(ch, so, dev, mode); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        staging.upload_ragged(ch, so, dev, mode)
    torch.cuda.synchronize()
    print(mode, round((time.perf_counter() - t) / 5 * 1e3, 2), "ms")
