#!/usr/bin/env python3
"""Which front-end kernel serves which (n_fft, n_mels): builds a plan for every multiple of 16 from
128 to 1024 plus 2048, at several filterbank sizes and sample rates, and prints the kernel
names -- a 'dense_dft' in an in-register length's row means the filterbank did not fit its LDS.
    python tools/kernel_map.py [out.json]"""
import json
import sys
import warnings
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import speech_to_text_amd as S  # noqa: E402

warnings.simplefilter("ignore")
lengths = sorted(set(range(128, 1025, 16)) | {2048})
mels = (13, 23, 40, 64, 80, 128, 160, 256)
rates = (8000, 16000, 22050, 48000)
table, misses = {}, []
for n in lengths:
    row = {}
    for m in mels:
        names = []
        for sr in rates:
            plan = S.MfccPlan(S.make_config(n_fft=n, hop_length=max(1, n // 4), n_mels=m, n_mfcc=min(13, m), sr=sr))
            names.append(plan.front_kernel)
            if plan.front_kernel == "dense_dft":
                misses.append((n, m, sr))
            plan.close()
        row[m] = names[0] if len(set(names)) == 1 else "/".join(names)
    table[n] = row
    print(n, " ".join(f"{m}:{row[m]}" for m in mels), flush=True)
print(f"{len(misses)} of {len(lengths) * len(mels) * len(rates)} combinations fall back to the dense-DFT kernel")
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps({"rates": rates, "table": table, "dense_dft": misses}, indent=1))
