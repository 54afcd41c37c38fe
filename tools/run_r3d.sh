mkdir -p gpurun_out/r3d
timeout -k 10 400 python tools/bench_preprocess.py --out gpurun_out/r3d/preprocess_host.json > gpurun_out/r3d/pre.txt 2>&1 || { tail -30 gpurun_out/r3d/pre.txt; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3d/preprocess_host.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'audio_hours_per_s' in v: print(k, round(v['seconds'],4), round(v['audio_hours_per_s'],1), (v.get('stats') or {}).get('stage_seconds'))
for k,v in d.get('stages',{}).items(): print('  stage',k,v if not isinstance(v,dict) else {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
PY
timeout -k 10 200 python - <<'PY'
import sys, time, tempfile, shutil
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import speech_to_text_amd as S
from bench_preprocess import make_items
items, hours = make_items(10.0)
d = tempfile.mkdtemp(prefix="tl_", dir="/dev/shm")
for rep in range(2):
    pre = S.Preprocessor(items, d + f"/o{rep}")
    t0 = time.perf_counter(); pre.preprocess(verbose=False, multiprocess=True); t1 = time.perf_counter()
    print("wall", t1 - t0, pre.stats["stage_seconds"])
tl = pre.timeline
base = min(a for v in tl.values() for a, _ in v)
for k in ("stage", "compute", "fetch", "write"):
    print(k, " ".join(f"{(a-base)*1e3:.1f}-{(b-base)*1e3:.1f}" for a, b in tl[k][:10]))
shutil.rmtree(d)
PY
