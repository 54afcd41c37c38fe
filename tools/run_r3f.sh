mkdir -p gpurun_out/r3f
echo skip-pytest

timeout -k 10 300 python tools/soak.py --seconds 60 --seed 31 --lengths 128,160,192,384,576,768,400,2048 > gpurun_out/r3f/soak_new_lengths.log 2>&1 || { tail -20 gpurun_out/r3f/soak_new_lengths.log; exit 1; }
tail -2 gpurun_out/r3f/soak_new_lengths.log
timeout -k 10 400 python tools/bench_preprocess.py --out gpurun_out/r3f/preprocess_host.json > gpurun_out/r3f/pre.txt 2>&1 || { tail -30 gpurun_out/r3f/pre.txt; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3f/preprocess_host.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'audio_hours_per_s' in v: print(k, round(v['seconds'],4), round(v['audio_hours_per_s'],1), (v.get('stats') or {}).get('stage_seconds'))
for k,v in d.get('stages',{}).items(): print('  stage',k,v if not isinstance(v,dict) else {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
PY
