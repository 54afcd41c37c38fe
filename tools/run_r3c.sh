mkdir -p gpurun_out/r3c
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3c/pytest.log 2>&1 || { tail -30 gpurun_out/r3c/pytest.log; exit 1; }
tail -3 gpurun_out/r3c/pytest.log
timeout -k 10 300 python tools/lib_ab.py ref 10 build/ab/edge0.so build/ab/edge1.so build/ab/mb4.so > gpurun_out/r3c/ab.txt 2>&1 || { tail gpurun_out/r3c/ab.txt; exit 1; }
cat gpurun_out/r3c/ab.txt
timeout -k 10 400 python tools/bench_preprocess.py --out gpurun_out/r3c/preprocess_host.json > gpurun_out/r3c/pre.txt 2>&1 || { tail -30 gpurun_out/r3c/pre.txt; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c/preprocess_host.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'audio_hours_per_s' in v: print(k, round(v['seconds'],4), round(v['audio_hours_per_s'],1))
for k,v in d.get('stages',{}).items(): print('  stage',k,v if not isinstance(v,dict) else {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
PY
