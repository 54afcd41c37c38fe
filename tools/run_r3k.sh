O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for c in asr400 asr480 asr640; do timeout -k 10 200 python tools/lib_ab.py $c 10 build/ab/base.so build/ab/edge16a.so > $O/ab_$c.txt 2>&1 || { tail $O/ab_$c.txt; exit 1; }; echo $c; cat $O/ab_$c.txt; done
