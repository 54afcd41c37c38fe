set -e
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
timeout -k 10 500 bash tools/pmc_traffic.sh r3g_traffic > $O/traffic.txt 2>&1
tail -8 $O/traffic.txt
timeout -k 10 500 bash tools/sq_counters.sh round3 2 ref > $O/sq_ref.txt 2>&1
timeout -k 10 500 bash tools/sq_counters.sh round3_asr512 2 asr512 > $O/sq_asr512.txt 2>&1
timeout -k 10 500 bash tools/sq_counters.sh round3_asr400 2 asr400 > $O/sq_asr400.txt 2>&1
timeout -k 10 500 bash tools/sq_counters.sh round3_fft600 0.5 fft600 > $O/sq_fft600.txt 2>&1
tail -3 $O/sq_fft600.txt
timeout -k 10 600 python tools/bench_configs.py --out $O/other_configs.json > $O/other.txt 2>&1
timeout -k 10 600 python tools/bench_config3.py --out $O/config3_1000h.json > $O/config3.txt 2>&1
tail -5 $O/config3.txt | cut -c1-300
echo done
