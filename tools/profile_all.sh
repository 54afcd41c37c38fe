#!/bin/bash
# usage: tools/profile_all.sh <tag>   (on the GPU box)
# Every measurement the project quotes, in one go: the bench line, kernel trace + HBM traffic
# passes, SQ / MFMA counter passes of the four pipelines, the other BASELINE configurations and
# the 1000-hour run.  Results land under gpurun_out/<tag>*; tools/collect_profiles.sh copies the
# summaries into profiles/.
set -e
TAG=${1:?usage: tools/profile_all.sh <tag>}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
timeout -k 10 500 bash tools/pmc_traffic.sh ${TAG}_traffic > $O/traffic.txt 2>&1
timeout -k 10 800 bash tools/sq_counters.sh ${TAG} 10 ref > $O/sq_ref.txt 2>&1      # the bench's own 10-hour batch
timeout -k 10 500 bash tools/sq_counters.sh ${TAG}_asr512 2 asr512 > $O/sq_asr512.txt 2>&1
timeout -k 10 500 bash tools/sq_counters.sh ${TAG}_asr400 2 asr400 > $O/sq_asr400.txt 2>&1
timeout -k 10 500 bash tools/sq_counters.sh ${TAG}_fft600 0.5 fft600 > $O/sq_fft600.txt 2>&1
timeout -k 10 600 python tools/bench_configs.py --out $O/other_configs.json > $O/other.txt 2>&1
timeout -k 10 600 python tools/bench_config3.py --out $O/config3_1000h.json > $O/config3.txt 2>&1
timeout -k 10 300 python bench.py > $O/bench_after.json 2> $O/bench_after.err
cut -c1-300 $O/bench_after.json
echo done
