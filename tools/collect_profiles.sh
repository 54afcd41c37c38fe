#!/bin/bash
# usage: tools/collect_profiles.sh <tag>   (in the build container, after tools/profile_all.sh <tag> ran on the GPU box)
# Copies the summaries of gpurun_out/<tag>* into profiles/ (tracked).
set -e
TAG=${1:?usage: tools/collect_profiles.sh <tag>}; O=gpurun_out/$TAG
python3 tools/make_profile_summaries.py gpurun_out/${TAG}_traffic $TAG > /dev/null
for t in $TAG ${TAG}_asr512 ${TAG}_asr400 ${TAG}_fft600; do cp gpurun_out/sq_$t/${t}_sq_counters.json profiles/; done
cp $O/other_configs.json profiles/${TAG}_other_configs.json
cp $O/config3_1000h.json profiles/${TAG}_config3_1000h.json
cp $O/bench.json profiles/${TAG}_bench_line.json
