#!/bin/bash
# usage: ab_run.sh <hours> <variants> <ablate> lib1.so lib2.so ... -- times tools/variant_sweep.py per library.
# Development aid.
H=$1; V=$2; A=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
for L in "$@"; do
  echo "== $L ablate=$A"
  MFCC_HIP_LIBRARY=$R/speech_to_text_amd/lib/ab/$L MFCC_HIP_ABLATE=$A timeout -k 10 120 python $R/tools/variant_sweep.py $H $V 5 2>&1 | grep "front median" || exit 1
done
