#!/bin/bash
# usage: ablate_run.sh <hours> <variants> <ablate values...> -- development aid
H=$1; V=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
for A in "$@"; do
  echo "== ablate=$A"
  MFCC_HIP_ABLATE=$A timeout -k 10 120 python $R/tools/variant_sweep.py $H $V 5 2>&1 | grep "front median" || exit 1
done
