#!/bin/bash
# usage: pmc_lds_multi.sh <outdir> <hours> <ablate values...> -- LDS counters per ablation setting
O=$1; H=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
for A in "$@"; do
  echo "== ablate=$A"
  bash $R/tools/pmc_lds.sh $O/a$A $H $A | grep -E "LDS|VALU|WAVE_CYCLES" || exit 1
done
