#!/bin/bash
# usage: share_run.sh <share values...> -- time, parity and L2->fabric read requests per setting.  Development aid.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for S in "$@"; do
  echo "== share=$S"
  export MFCC_HIP_SHARE=$S
  timeout -k 10 120 python $R/tools/variant_sweep.py 2 10 5 2>&1 | grep -E "front median|max" || exit 1
  O=$R/gpurun_out/share$S; mkdir -p $O
  timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum --output-format csv -d $O -- python3 $R/tools/prof_front.py 2 2 > $O.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $O k_melspec | grep RDREQ_128B
done
