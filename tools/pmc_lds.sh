#!/bin/bash
# usage: pmc_lds.sh <outdir> <hours> [ablate] [config] -- one PMC pass with the LDS counters.  Development aid.
O=$1; H=${2:-1}; export MFCC_HIP_ABLATE=${3:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/p -- python3 $R/tools/prof_front.py $H 2 ${4:-ref} > $O/p.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $O k_melspec
