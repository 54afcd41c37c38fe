#!/bin/bash
# usage: pmc_traffic_cfg.sh <outdir name> <config> [hours] -- kernel trace and HBM-traffic counter
# passes of tools/prof_front.py for one of its configurations (ref, asr512, asr400, fft1024).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-traffic_cfg}; C=${2:-asr512}; H=${3:-10}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/prof_front.py $H 5 $C > $O/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/req -- python3 $R/tools/prof_front.py $H 3 $C > $O/req.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/l2 -- python3 $R/tools/prof_front.py $H 3 $C > $O/l2.log 2>&1
python3 $R/tools/pmc_summary.py $O k_melspec
python3 $R/tools/pmc_summary.py $O k_db_dct
cat $O/trace/*/*kernel_stats.csv | cut -c1-160 | grep -E "Name|k_melspec|k_db_dct|k_prepare"
