#!/bin/bash
# HBM traffic of the bench workload: separate --pmc passes (TCC slots), kernel-trace in its own run.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-traffic}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/req -- $B > $O/req.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/l2 -- $B > $O/l2.log 2>&1
python3 $R/tools/pmc_summary.py $O k_melspec
python3 $R/tools/pmc_summary.py $O k_db_dct
cat $O/trace/*/*kernel_stats.csv | cut -c1-160 | grep -E "Name|k_melspec|k_db_dct|k_prepare"
tail -1 $O/trace.log | cut -c1-400
