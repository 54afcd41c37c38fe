#!/bin/bash
# usage: tools/sq_calibrate.sh <tag> [micro binary, default valu_rate]
# The VALU-busy counters of tools/sq_counters.sh on kernels whose VALU load is known: the loops of
# tools/micro/valu_rate (one instruction kind each, 1 / 2 / 4 wavefronts per SIMD).  Shows what
# SQ_THREAD_CYCLES_VALU and SQ_ACTIVE_INST_VALU2 read when the VALU is saturated, so that the
# same counters of the front-end kernel can be read as a busy fraction.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; BIN=${2:-valu_rate}
O=$R/gpurun_out/cal_$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_CYCLES --output-format csv -d $O/c1 -- $R/tools/micro/$BIN > $O/c1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/c2 -- $R/tools/micro/$BIN > $O/c2.log 2>&1
python3 $R/tools/pmc_table.py $O > $O/table.txt
cp $O/c1.log $O/$BIN.txt
