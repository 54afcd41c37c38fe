#!/bin/bash
# usage: tools/sq_counters.sh <tag> [hours] [config]
# SQ / LDS / MFMA counter passes of the pipeline's kernels (rocprofv3 --pmc runs of tools/prof_front.py,
# at most 8 SQ counters each), summarised per frame into profiles/<tag>_sq_counters.json.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; H=${2:-2}; CFG=${3:-ref}
O=$R/gpurun_out/sq_$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $R/tools/prof_front.py $H 3 $CFG > $O/p1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d $O/p2 -- python3 $R/tools/prof_front.py $H 3 $CFG > $O/p2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_FMA_F32 SQ_IFETCH --output-format csv -d $O/p3 -- python3 $R/tools/prof_front.py $H 3 $CFG > $O/p3.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p4 -- python3 $R/tools/prof_front.py $H 3 $CFG > $O/p4.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_CYCLES --output-format csv -d $O/p5 -- python3 $R/tools/prof_front.py $H 3 $CFG > $O/p5.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $O/p6 -- python3 $R/tools/prof_front.py $H 3 $CFG > $O/p6.log 2>&1
python3 $R/tools/make_sq_summary.py $O $TAG $CFG
