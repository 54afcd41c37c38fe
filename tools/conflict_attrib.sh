#!/bin/bash
# usage: tools/conflict_attrib.sh   (after: for n in 0 1 16 128 256 257; do tools/build_ab.sh abl$n -DMFCC_ABL=$n; done)
# LDS bank-conflict and LDS-array cycles per frame of the general 2048-point kernel with one stage
# compiled out at a time (MFCC_ABL bits: 1 mel, 16 exchange, 128 column 0, 256 power store):
# the difference to the full kernel attributes the cycles to the stage.  Development aid.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/conflicts; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export MFCC_HIP_FAST=0
for n in 0 1 16 128 256 257; do
    export MFCC_HIP_LIBRARY=$R/build/ab/abl$n.so
    timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/abl$n -- python3 $R/tools/prof_front.py 2 3 ref > $O/abl$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for n in (0, 1, 16, 128, 256, 257):
    acc = collections.defaultdict(float); disp = set()
    for f in glob.glob("$O/abl%d/*/*counter_collection.csv" % n):
        for r in csv.DictReader(open(f)):
            if "k_melspec" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    frames = 225199.0 * max(1, len(disp))
    print("MFCC_ABL=%-4d" % n, {k: round(v / frames, 1) for k, v in sorted(acc.items())})
PY
