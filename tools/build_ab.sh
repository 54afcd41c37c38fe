#!/bin/bash
# usage: tools/build_ab.sh <name> [extra hipcc flags...] -- builds the current csrc/ into build/ab/<name>.so
# (for same-box A/B timing with tools/lib_ab.py).  Development aid.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
mkdir -p $R/build/ab
cd $R/speech_to_text_amd/csrc
/opt/rocm/bin/hipcc -fno-slp-vectorize -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wall -Wno-unused-function "$@" \
    -x hip mfcc_hip.hip -x hip host_tables.cpp -x hip hdf5_file.cpp -o $R/build/ab/$N.so
echo built build/ab/$N.so
