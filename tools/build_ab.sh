#!/bin/bash
# usage: tools/build_ab.sh <name> [extra hipcc flags...] -- builds the current csrc/ into build/ab/<name>.so
# (for same-box A/B timing with tools/lib_ab.py).  Development aid.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
mkdir -p $R/build/ab
make -j8 -C $R/speech_to_text_amd/csrc OUT=$R/build/ab/$N.so OBJDIR=$R/build/ab/obj_$N EXTRA="$*" > /dev/null
rm -rf $R/build/ab/obj_$N
echo built build/ab/$N.so
