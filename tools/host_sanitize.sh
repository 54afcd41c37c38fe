#!/bin/bash
# Host code of the library under sanitizers (the GPU code is built as usual and not run):
#   1. AddressSanitizer + UBSan   2. ThreadSanitizer
# each a separate build of the whole library under build/ (csrc/Makefile with HOSTFLAGS), with tests/csrc/host_stress.cpp linked
# against it.  CPU only; about four minutes.  usage: tools/host_sanitize.sh [log file]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CLANGXX="${CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}"
LOG="${1:-/dev/stdout}"
case "$LOG" in /*) ;; *) LOG="$PWD/$LOG" ;; esac
run() {   # name, host flags...
    local name="$1"; shift
    local out="$ROOT/build/$name"
    mkdir -p "$out"
    make -j8 -C "$ROOT/speech_to_text_amd/csrc" OUT="$out/libmfcc_hip.so" OBJDIR="$out/obj" HOSTFLAGS="$*" > /dev/null
    "$CLANGXX" -O1 -g -std=c++17 "$@" -I"$ROOT/include" "$ROOT/tests/csrc/host_stress.cpp" -o "$out/host_stress" \
        -L"$out" -lmfcc_hip -Wl,-rpath,"$out" -pthread
    local dir
    dir="$(mktemp -d)"
    echo "== $name: $*" >> "$LOG"
    "$out/host_stress" "$dir" >> "$LOG" 2>&1
    rmdir "$dir"
}
[ "$LOG" = /dev/stdout ] || : > "$LOG"
run san -fsanitize=address -fsanitize=undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
run tsan -fsanitize=thread
echo "host sanitizers: clean" >> "$LOG"
