#!/usr/bin/env python3
"""Instruction histogram of one kernel from `hipcc -S --cuda-device-only` output.

    python tools/isa_hist.py all.s <mangled kernel name> [out.s]
"""
import collections
import re
import sys


def extract(path, name):
    lines, on = [], False
    for ln in open(path):
        if ln.startswith(name + ":"):
            on = True
        if on:
            lines.append(ln)
            if ln.startswith(".Lfunc_end"):
                break
    return lines


def main():
    src, name = sys.argv[1], sys.argv[2]
    body = extract(src, name)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").writelines(body)
    hist = collections.Counter()
    for ln in body:
        m = re.match(r"\s+(v_\w+|s_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)", ln)
        if m:
            hist[m.group(1)] += 1
    valu = sum(c for k, c in hist.items() if k.startswith("v_"))
    lds = sum(c for k, c in hist.items() if k.startswith("ds_"))
    print(f"{len(body)} lines, VALU {valu}, LDS {lds}")
    for k, c in hist.most_common(45):
        print(f"{c:6d} {k}")


if __name__ == "__main__":
    main()
