"""Summarises rocprofv3 --pmc CSV output: per kernel-name mean of each counter."""
import csv, sys, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"][:60]
            rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for name, cs in rows.items():
    if pat not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"  {c:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
