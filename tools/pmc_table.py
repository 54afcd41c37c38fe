"""Per-dispatch table of rocprofv3 --pmc CSV output (one line per dispatch: kernel, duration when a
kernel trace is present, every counter): for runs whose dispatches of one kernel differ (the
microbenchmarks launch each kernel with several grid sizes)."""
import collections, csv, glob, sys

src = sys.argv[1]
rows = collections.OrderedDict()
for path in sorted(glob.glob(src + "/**/*counter_collection.csv", recursive=True)):
    pass_name = path[len(src):].strip("/").split("/")[0]
    seq = collections.Counter()
    for r in csv.DictReader(open(path)):
        key = (r["Kernel_Name"], r.get("Grid_Size", ""), r["Dispatch_Id"])
        rows.setdefault((pass_name, key), {})[r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for path in sorted(glob.glob(src + "/**/*kernel_trace.csv", recursive=True)):
    pass_name = path[len(src):].strip("/").split("/")[0]
    for r in csv.DictReader(open(path)):
        dur[(pass_name, r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for (pass_name, (k, grid, disp)), cs in rows.items():
    d = dur.get((pass_name, disp))
    print(f"{pass_name} {k[:48]:48s} grid {grid:>8s} disp {disp:>4s} " + (f"{d / 1e3:9.1f} us " if d else "") +
          " ".join(f"{c}={v:.0f}" for c, v in sorted(cs.items())))
