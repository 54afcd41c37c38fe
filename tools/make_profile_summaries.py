"""Turns a gpurun_out/<traffic dir> (made by tools/pmc_traffic.sh) into the committed summaries
under profiles/: the rocprofv3 --kernel-trace --stats table of the bench command and the HBM
traffic of the dominant kernel from separate --pmc passes.

HBM bytes follow the MI355X guide's counter notes: TCC_EA0_RDREQ counts fabric read requests of
32, 64 or 128 bytes (the 128-byte ones are the wide coalesced reads that FETCH_SIZE tallies at
half their size on gfx950), TCC_EA0_WRREQ counts 32- or 64-byte write requests."""
import csv, glob, json, os, sys, collections, shutil
from pathlib import Path


def newest_per_directory(paths):
    """gpurun merges every run's files into the same local directory (rocprofv3 names them by
    process id): of each directory only the most recent file belongs to the run being summarised."""
    best = {}
    for q in paths:
        d = os.path.dirname(q)
        if d not in best or os.path.getmtime(q) > os.path.getmtime(best[d]):
            best[d] = q
    return sorted(best.values())


src = Path(sys.argv[1]); tag = sys.argv[2]
command = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
out = Path(__file__).resolve().parent.parent / "profiles"
out.mkdir(exist_ok=True)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in newest_per_directory(glob.glob(str(src / "**/*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in vals.items():
    if "k_melspec" not in k and "k_db_dct" not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rd128, rd32, rd = m.get("TCC_EA0_RDREQ_128B_sum", 0), m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_sum", 0)
    wr, wr64 = m.get("TCC_EA0_WRREQ_sum", 0), m.get("TCC_EA0_WRREQ_64B_sum", 0)
    read_b = 128 * rd128 + 32 * rd32 + 64 * (rd - rd128 - rd32)
    write_b = 64 * wr64 + 32 * (wr - wr64)
    summary[k] = {"counters_mean_per_launch": m, "hbm_read_bytes": read_b, "hbm_write_bytes": write_b,
                  "hbm_bytes": read_b + write_b,
                  "fetch_size_x2_plus_write_size_bytes": (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024}
stats = newest_per_directory(glob.glob(str(src / "trace/**/*kernel_stats.csv"), recursive=True))
rows = []
if stats:
    shutil.copy(stats[0], out / f"{tag}_bench_kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(stats[0])) if "mfcc" in r["Name"].lower() or "k_" in r["Name"]]
bench_line = None
log = src / "trace.log"
if log.exists():
    for line in log.read_text().splitlines():
        if line.startswith("{") and '"metric"' in line:
            bench_line = json.loads(line)
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import csrc_sha256  # noqa: E402  (bench.py quotes this profile only for the same kernel sources)

ran = ((bench_line or {}).get("roofline") or {}).get("csrc_sha256")
if ran is not None and ran != csrc_sha256():
    sys.exit(f"the profiled run used a library built from other kernel sources ({ran[:12]}) than the tree holds ({csrc_sha256()[:12]})")
json.dump({"command": f"rocprofv3 ... -- {command} (one run per counter set)",
           "csrc_sha256": csrc_sha256(),
           "kernels": summary, "kernel_stats": rows, "bench_line_under_profiler": bench_line},
          open(out / f"{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: {"hbm_bytes": v["hbm_bytes"]} for k, v in summary.items()}, indent=1))
