"""Turns the counter passes of tools/sq_counters.sh into profiles/<tag>_sq_counters.json: per
kernel, the mean of every counter per launch and the same divided by the frames of the launch.

Units: SQ_INSTS_* count wavefront instructions; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count
quad-cycles (x4 = shader cycles) summed over wavefronts; SQ_BUSY_CYCLES is summed over the shader
engines; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT are LDS-array cycles summed over the CUs, so
per frame they read directly as "LDS cycles one frame costs its CU"; GRBM_GUI_ACTIVE is summed over
the 8 XCDs (/ 8 / kernel time = the clock the chip held)."""
import collections, csv, glob, json, re, sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "ref"
root = Path(__file__).resolve().parent.parent
frames = None
for log in sorted(src.glob("p*.log")):
    m = re.search(r"frames (\d+) utts (\d+)", log.read_text())
    if m:
        frames = int(m.group(1))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(str(src / "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for path in glob.glob(str(src / "**/*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {"command": f"tools/sq_counters.sh {tag} (rocprofv3 --pmc passes of tools/prof_front.py, config {cfg})",
       "frames_per_launch": frames, "kernels": {}}
sys.path.insert(0, str(root))
from bench import csrc_sha256  # noqa: E402
out["csrc_sha256"] = csrc_sha256()
for k, cs in vals.items():
    if "k_melspec" not in k and "k_db_dct" not in k:
        continue
    mean = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    e = {"per_launch": mean}
    if frames:
        e["per_frame"] = {c: v / frames for c, v in mean.items()}
        pf = e["per_frame"]
        d = {}
        if "SQ_LDS_IDX_ACTIVE" in pf:
            d["lds_array_cycles_per_frame"] = pf["SQ_LDS_IDX_ACTIVE"]
            d["lds_bank_conflict_cycles_per_frame"] = pf.get("SQ_LDS_BANK_CONFLICT")
        if "SQ_WAVE_CYCLES" in pf and "SQ_WAVES" in mean:
            d["wave_shader_cycles_per_frame"] = 4.0 * pf["SQ_WAVE_CYCLES"]
        for name, key in (("wait_fraction", "SQ_WAIT_ANY"), ("issue_stall_fraction", "SQ_WAIT_INST_ANY"),
                          ("issuing_fraction", "SQ_ACTIVE_INST_ANY"), ("valu_issue_fraction", "SQ_ACTIVE_INST_VALU"),
                          ("lds_issue_fraction", "SQ_ACTIVE_INST_LDS")):
            if key in pf and "SQ_WAVE_CYCLES" in pf:
                d[name + "_of_wave_cycles"] = pf[key] / pf["SQ_WAVE_CYCLES"]
        e["derived"] = d
        # Vector-issue occupancy.  A SIMD has one vector-issue slot per quad-cycle; it takes one
        # instruction, or two from different wavefronts when both are plain VOP1/VOP2/VOP3 forms
        # (SQ_ACTIVE_INST_VALU2 counts those slots; DPP, packed, scalar-register-operand and
        # v_cndmask forms never pair -- tools/sq_calibrate.sh).  Slots used = instructions - pairs;
        # slots available = SQ_BUSY_CU_CYCLES (quad-cycles per SIMD; reads 1.01-1.04 x the slots
        # used in the saturated loops of tools/micro/valu_rate).
        if all(c in pf for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_BUSY_CU_CYCLES")):
            used = pf["SQ_INSTS_VALU"] - pf["SQ_ACTIVE_INST_VALU2"]
            e["valu"] = {"instructions_per_frame": pf["SQ_INSTS_VALU"],
                         "dual_issued_pairs_per_frame": pf["SQ_ACTIVE_INST_VALU2"],
                         "issue_slots_used_per_frame": used,
                         "issue_slots_available_per_frame": pf["SQ_BUSY_CU_CYCLES"],
                         "issue_slot_occupancy": used / pf["SQ_BUSY_CU_CYCLES"],
                         "unit": "quad-cycles (4 shader cycles) of one SIMD"}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in pf and pf.get("SQ_INSTS_MFMA", 0) > 0 and "SQ_BUSY_CU_CYCLES" in pf:
            # SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles, SQ_BUSY_CU_CYCLES quad-cycles
            e["mfma"] = {"instructions_per_frame": pf["SQ_INSTS_MFMA"],
                         "mops_f32_per_frame_x512": pf.get("SQ_INSTS_VALU_MFMA_MOPS_F32"),
                         "busy_cycles_per_frame": pf["SQ_VALU_MFMA_BUSY_CYCLES"],
                         "coexec_cycles_per_frame": pf.get("SQ_VALU_MFMA_COEXEC_CYCLES"),
                         "mfma_pipe_busy_fraction": pf["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * pf["SQ_BUSY_CU_CYCLES"])}
    if dur.get(k):
        e["kernel_ns_under_profiler"] = sum(dur[k]) / len(dur[k])
        if "GRBM_GUI_ACTIVE" in mean:
            e["clock_GHz"] = mean["GRBM_GUI_ACTIVE"] / 8.0 / e["kernel_ns_under_profiler"]
    out["kernels"][k] = e
dst = root / "profiles" / f"{tag}_sq_counters.json"
(src / f"{tag}_sq_counters.json").write_text(json.dumps(out, indent=1))   # gpurun merges gpurun_out/ back, not profiles/
dst.write_text(json.dumps(out, indent=1))
for k, e in out["kernels"].items():
    print(k[:70])
    for c, v in (e.get("per_frame") or {}).items():
        print(f"   {c:28s} {v:12.2f} per frame")
    print("   ", json.dumps(e.get("derived")), e.get("kernel_ns_under_profiler"), e.get("clock_GHz"))
