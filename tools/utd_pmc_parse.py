"""Summarises the rocprofv3 --pmc passes of tools/utd_pmc.sh for the fused-stage kernel -> <outdir>/summary.json"""
import csv, glob, json, os, sys
out = sys.argv[1]
vals, meta = {}, {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "k_utd" not in r["Kernel_Name"]:
                continue
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            meta["kernel"] = r["Kernel_Name"][:60]; meta["wg"] = int(r["Workgroup_Size"]); meta["grid"] = int(r["Grid_Size"])
            meta.setdefault("dur", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
med = {k: sorted(v)[len(v) // 2] for k, v in vals.items()}
waves = meta["grid"] / 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 540   # LR rows one workgroup marches
s = {"kernel": meta.get("kernel"), "workgroup": meta.get("wg"), "waves": waves,
     "dispatch_ns_under_pmc": sorted(meta["dur"])[len(meta["dur"]) // 2], "counters": med}
if "SQ_WAVE_CYCLES" in med:
    wc = med["SQ_WAVE_CYCLES"]
    s["wave_cycle_split"] = {k: med[n] / wc for k, n in (("active_issue", "SQ_ACTIVE_INST_ANY"), ("wait_any(waitcnt/barrier)", "SQ_WAIT_ANY"),
                                                          ("wait_inst_any(issue stall)", "SQ_WAIT_INST_ANY"), ("wait_inst_lds", "SQ_WAIT_INST_LDS")) if n in med}
    s["quad_cycles_per_wave_per_row"] = wc / waves / steps
if "SQ_INSTS_MFMA" in med:
    s["insts_per_wave_per_row"] = {k: med[n] / waves / steps for k, n in (("mfma", "SQ_INSTS_MFMA"), ("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS")) if n in med}
if "SQ_VALU_MFMA_BUSY_CYCLES" in med and "SQ_BUSY_CYCLES" in med:
    s["mfma_busy_cycles_per_simd_per_row"] = med["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4) / steps
if "GRBM_GUI_ACTIVE" in med:
    s["gui_active_cycles_per_xcd"] = med["GRBM_GUI_ACTIVE"] / 8
if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
    s["hbm"] = {"FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"],
                "traffic_bytes_per_launch": (2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024,
                "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of wide streaming reads -> doubled; WRITE_SIZE exact"}
json.dump(s, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(s, indent=1))
