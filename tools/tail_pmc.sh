#!/bin/bash
# PMC passes over the fused tail k_tail3 (full frames and the decimated pass, folded compress_out; tools/tail_time.py as the workload):
# instruction mix per wave, MFMA-pipe busy cycles, wave-cycle split, HBM bytes.  One rocprofv3 --pmc run per counter group.
# usage (GPU box, repo root): bash tools/tail_pmc.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { name=$1; shift; (cd /tmp && timeout -k 10 280 rocprofv3 --output-format csv --pmc "$@" -d $ROOT/$OUT/$name -o run -- python3 $ROOT/tools/tail_time.py > $ROOT/$OUT/$name.log 2>&1); echo "pass $name done"; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run d FETCH_SIZE
run e WRITE_SIZE
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_tail3" not in k: continue
        # template arguments <ALLMAX, DEC, FOLD, DIAG>: ILb1ELb0ELb1E.. = full frame folded, ILb1ELb1ELb1E.. = decimated folded, ..ELb0E = not folded
        tag = ("decimated" if "ILb1ELb1E" in k or "ILb0ELb1E" in k else "full") + (" folded" if "ELb1ELi" in k else " unfolded")
        vals[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
        vals[tag]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for tag, d in vals.items():
    med = {k: sorted(v)[len(v) // 2] for k, v in d.items()}
    waves = med.get("SQ_WAVES", 992.0)
    steps = 543.0   # LR rows + 3 steps of a 540-row march
    s = {"dispatch_ns_under_pmc": med.pop("_dur_ns"), "waves": waves, "counters": med}
    if "SQ_INSTS_MFMA" in med:
        s["insts_per_wave_per_step"] = {k: round(med[n] / waves / steps, 1) for k, n in (("mfma", "SQ_INSTS_MFMA"), ("valu_incl_mfma", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR")) if n in med}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in med:
        s["mfma_busy_cycles_per_simd_per_step"] = round(med["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4) / steps, 1)
    if "SQ_WAVE_CYCLES" in med:
        wc = med["SQ_WAVE_CYCLES"]
        s["cycles_per_wave_per_step"] = round(4 * wc / waves / steps, 1)
        s["wave_cycle_split"] = {k: round(med[n] / wc, 3) for k, n in (("active_issue", "SQ_ACTIVE_INST_ANY"), ("wait_any(waitcnt/barrier)", "SQ_WAIT_ANY"), ("wait_inst_any(issue stall)", "SQ_WAIT_INST_ANY")) if n in med}
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        s["hbm"] = {"FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"], "traffic_bytes_per_launch": (2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024,
                    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request (profiles/r03_fetch_calibration.txt) -> doubled; WRITE_SIZE exact"}
    res[tag] = s
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({t: {k: v for k, v in s.items() if k != "counters"} for t, s in res.items()}, indent=1))
PY
