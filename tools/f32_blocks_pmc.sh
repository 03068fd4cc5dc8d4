#!/bin/bash
# PMC passes over the two float32 block kernels of the SR net (tools/f32_blocks_time.py as the workload); usage: bash tools/f32_blocks_pmc.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { name=$1; shift; (cd /tmp && timeout -k 10 280 rocprofv3 --output-format csv --pmc "$@" -d $ROOT/$OUT/$name -o run -- python3 $ROOT/tools/f32_blocks_time.py > $ROOT/$OUT/$name.log 2>&1); echo "pass $name done"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = "deconv" if "k_deconv_mfma_sh" in r["Kernel_Name"] else ("conv" if "k_conv_mfma_sh" in r["Kernel_Name"] else None)
        if k: res.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summ = {}
for k, c in res.items():
    m = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    d = dict(counters=m)
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        d["wave_cycle_split"] = {a: round(m[b] / wc, 3) for a, b in (("issuing", "SQ_ACTIVE_INST_ANY"), ("waiting (waitcnt)", "SQ_WAIT_ANY"), ("waiting for an issue slot", "SQ_WAIT_INST_ANY"), ("... of which LDS", "SQ_WAIT_INST_LDS")) if b in m}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        d["mfma_busy_fraction"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8), 3)
    if "SQ_INSTS_MFMA" in m:
        d["per_mfma"] = {a: round(m[b] / m["SQ_INSTS_MFMA"], 2) for a, b in (("valu (incl. mfma)", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("vmem reads", "SQ_INSTS_VMEM_RD")) if b in m}
    summ[k] = d
json.dump(summ, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} for k, v in summ.items()}, indent=1))
PY
