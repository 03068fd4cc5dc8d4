#!/bin/bash
# Instruction mix of one layer of tools/gather_layers.py (case $2, VSR_TUNING from the environment): one PMC pass.
# usage (GPU box, repo root): VSR_TUNING=0 bash tools/pmc_insts.sh <outdir under gpurun_out> <case>
OUT=gpurun_out/$1; CASE=$2; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $ROOT/$OUT/b -o run -- python3 $ROOT/tools/gather_layers.py $CASE 5 > $ROOT/$OUT/b.log 2>&1) || echo "pass failed"
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_conv_" not in k: continue
        agg[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k[:60]][r["Counter_Name"]] += 1
for name, d in agg.items():
    w = d["SQ_WAVES"] / cnt[name]["SQ_WAVES"]
    print(name, f"waves {w:.0f}; per wave:", {k: round(v / cnt[name][k] / w, 1) for k, v in sorted(d.items()) if k != "SQ_WAVES"})
PY
