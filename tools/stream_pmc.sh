#!/bin/bash
# PMC passes over the 1x1 streaming kernel (tools/stream_exp.py): HBM bytes and wave-cycle split.
# usage (on the GPU box, from the repo root): bash tools/stream_pmc.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { name=$1; shift; (cd /tmp && timeout -k 10 200 rocprofv3 --output-format csv --pmc "$@" -d $ROOT/$OUT/$name -o run -- python3 $ROOT/tools/stream_exp.py one > $ROOT/$OUT/$name.log 2>&1); echo "pass $name done"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run d FETCH_SIZE
run e WRITE_SIZE
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
vals = {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_conv1x1_stream" not in r["Kernel_Name"]: continue
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print(f"{k:24s} median {sorted(v)[len(v)//2]:.4g}  (n={len(v)})")
PY
