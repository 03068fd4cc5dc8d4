#!/bin/bash
# FETCH_SIZE calibration (tools/microbench/fetch_calib.hip): prints counter value per kernel next to the bytes actually read
set -e
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/fetch_calib; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -w tools/microbench/fetch_calib.hip -o /tmp/fetch_calib
(cd /tmp && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o run -- /tmp/fetch_calib > $OUT/run.log 2>&1)
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = {}
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE": agg.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
print("# r03 FETCH_SIZE calibration on gfx950 (tools/microbench/fetch_calib.hip): 256 MiB read once per launch")
for k, v in sorted(agg.items()):
    kb = sum(v) / len(v)
    print(f"{k:40s} FETCH_SIZE {kb:12.1f} KB = {kb * 1024 / (256 << 20):.3f} x the bytes read  ({len(v)} launches)")
PY
