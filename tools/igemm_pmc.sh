#!/bin/bash
# PMC passes over FlowNet2 alone (tools/probe_one_trunk.py flow): L2 request / hit / miss counts and wave-cycle split of
# the gather kernels.  usage (on the GPU box, from the repo root): bash tools/igemm_pmc.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { name=$1; shift; (cd /tmp && timeout -k 10 250 rocprofv3 --output-format csv --pmc "$@" -d $ROOT/$OUT/$name -o run -- python3 $ROOT/tools/probe_one_trunk.py flow 3 > $ROOT/$OUT/$name.log 2>&1) || echo "pass $name failed"; echo "pass $name done"; }
run a TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
run c FETCH_SIZE
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_conv_igemm_d" not in k: continue
        name = "k_conv_igemm_d<128>" if "Li128E" in k or "<128" in k else ("k_conv_igemm_d<64>" if "Li64E" in k or "<64" in k else "k_conv_igemm_d<16/32>")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("TCC_REQ_sum", "SQ_WAVE_CYCLES", "FETCH_SIZE"): cnt[(name, r["Counter_Name"])] += 1
for name, d in agg.items():
    print(name, {k: f"{v:.4g}" for k, v in sorted(d.items())}, {k[1]: v for k, v in cnt.items() if k[0] == name})
PY
