#!/bin/bash
# PMC passes over ONE gather-kernel layer of tools/gather_layers.py (case index $2): wave-cycle split, instruction mix, L1 / L2
# requests and latency (at most three TCP / TCC counters fit one pass on gfx950).  usage (on the GPU box, from the repo root): bash tools/gather_pmc.sh <outdir under gpurun_out> <case>
OUT=gpurun_out/$1; CASE=$2; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { name=$1; shift; (cd /tmp && timeout -k 10 200 rocprofv3 --output-format csv --pmc "$@" -d $ROOT/$OUT/$name -o run -- python3 $ROOT/tools/gather_layers.py $CASE 5 > $ROOT/$OUT/$name.log 2>&1) || echo "pass $name failed"; echo "pass $name done"; }
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run c TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
run f SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA FETCH_SIZE
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_conv_" not in k and "k_splitk" not in k: continue
        name = "finish" if "k_splitk" in k else ("patch" if "k_conv_patch" in k else "gather")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
for name, d in agg.items():
    print(name)
    for k, v in sorted(d.items()):
        print(f"   {k:40s} {v / cnt[name][k]:14.4g} per dispatch ({cnt[name][k]} dispatches)")
PY
