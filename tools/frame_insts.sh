#!/bin/bash
# Instruction mix of EVERY kernel of the headline frame (one rocprofv3 --pmc pass over tools/frame_workload.py): per wave VALU / SALU /
# MFMA / VMEM / LDS instruction counts and the VALU-busy share of the wave cycles -- which kernels issue many instructions per MFMA.
# usage (GPU box, repo root): bash tools/frame_insts.sh <outdir under gpurun_out> [h w scale]
OUT=gpurun_out/$1; shift; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && timeout -k 10 500 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $ROOT/$OUT/i -o run -- python3 $ROOT/tools/frame_workload.py "$@" > $ROOT/$OUT/i.log 2>&1) || echo "pass failed"
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0][:58]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])
print(f"{'kernel':58s} {'disp':>5s} {'waves':>9s} {'VALU/w':>8s} {'SALU/w':>8s} {'MFMA/w':>8s} {'VMEM/w':>7s} {'LDS/w':>7s} {'VALU/MFMA':>9s} {'VALUbusy':>8s} {'wavecyc%':>8s}")
tot = sum(d["SQ_WAVE_CYCLES"] for _, d in rows)
for k, d in rows[:45]:
    w = d["SQ_WAVES"] or 1
    print(f"{k:58s} {cnt[k]:5d} {w:9.0f} {d['SQ_INSTS_VALU']/w:8.0f} {d['SQ_INSTS_SALU']/w:8.0f} {d['SQ_INSTS_MFMA']/w:8.0f} {d['SQ_INSTS_VMEM_RD']/w:7.0f} {d['SQ_INSTS_LDS']/w:7.0f} "
          f"{(d['SQ_INSTS_VALU']/d['SQ_INSTS_MFMA'] if d['SQ_INSTS_MFMA'] else 0):9.1f} {100*d['SQ_ACTIVE_INST_VALU']/d['SQ_WAVE_CYCLES']:7.1f}% {100*d['SQ_WAVE_CYCLES']/tot:7.1f}%")
PY
