#!/bin/bash
# HBM bytes per launch of EVERY kernel of the headline frame (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, as
# MI355X_MICROARCH.md prescribes) + a kernel-trace pass for the durations -> per-kernel HBM GB/s table.
# usage (on the GPU box, from the repo root): bash tools/frame_pmc.sh <outdir under gpurun_out>
set -e
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { name=$1; shift; (cd /tmp && timeout -k 10 400 rocprofv3 --output-format csv "$@" -d $ROOT/$OUT/$name -o run -- python3 $ROOT/tools/frame_workload.py > $ROOT/$OUT/$name.log 2>&1); echo "pass $name done"; }
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run trace --kernel-trace --stats
python3 tools/frame_pmc_parse.py $OUT
