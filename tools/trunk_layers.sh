#!/bin/bash
# per-layer device times of one trunk under a tuning: usage trunk_layers.sh <flow|depth|depth1|vos> <tag> [VSR_TUNING list] [h w]
set -e
export TMPDIR=/tmp VSR_FLOWSD_STREAM=0
ROOT=$(pwd); OUT=$ROOT/gpurun_out/tl_$2; mkdir -p $OUT
export VSR_TUNING=$3 VSR_ROUTES_OUT=$OUT/routes.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $ROOT/tools/probe_one_trunk.py $1 2 $4 $5 > $OUT/run.log 2>&1)
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/trunk_layers.py $f $OUT/routes.txt > $ROOT/gpurun_out/tl_$2.txt
rm -rf $OUT
