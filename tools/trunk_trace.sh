#!/bin/bash
# per-dispatch kernel trace of one trunk: usage trunk_trace.sh <flow|depth|depth1|vos|sr> <tag> [VSR_TUNING list]
set -e
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_$2; mkdir -p $OUT
export VSR_TUNING=$3
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $ROOT/tools/probe_one_trunk.py $1 3 > $OUT/run.log 2>&1)
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
case $1 in flow) MK=k_pair_sums;; vos) MK=nchw_to_nhwc;; depth*) MK=nchw_to_nhwc;; *) MK=k_head;; esac
cp $f $ROOT/gpurun_out/trace_$2.csv
python3 tools/trunk_trace.py $f $MK all > $ROOT/gpurun_out/trace_$2.txt
tail -3 $OUT/run.log
