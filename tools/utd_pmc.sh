#!/bin/bash
# PMC passes over the fused-stage kernel alone (tools/utd_microbench.py), one rocprofv3 --pmc run per counter group.
# usage (on the GPU box, from the repo root): [UTD_N=5 UTD_ROWS=327] bash tools/utd_pmc.sh <outdir under gpurun_out> [hbm] [s2]
#   UTD_N: planes per launch (default 8); UTD_ROWS: LR rows one workgroup marches (540; 327 in the flat split of 5 planes over 256 workgroups)
#   s2: the x2 stage kernel k_utd_s2 at 8 x 1080 x 1920 (tools/utd_s2_microbench.py) instead of k_utd3 at 8 x 540 x 960
set -e
OUT=gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
if [ "$3" = "s2" ]; then WORK=("$ROOT/tools/utd_s2_microbench.py" 1080 1920 5); ROWS=1080; else WORK=("$ROOT/tools/utd_microbench.py" 540 960 5 "${UTD_VARIANT:-uniform}"); ROWS=${UTD_ROWS:-540}; fi   # (UTD_VARIANT may hold spaces: "k_utd4 post")
run() { name=$1; shift; (cd /tmp && timeout -k 10 280 rocprofv3 --output-format csv --pmc "$@" -d $ROOT/$OUT/$name -o run -- python3 "${WORK[@]}" > $ROOT/$OUT/$name.log 2>&1); echo "pass $name done"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA
if [ "$2" = "hbm" ]; then run d FETCH_SIZE; run e WRITE_SIZE; fi
python3 tools/utd_pmc_parse.py $OUT $ROWS
