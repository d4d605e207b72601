#!/bin/bash
# Quick kernel-time profile of the default bench step (overlapped and one-stream) on the GPU box.
# usage: bash tools/profile_quick.sh <tag> [extra bench args]
set -e
TAG=$1; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
K=10; W=3; TOTAL=$((K + W + 3))
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
python3 tools/summarize_profile.py $OUT/stats/s_kernel_stats.csv $TOTAL > $OUT/${TAG}_summary.json
VLMO_OVERLAP_WGRAD=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/serial -o s --output-format csv -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline "$@" > $OUT/serial.log 2>&1
python3 tools/summarize_profile.py $OUT/serial/s_kernel_stats.csv $TOTAL > $OUT/${TAG}_serial_summary.json
cp $OUT/stats/s_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
cp $OUT/serial/s_kernel_stats.csv $OUT/${TAG}_serial_kernel_stats.csv
rm -rf $OUT/stats $OUT/serial
