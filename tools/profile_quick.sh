#!/bin/bash
# Quick kernel-time profile of the default bench step (weight gradients on the stream the engine's rule picks: the main
# stream for VLMo-Base at 64 pairs) and of the same step with the side stream forced, on the GPU box.
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
VLMO_OVERLAP_WGRAD=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/serial -o s --output-format csv -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline "$@" > $OUT/serial.log 2>&1
python3 tools/summarize_profile.py $OUT/serial/s_kernel_stats.csv $TOTAL > $OUT/${TAG}_side_summary.json
cp $OUT/stats/s_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
cp $OUT/serial/s_kernel_stats.csv $OUT/${TAG}_side_kernel_stats.csv
rm -rf $OUT/stats $OUT/serial
