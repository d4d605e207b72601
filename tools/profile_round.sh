#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the bench plus the rocprofv3 passes whose summaries are committed under
# profiles/.  usage: bash tools/profile_round.sh r02 [extra bench args]
# Each pass is its own run of the same command; PMC passes carry only --kernel-trace beside --pmc.
set -e
TAG=$1; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
K=10; W=3; TOTAL=$((K + W + 3))       # bench.py runs W warm-up + K timed + 3 host-timing steps
python3 bench.py --steps 40 --warmup 10 "$@" > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" >> $OUT/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
echo "stats done" >> $OUT/progress.txt
# the same step with the weight-gradient work forced onto the side stream (the default under a gradient reducer and for
# small passes): kernel durations with CU sharing between streams
VLMO_OVERLAP_WGRAD=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/serial -o s --output-format csv -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline "$@" > $OUT/serial.log 2>&1
echo "side-stream stats done" >> $OUT/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/dvae -o s --output-format csv -- python3 tools/dvae_bench.py > $OUT/dvae.log 2>&1
echo "dvae stats done" >> $OUT/progress.txt
# the data-parallel step at world 1 through either communicator, and the four-objective step (summaries only)
for m in torch native; do
  VLMO_DP_COMM=$m timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/red_$m -o s --output-format csv -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline --force-reducer > $OUT/red_$m.log 2>&1
  python3 tools/summarize_profile.py $OUT/red_$m/s_kernel_stats.csv $TOTAL > $OUT/${TAG}_reducer_${m}_summary.json
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/full -o s --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --batch 32 --objective full --merge-passes > $OUT/full.log 2>&1
python3 tools/summarize_profile.py $OUT/full/s_kernel_stats.csv 11 > $OUT/${TAG}_full_objective_summary.json
echo "reducer / full-objective stats done" >> $OUT/progress.txt
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o c --output-format csv -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline "$@" > $OUT/$C.log 2>&1
  echo "$C done" >> $OUT/progress.txt
done
MS=$(python3 -c "import json;print(json.load(open('$OUT/bench.json'))['ms_per_step'])")
python3 tools/summarize_pmc.py $TAG $TOTAL $MS --stats $OUT/stats/s_kernel_stats.csv --fetch $OUT/FETCH_SIZE/c_counter_collection.csv \
    --write $OUT/WRITE_SIZE/c_counter_collection.csv --mfma $OUT/SQ_VALU_MFMA_BUSY_CYCLES/c_counter_collection.csv --out $OUT
cp $OUT/stats/s_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 tools/summarize_profile.py $OUT/stats/s_kernel_stats.csv $TOTAL > $OUT/${TAG}_summary.json
cp $OUT/bench.json $OUT/${TAG}_bench.json
cp $OUT/serial/s_kernel_stats.csv $OUT/${TAG}_side_kernel_stats.csv
python3 tools/summarize_profile.py $OUT/serial/s_kernel_stats.csv $TOTAL > $OUT/${TAG}_side_summary.json
cp $OUT/dvae/s_kernel_stats.csv $OUT/${TAG}_dvae_kernel_stats.csv
python3 tools/summarize_profile.py $OUT/dvae/s_kernel_stats.csv 7 > $OUT/${TAG}_dvae_summary.json
grep "images/s" $OUT/dvae.log > $OUT/${TAG}_dvae_bench.txt
ls -la $OUT | tail -20
