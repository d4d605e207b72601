#!/bin/bash
# usage: ab4.sh rounds "envA" "envB" "envC" ...
R=$1; shift
mkdir -p gpurun_out/ab4
for r in $(seq 1 $R); do
  i=0
  for E in "$@"; do
    i=$((i+1))
    env $E timeout -k 10 300 python bench.py --steps 60 --warmup 20 --no-cpu-baseline > gpurun_out/ab4/c${i}_$r.log 2>&1
    python - "gpurun_out/ab4/c${i}_$r.log" "$E" <<'PY'
import json, sys
lines = [x for x in open(sys.argv[1]) if x.startswith('{')]
if not lines: print(sys.argv[2], 'NO RESULT', open(sys.argv[1]).read()[-800:])
else:
    d = json.loads(lines[-1]); print(f'[{sys.argv[2]}]', d['value'], 'pairs/s', d['ms_per_step'], 'ms', flush=True)
PY
  done
done
