#!/bin/bash
# In-session A/B of two environments over several bench configurations.
# usage: tools/ab_args.sh "<env A>" "<env B>" rounds "<bench args 1>" "<bench args 2>" ...
A="$1"; B="$2"; R="$3"; shift 3
mkdir -p gpurun_out/abx
c=0
for ARGS in "$@"; do
  c=$((c+1))
  for r in $(seq 1 $R); do
    for v in A B; do
      if [ $v = A ]; then E="$A"; else E="$B"; fi
      env $E timeout -k 10 400 python bench.py --no-cpu-baseline $ARGS > gpurun_out/abx/c${c}_$v$r.log 2>&1
      python - "gpurun_out/abx/c${c}_$v$r.log" "$E" "$ARGS" <<'PY'
import json, sys
lines = [x for x in open(sys.argv[1]) if x.startswith('{')]
if not lines: print(sys.argv[3], sys.argv[2], 'NO RESULT', open(sys.argv[1]).read()[-600:])
else:
    d = json.loads(lines[-1]); print(f'({sys.argv[3]}) [{sys.argv[2]}]', d['value'], 'pairs/s', d['ms_per_step'], 'ms', flush=True)
PY
    done
  done
done
