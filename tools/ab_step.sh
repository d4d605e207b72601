#!/bin/bash
# In-step A/B inside ONE gpurun session: bench.py alternately with two environments.
# usage: tools/ab_step.sh "<env A>" "<env B>" [rounds] [extra bench args]
A="$1"; B="$2"; R="${3:-2}"; shift 3 || true
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    env $E timeout -k 10 300 python bench.py --steps 60 --warmup 20 "$@" > gpurun_out/ab/$v$r.log 2>&1
    python - "$v$r" "$E" <<'PY'
import json, sys
tag, env = sys.argv[1], sys.argv[2]
lines = [x for x in open(f'gpurun_out/ab/{tag}.log') if x.startswith('{')]
if not lines:
    print(tag, env, 'NO RESULT'); print(open(f'gpurun_out/ab/{tag}.log').read()[-1500:])
else:
    d = json.loads(lines[-1]); print(tag, f'[{env}]', d['value'], 'pairs/s', d['ms_per_step'], 'ms', flush=True)
PY
  done
done
