#!/bin/bash
# The bench configurations BASELINE.md section 4 tabulates, one line each (run on the GPU box).
mkdir -p gpurun_out/matrix
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/matrix/$tag.err | tail -1 > gpurun_out/matrix/$tag.json
  python3 - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f'gpurun_out/matrix/{t}.json'))
    print(t, d['value'], 'pairs/s', d['ms_per_step'], 'ms', d.get('step_tflops'), 'TF/s', 'comm:', (d.get('comm') or {}).get('comm_stream_ms_per_step'), flush=True)
except Exception as e:
    print(t, 'FAILED', e); print(open(f'gpurun_out/matrix/{t}.err').read()[-800:])
PY
}
run base --steps 60 --warmup 20
run base_reducer_torch --steps 40 --warmup 10 --force-reducer
VLMO_DP_COMM=native run base_reducer_native --steps 40 --warmup 10 --force-reducer
run base_reducer_adam --steps 40 --warmup 10 --force-reducer --optimizer
run base_zero2 --steps 40 --warmup 10 --force-reducer --optimizer --zero2
run large_b32 --steps 30 --warmup 10 --preset large --batch 32
run base_full_b32 --steps 12 --warmup 4 --batch 32 --objective full --merge-passes
run large_full_zero2_b32 --steps 8 --warmup 3 --preset large --batch 32 --objective full --merge-passes --force-reducer --optimizer --zero2
