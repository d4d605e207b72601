"""`python bench.py --gpus 2` from a plain shell on the GPU box: the self-launcher starts two ranks (both on the box's one
GPU, gradient exchange over gloo: --rehearse-gloo), each runs the data-parallel step -- GradReducer with the engine's
gradient sinks, per-block ready events, pack / collective / unpack on the communication stream -- and rank 0 prints the
ONE JSON line of the bench contract with n_gpus = 2.  What the driver's N > 1 runs do, minus RCCL's links.

The file name sorts first on purpose: the children must be started from a process that has NOT initialised the GPU (this
pool refuses an exec from one that has), i.e. before any other GPU test of the session ran in this interpreter."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_rehearsal_prints_one_json_line():
    import torch
    if torch.cuda.is_initialized():
        pytest.skip('the test runner has already initialised the GPU: starting other programs from it is not allowed here')
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rehearse-gloo', '--steps', '2',
                          '--warmup', '1', '--batch', '8', '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 16 and d['config']['parallelism'] == 'dp2'
    assert d['value'] > 0 and d['unit'] == 'pairs/s'
