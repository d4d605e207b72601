"""`python bench.py --gpus N` from a plain shell starts its own ranks (bench.spawn_ranks): the environment each rank
sees (what utils/utils.py:298-334 `init_distributed_mode` reads), the single JSON line relayed from rank 0, and the
exit code when a rank fails.  The children here are tiny stand-ins, not the benchmark: no GPU is touched."""
import importlib.util
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)       # __name__ != '__main__': the launcher branch does not run on import
    return m


CHILD_OK = r'''
import json, os, sys
r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
print(json.dumps({'n_gpus': w, 'rank': r, 'port': int(os.environ['MASTER_PORT'])}))       # only rank 0's line may surface
'''
CHILD_FAIL = r'''
import os, sys, time
if os.environ['RANK'] == '1':
    sys.exit(7)
time.sleep(60)
'''


def test_spawn_ranks_relays_rank0_json(capfd):
    b = _bench()
    rc = b.spawn_ranks(3, [sys.executable, '-c', CHILD_OK], timeout=60)
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0
    assert len(out) == 1, out
    rec = json.loads(out[0])
    assert rec['n_gpus'] == 3 and rec['rank'] == 0


def test_spawn_ranks_propagates_failure_and_stops_the_rest():
    import time
    b = _bench()
    t0 = time.time()
    rc = b.spawn_ranks(2, [sys.executable, '-c', CHILD_FAIL], timeout=60)
    assert rc == 7
    assert time.time() - t0 < 30        # rank 0 (sleeping) was terminated, not waited for


def test_gpus_flag_parsing():
    b = _bench()
    assert b._wanted_gpus(['--steps', '3']) == 1
    assert b._wanted_gpus(['--gpus', '8', '--steps', '3']) == 8
    assert b._wanted_gpus(['--gpus=4']) == 4
