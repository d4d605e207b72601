"""Turn the rocprofv3 passes of one `bench.py` command into the files kept under profiles/ (what bench.py's
`roofline.traffic`, `hbm_gbs` and `mfma_busy` read).

usage: summarize_pmc.py <round tag, e.g. r02> <steps in each profiled run> <ms_per_step of the UNPROFILED run>
           --stats <kernel_stats.csv> --fetch <counter_collection.csv> --write <counter_collection.csv>
           --mfma <counter_collection.csv> [--out profiles/]

Passes (each its own run of the same command, `--kernel-trace` only beside `--pmc`, as the MI355X guide prescribes):
    rocprofv3 --kernel-trace --stats                         -> kernel_stats.csv
    rocprofv3 --kernel-trace --pmc FETCH_SIZE                -> HBM read bytes  (KiB; x2 on gfx950: 128-B requests are
                                                               tallied at 64 B, MI355X_MICROARCH.md section HBM)
    rocprofv3 --kernel-trace --pmc WRITE_SIZE                -> HBM write bytes (KiB)
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES  -> cycles with an MFMA executing, summed over SIMDs
FETCH_SIZE / WRITE_SIZE count fabric requests of the L2s: Infinity-Cache hits are included, so `hbm_gbs` is an UPPER bound
of the DRAM traffic.  `mfma_busy` = sum of MFMA-busy cycles / (1024 SIMDs x step time x 2.4 GHz): the fraction of the
chip's peak-clock matrix-pipe cycles that issued an MFMA (the chip holds ~2.0 GHz under this load, so 83 % is the most
this figure can reach)."""
import argparse
import collections
import csv
import json
import os
import re

PEAK_CLOCK_HZ = 2.4e9
SIMDS = 1024


def short(n):
    n = re.sub(r'_ZN\d+_GLOBAL__N_1\d*', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\.kd$', '', n)
    return n[:110]


def counter(path, name):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == name:
            a = agg[short(r['Kernel_Name'])]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('tag')
    ap.add_argument('steps', type=int)
    ap.add_argument('ms_per_step', type=float)
    ap.add_argument('--stats')
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--mfma')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles'))
    a = ap.parse_args()
    steps = a.steps
    out = {'note': __doc__.split('\n\n')[2] if False else
           'per-launch HBM-side bytes and MFMA-busy cycles of every kernel of `python bench.py` (VLMo-Base, B=64), from '
           'separate rocprofv3 --pmc passes; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); Infinity-Cache '
           'hits are included in both byte counters',
           'steps_profiled': steps, 'ms_per_step_unprofiled': a.ms_per_step, 'kernels': {}}
    fetch = counter(a.fetch, 'FETCH_SIZE') if a.fetch else {}
    write = counter(a.write, 'WRITE_SIZE') if a.write else {}
    mfma = counter(a.mfma, 'SQ_VALU_MFMA_BUSY_CYCLES') if a.mfma else {}
    names = set(fetch) | set(write) | set(mfma)
    tot_f = tot_w = tot_m = 0.0
    for k in sorted(names):
        e = {}
        if k in fetch:
            e['fetch_bytes_per_launch'] = fetch[k][0] * 1024 * 2.0 / fetch[k][1]
            e['launches'] = fetch[k][1]
            tot_f += fetch[k][0] * 1024 * 2.0
        if k in write:
            e['write_bytes_per_launch'] = write[k][0] * 1024 / write[k][1]
            tot_w += write[k][0] * 1024
        if k in mfma:
            e['mfma_busy_cycles_per_launch'] = mfma[k][0] / mfma[k][1]
            tot_m += mfma[k][0]
        out['kernels'][k] = e
    if a.stats:
        rows = list(csv.DictReader(open(a.stats)))
        dur = {short(r['Name']): float(r['AverageNs']) for r in rows}
        for k, e in out['kernels'].items():
            if k in dur:
                e['avg_us'] = dur[k] / 1e3
                if 'mfma_busy_cycles_per_launch' in e:
                    e['mfma_busy'] = e['mfma_busy_cycles_per_launch'] / (SIMDS * dur[k] * 1e-9 * PEAK_CLOCK_HZ)
                if 'fetch_bytes_per_launch' in e and 'write_bytes_per_launch' in e:
                    e['hbm_gbs'] = (e['fetch_bytes_per_launch'] + e['write_bytes_per_launch']) / dur[k]
    step_s = a.ms_per_step * 1e-3
    out['step'] = {
        'fetch_gb': tot_f / steps / 1e9, 'write_gb': tot_w / steps / 1e9,
        'hbm_gbs': round((tot_f + tot_w) / steps / step_s / 1e9, 1),
        'mfma_busy': round(tot_m / steps / (SIMDS * step_s * PEAK_CLOCK_HZ), 4),
        'mfma_busy_definition': 'sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x step time x 2.4 GHz)',
    }
    path = os.path.join(a.out, f'{a.tag}_traffic.json')
    json.dump(out, open(path, 'w'), indent=1)
    print('wrote', path, json.dumps(out['step']))


if __name__ == '__main__':
    main()
