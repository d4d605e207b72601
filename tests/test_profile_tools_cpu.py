"""The arithmetic of tools/summarize_pmc.py (what bench.py's roofline.traffic / hbm_gbs / mfma_busy are read from):
rocprofv3 counter CSVs -> per-launch HBM bytes (FETCH_SIZE in KiB, doubled on gfx950; WRITE_SIZE in KiB) and the
MFMA-busy fraction (busy cycles / (1024 SIMDs x time x 2.4 GHz)), on a synthetic two-kernel trace."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _counter_csv(path, name, rows):
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kernel_Name', 'Counter_Name', 'Counter_Value'])
        for kernel, value in rows:
            w.writerow([kernel, name, value])


def test_summarize_pmc_units_and_corrections(tmp_path):
    steps, ms = 2, 10.0
    # kernel A: 4 launches (2 per step), kernel B: 2 launches
    _counter_csv(tmp_path / 'f.csv', 'FETCH_SIZE', [('A.kd', 1000)] * 4 + [('B.kd', 500)] * 2)
    _counter_csv(tmp_path / 'w.csv', 'WRITE_SIZE', [('A.kd', 100)] * 4 + [('B.kd', 50)] * 2)
    _counter_csv(tmp_path / 'm.csv', 'SQ_VALU_MFMA_BUSY_CYCLES', [('A.kd', 2.4e6)] * 4 + [('B.kd', 0)] * 2)
    with open(tmp_path / 's.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
        w.writerow(['A.kd', 4, 4000, 1000.0, 80, 1, 1, 0])
        w.writerow(['B.kd', 2, 1000, 500.0, 20, 1, 1, 0])
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'summarize_pmc.py'), 'rXX', str(steps), str(ms),
                        '--stats', str(tmp_path / 's.csv'), '--fetch', str(tmp_path / 'f.csv'),
                        '--write', str(tmp_path / 'w.csv'), '--mfma', str(tmp_path / 'm.csv'), '--out', str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.load(open(tmp_path / 'rXX_traffic.json'))
    a, b = out['kernels']['A'], out['kernels']['B']
    assert a['launches'] == 4 and b['launches'] == 2
    assert a['fetch_bytes_per_launch'] == 1000 * 1024 * 2.0          # KiB, and 128-B requests tallied at 64 B on gfx950
    assert a['write_bytes_per_launch'] == 100 * 1024
    assert b['fetch_bytes_per_launch'] == 500 * 1024 * 2.0
    # MFMA busy of kernel A: 2.4e6 busy cycles / (1024 SIMDs * 1 us * 2.4e9 Hz) = 2.4e6 / 2.4576e6
    assert abs(a['mfma_busy'] - 2.4e6 / (1024 * 1000e-9 * 2.4e9)) < 1e-9
    assert abs(a['hbm_gbs'] - (1000 * 2048 + 100 * 1024) / 1000.0) < 1e-9          # bytes per ns = GB/s
    step = out['step']
    tot_f = (4 * 1000 + 2 * 500) * 1024 * 2.0
    tot_w = (4 * 100 + 2 * 50) * 1024
    assert abs(step['fetch_gb'] - tot_f / steps / 1e9) < 1e-12 and abs(step['write_gb'] - tot_w / steps / 1e9) < 1e-12
    assert step['hbm_gbs'] == round((tot_f + tot_w) / steps / (ms * 1e-3) / 1e9, 1)
    assert step['mfma_busy'] == round(4 * 2.4e6 / steps / (1024 * ms * 1e-3 * 2.4e9), 4)
