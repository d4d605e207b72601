"""Timeline analysis of a rocprofv3 --kernel-trace database of bench.py: per-queue busy time, GPU idle time and
the biggest gaps, for the last full step.  usage: trace_timeline.py results.db [steps_in_trace]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
names = {r[0]: r[1] for r in cur.execute(f'select id, kernel_name from {ks}')}
rows = list(cur.execute(f'select kernel_id, queue_id, stream_id, start, end from {kd} order by start'))


def short(n):
    n = re.sub(r'_ZN\d+_GLOBAL__N_1\d+', '', n)
    n = n.replace('(anonymous namespace)::', '')
    return n[:48]


# steps are delimited by the patchify kernel (first kernel of each forward)
marks = [i for i, r in enumerate(rows) if 'patchify' in names[r[0]]]
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0, t1 = step[0][3], max(r[4] for r in step)
print(f'step: {len(step)} kernels, {(rows[b][3] - t0) / 1e6:.3f} ms start-to-start, {(t1 - t0) / 1e6:.3f} ms span')
queues = {}
for r in step:
    queues.setdefault(r[1], []).append(r)
for q, rs in queues.items():
    busy = sum(r[4] - r[3] for r in rs)
    print(f'queue {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms')
# union busy / idle
ev = sorted((r[3], r[4]) for r in step)
cur_e = ev[0][0]
idle = 0
gaps = []
for s, e in ev:
    if s > cur_e:
        idle += s - cur_e
        gaps.append((s - cur_e, s))
    cur_e = max(cur_e, e)
print(f'GPU idle inside the step: {idle / 1e6:.3f} ms in {len(gaps)} gaps; largest: {[round(g[0] / 1e3, 1) for g in sorted(gaps, reverse=True)[:8]]} us')
# main-queue gaps (time between consecutive kernels of the busiest queue)
mainq = max(queues, key=lambda q: len(queues[q]))
rs = queues[mainq]
gsum = sum(max(0, rs[i + 1][3] - rs[i][4]) for i in range(len(rs) - 1))
print(f'main queue {mainq}: sum of inter-kernel gaps {gsum / 1e6:.3f} ms')
# overlap: time during which >= 2 kernels run
pts = []
for s, e in ev:
    pts += [(s, 1), (e, -1)]
pts.sort()
lvl = 0
last = pts[0][0]
ov = 0
for t, d in pts:
    if lvl >= 2:
        ov += t - last
    lvl += d
    last = t
print(f'time with >= 2 kernels resident: {ov / 1e6:.3f} ms')
if len(sys.argv) > 2:
    for r in rs[:int(sys.argv[2])]:
        print(f'{(r[3] - t0) / 1e3:9.1f} {(r[4] - r[3]) / 1e3:8.1f}  {short(names[r[0]])}')
print('--- gaps > 8 us on the main queue (offset_us gap_us  before -> after)')
for i in range(len(rs) - 1):
    g = rs[i + 1][3] - rs[i][4]
    if g > 8000:
        print(f'{(rs[i][4] - t0) / 1e3:9.1f} {g / 1e3:7.1f}  {short(names[rs[i][0]])} -> {short(names[rs[i + 1][0]])}')
