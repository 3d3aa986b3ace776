"""Where the GPU idles inside an LM step: from a rocprofv3 --kernel-trace CSV (kernel name, start, end) take the steps
between consecutive launches of the J^T J kernel and list the gaps between kernels (who came before, who after).
usage: python3 tools/idle_gaps.py <dir with *_kernel_trace.csv> [kernel-name substring that marks a step]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
mark = sys.argv[2] if len(sys.argv) > 2 else 'gemm_tn_f64_interior_kernel<false, true>'
rows = []
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if mark in r[2]]
print('%d kernels, %d step marks' % (len(rows), len(marks)))
marks = marks[-21:]
tot_wall = tot_busy = 0.0
gaps = defaultdict(lambda: [0, 0.0])
nsteps = len(marks) - 1
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b + 1]
    wall = seg[-1][0] - seg[0][0]
    busy = 0
    end = seg[0][0]
    for (s, e, n), nxt in zip(seg[:-1], seg[1:]):
        busy += max(0, min(e, nxt[0]) - max(s, end)) if e > end else 0
        end = max(end, e)
        g = nxt[0] - end
        if g > 0:
            key = (n.split('(')[0][:60], nxt[2].split('(')[0][:60])
            gaps[key][0] += 1
            gaps[key][1] += g
    tot_wall += wall
    tot_busy += busy
print('per step: wall %.1f us, kernels %.1f us, idle %.1f us' % (tot_wall / nsteps / 1e3, tot_busy / nsteps / 1e3, (tot_wall - tot_busy) / nsteps / 1e3))
for (p, n), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%8.1f us/step  %5.1f x/step  avg %6.1f us   %s  ->  %s' % (t / nsteps / 1e3, c / nsteps, t / c / 1e3, p, n))
