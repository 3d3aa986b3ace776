"""Developer tool: largest kernels and largest idle gaps in a rocprofv3 --kernel-trace csv (which stall is it: a kernel that
took long, or a hole between kernels?)  usage: dbg_find_gaps.py <kernel_trace.csv>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows))
print('%d kernels, span %.1f ms' % (len(ev), (ev[-1][1] - ev[0][0]) / 1e6))
print('longest kernels:')
for s, e, n in sorted(ev, key=lambda t: t[0] - t[1])[:8]:
    print('   %.3f ms  %s  (at %.1f ms)' % ((e - s) / 1e6, n, (s - ev[0][0]) / 1e6))
gaps = []
end = ev[0][1]
prev = ev[0][2]
for s, e, n in ev[1:]:
    if s > end:
        gaps.append((s - end, end, prev, n))
    if e > end:
        end, prev = e, n
print('largest idle gaps (no kernel running):')
for g, at, a, b in sorted(gaps, reverse=True)[:12]:
    print('   %.3f ms at %.1f ms, between %s and %s' % (g / 1e6, (at - ev[0][0]) / 1e6, a, b))
