"""rocprofv3 --kernel-trace CSV of a device-resident solve -> kernel durations and the gaps between consecutive kernels.
usage: python scripts/lm_trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, statistics
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=lambda p: __import__('os').path.getmtime(p))
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ev = [(r['Kernel_Name'].split('(')[0].replace('void ea::', '').replace('ea::', '')[:40], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
dur, gap = {}, {}
for i, (n, s, e) in enumerate(ev):
    dur.setdefault(n, []).append(e - s)
    if i + 1 < len(ev):
        g = ev[i + 1][1] - e
        if g < 20000:
            gap.setdefault(n + ' -> ' + ev[i + 1][0], []).append(g)
for n, v in dur.items():
    if len(v) >= 5: print('%-42s n=%4d  median %6.0f ns  (p10 %6.0f, p90 %6.0f)' % (n, len(v), statistics.median(v), sorted(v)[len(v) // 10], sorted(v)[len(v) * 9 // 10]))
for n, v in gap.items():
    if len(v) >= 5: print('gap %-70s n=%4d  median %6.0f ns' % (n, len(v), statistics.median(v)))
