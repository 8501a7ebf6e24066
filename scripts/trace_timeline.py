"""Timeline (kernels + copies, gaps) of the LAST call in a rocprofv3 trace directory written for scripts/prof_now_frame.py."""
import csv, glob, sys
d = sys.argv[1]
k = list(csv.DictReader(open(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])))
mfiles = glob.glob(d + '/**/*_memory_copy_trace.csv', recursive=True)
m = list(csv.DictReader(open(mfiles[0]))) if mfiles else []
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-44:]) for r in k]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')) for r in m]
ev.sort()
# calls are separated by the largest idle gaps: take everything after the last gap > 60 us that is followed by a H2D copy
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > 60000 and 'HOST_TO_DEVICE' in ev[i][2]: cut = i
seq = ev[cut:]
t0 = seq[0][0]; prev = None; busy = 0
for s, e, n in seq:
    print('%8.1f us  +%6.1f gap  dur %7.1f  %s' % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0, (e - s) / 1e3, n))
    prev = e; busy += (e - s) / 1e3
print('span %.1f us, busy %.1f us, %d operations' % ((seq[-1][1] - t0) / 1e3, busy, len(seq)))
