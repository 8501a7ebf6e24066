"""gpurun_out/ (scratch, written by scripts/gpu_batch.sh on the GPU box) -> profiles/r01_* (tracked)."""
import csv, glob, json, os, shutil, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
newest = lambda pat: max(glob.glob(os.path.join(O, pat), recursive=True), key=os.path.getmtime)
for w, name in (('c2', 'kernel_stats_c2'), ('c5', 'kernel_stats_c5'), ('pre', 'kernel_stats_preprocess')):
    shutil.copy(newest('prof_%s/**/*kernel_stats.csv' % w), os.path.join(P, '%s_%s.csv' % (tag, name)))
for w in ('c2', 'c5'):
    src = os.path.join(O, 'bench_%s.json' % w)
    line = [l for l in open(src).read().splitlines() if l.startswith('{')][-1]
    open(os.path.join(P, '%s_bench_%s.json' % (tag, w)), 'w').write(line + '\n')
shutil.copy(os.path.join(O, 'sweep.log'), os.path.join(P, '%s_sweep.txt' % tag))
shutil.copy(os.path.join(O, 'preprocess_times.txt'), os.path.join(P, '%s_preprocess_times.txt' % tag))
lines, traffic = [], {}
note = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 '
        '(gfx950 FETCH_SIZE counts 64 B per 128-B request; calibrated in the guide for 16-B/lane streams only, the 4/8-B '
        'point loads of this kernel are uncalibrated)')
for w in ('c2', 'c5'):
    med = {}
    for ctr, d in (('FETCH_SIZE', 'pmc_fetch_%s' % w), ('WRITE_SIZE', 'pmc_write_%s' % w)):
        rows = list(csv.DictReader(open(newest(d + '/**/*counter_collection.csv'))))
        by = {}
        for r in rows:
            if r['Counter_Name'] == ctr:
                by.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
        for k, v in by.items():
            lines.append('%s %s kernel=%s dispatches=%d median_KB=%s min_KB=%s max_KB=%s' % (w, ctr, k[:70], len(v), statistics.median(v), min(v), max(v)))
            if 'ea_eval_fused' in k:
                med[ctr] = statistics.median(v)
    hb = (2 * med['FETCH_SIZE'] + med['WRITE_SIZE']) * 1024
    lines.append('%s => hbm_bytes_per_launch=%d' % (w, hb))
    traffic[w] = {'hbm_bytes_per_launch': hb, 'FETCH_SIZE_KB': med['FETCH_SIZE'], 'WRITE_SIZE_KB': med['WRITE_SIZE'], 'note': note}
open(os.path.join(P, '%s_pmc_summary.txt' % tag), 'w').write('\n'.join(lines) + '\n')
json.dump(traffic, open(os.path.join(P, 'pmc_traffic.json'), 'w'), indent=1)
print('\n'.join(lines))
