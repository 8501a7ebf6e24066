"""Per-iteration timeline of the device-resident LM loop from a rocprofv3 --kernel-trace of `scripts/prof_run.py lm`:
durations of the evaluation and step kernels and the gaps between them.   python scripts/lm_trace_gaps.py <trace.csv>"""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = [(s, e, "eval" if "ea_eval_fused" in n else ("step" if "ea_lm_step" in n else "other")) for s, e, n in ks]
dur = {"eval": [], "step": []}
gap = {"eval->step": [], "step->eval": []}
for (s0, e0, k0), (s1, e1, k1) in zip(ev, ev[1:]):
    if k0 in dur and (k1 in dur) and k0 != k1 and s1 - e0 < 20000:
        gap[k0 + "->" + k1].append(s1 - e0)
for s, e, k in ev:
    if k in dur:
        dur[k].append(e - s)
for k, v in dur.items():
    if v:
        print("%-5s kernels %5d: duration median %.0f ns, p10 %.0f, p90 %.0f" % (k, len(v), statistics.median(v), sorted(v)[len(v) // 10], sorted(v)[len(v) * 9 // 10]))
for k, v in gap.items():
    if v:
        print("gap %-11s %5d: median %.0f ns, p10 %.0f, p90 %.0f" % (k, len(v), statistics.median(v), sorted(v)[len(v) // 10], sorted(v)[len(v) * 9 // 10]))
# iteration period: start of eval k to start of eval k+1 inside a solve
st = [s for s, e, k in ev if k == "eval"]
per = [b - a for a, b in zip(st, st[1:]) if b - a < 30000]
if per:
    print("evaluation start to next evaluation start: median %.0f ns (%d pairs)" % (statistics.median(per), len(per)))
