#!/bin/bash
# end-of-round visit: GPU tests, smoke, default + driver-shaped bench, rocprofv3 kernel stats of the bench command
cd /tmp && export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest_exit=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_final_s20.json 2>> $O/bench_final.err; echo "bench_s20=$?"
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --no-extras > $O/bench_final_c5.json 2>> $O/bench_final.err; echo "bench_c5=$?"
for w in c2 c5; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02g_prof_$w -o prof -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-extras > $O/r02g_prof_$w.log 2>&1); echo "prof_$w=$?"
done
python - <<'PY'
import json
for f in ("bench_final", "bench_final_s20", "bench_final_c5"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value %.3e us/step %.3f | kernel_ms %.5f b2b %.5f fold %.5f frac %.3f b2b %.3f | sec %s" % (
        d["value"], d["ms_per_step"] * 1e3, r["kernel_ms"], r["kernel_ms_back_to_back"], r["fold_kernel_ms"], r["frac"], r["frac_back_to_back"],
        {k: round(v, 3) for k, v in (r.get("secondary") or {}).items() if k.startswith("frac")}))
PY
cat $O/r02g_prof_c2/prof_kernel_stats.csv | head -4
