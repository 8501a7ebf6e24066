#!/bin/bash
# round-2 second session: GPU suite, bench lines (default / driver shape / serial / C5), rocprofv3 kernel stats of the bench
# commands, PMC traffic of the bench's own kernel and of the materialised-mode kernel
cd /tmp && export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest_exit=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py > $O/b8.json 2> $O/b8.err; echo "bench=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/b8_s20.json 2>> $O/b8.err; echo "bench_s20=$?"
timeout -k 10 300 python bench.py --serial-steps --no-extras --no-cpu-baseline > $O/b8_serial.json 2>> $O/b8.err; echo "bench_serial=$?"
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --no-extras > $O/b8_c5.json 2>> $O/b8.err; echo "bench_c5=$?"
for w in c2 c5; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02h_prof_$w -o prof -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-extras > $O/r02h_prof_$w.log 2>&1); echo "prof_$w=$?"
done
bash $R/scripts/pmc_traffic_bench.sh > $O/pmc_traffic_bench.txt 2>&1; echo "pmc_bench=$?"
bash $R/scripts/pmc_rows.sh > $O/pmc_rows.txt 2>&1; echo "pmc_rows=$?"
python - <<'PY'
import json
for f in ("b8", "b8_s20", "b8_serial", "b8_c5"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]; m = d.get("materialised_mode") or {}
        print(f, "value %.3e us/step %.3f | %s kernel_ms %.5f b2b %.5f frac %.3f b2b %.3f | serial graph %.5f | rows kernel_ms %s frac %s" % (
            d["value"], d["ms_per_step"] * 1e3, r["kernel"], r["kernel_ms"], r["kernel_ms_back_to_back"], r["frac"],
            r["frac_back_to_back"], r["step_ms_events_serial_graph"], m.get("kernel_ms"), m.get("frac")))
    except Exception as e:
        print(f, "unreadable", repr(e))
PY
head -4 $O/r02h_prof_c2/prof_kernel_stats.csv; head -4 $O/r02h_prof_c5/prof_kernel_stats.csv
tail -12 $O/pmc_traffic_bench.txt; tail -25 $O/pmc_rows.txt
