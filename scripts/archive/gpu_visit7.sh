#!/bin/bash
# the riding fold: its GPU tests, the whole GPU suite, default / driver-shaped / serial / C5 bench lines
cd /tmp && export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pipelined.py -x -q > $O/pytest_pipelined.log 2>&1; rc=$?; echo "pipelined_exit=$rc"; tail -5 $O/pytest_pipelined.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest_exit=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 400 python bench.py > $O/b7.json 2> $O/b7.err; echo "bench=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/b7_s20.json 2>> $O/b7.err; echo "bench_s20=$?"
timeout -k 10 300 python bench.py --serial-steps --no-extras --no-cpu-baseline > $O/b7_serial.json 2>> $O/b7.err; echo "bench_serial=$?"
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --no-extras > $O/b7_c5.json 2>> $O/b7.err; echo "bench_c5=$?"
python - <<'PY'
import json
for f in ("b7", "b7_s20", "b7_serial", "b7_c5"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "value %.3e us/step %.3f | %s kernel_ms %.5f b2b %.5f fold %.5f frac %.3f b2b %.3f | serial graph %.5f | %s" % (
            d["value"], d["ms_per_step"] * 1e3, r["kernel"], r["kernel_ms"], r["kernel_ms_back_to_back"], r["fold_kernel_ms"], r["frac"],
            r["frac_back_to_back"], r["step_ms_events_serial_graph"], d["config"]["timed_region"]))
    except Exception as e:
        print(f, "unreadable", repr(e))
PY
tail -5 $O/b7.err
