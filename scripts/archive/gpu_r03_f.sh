#!/bin/bash
# round 3, sixth GPU visit: pose-batched ea_batch_eval_poses -- suite, bench lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03f_tests.txt 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03f_tests.txt
tail -5 gpurun_out/r03f_tests.txt
for args in "--steps 20 --warmup 5" "" "--workload c5 --no-extras --no-cpu-baseline"; do
  tag=$(echo "$args" | tr -d ' -' | cut -c1-24); tag=${tag:-default}
  timeout -k 10 600 python bench.py $args > gpurun_out/r03f_bench_$tag.json 2> gpurun_out/r03f_bench_$tag.err; echo "bench [$args] rc=$?"
  python - gpurun_out/r03f_bench_$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("  value %.4g ms/step %.5f serial-dependent %.4g single_eval %.4f eval_poses_call %s" % (d["value"], d["ms_per_step"], d["value_serial_dependent_steps"], d["single_eval_call_ms"], d["eval_poses_call_ms"]))
print("  kernel_ms %.5f launches %s poses/launch %s frac %.3f floor %s ceiling %s" % (r["kernel_ms"], r["evaluation_launches_in_timed_region"], r["poses_per_launch"], r["frac"], r["launch_floor_ms"], r["frac_ceiling_at_floor"]))
print("  one pose per launch:", {k: r["one_pose_per_launch"][k] for k in ("kernel_ms_back_to_back", "frac", "launch_floor_ms", "frac_ceiling_at_floor")})
print("  secondary", r["secondary"] and r["secondary"]["frac_of_measured_ceiling"], "lm it/s", d.get("lm_iters_per_s_at_1e5_pts"))
PY
done
