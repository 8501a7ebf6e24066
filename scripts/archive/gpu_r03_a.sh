#!/bin/bash
# round 3, first GPU visit: the whole GPU suite, the driver-shaped bench line, the plain-command 2-rank rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_tests.txt 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03a_tests.txt
tail -5 gpurun_out/r03a_tests.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a_bench_c2_steps20.json 2> gpurun_out/r03a_bench_c2_steps20.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03a_bench_c2_steps20.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.5f serial-dependent %.4g single_eval %.4f eval_poses_call %.4f" % (d["value"], d["ms_per_step"], d["value_serial_dependent_steps"], d["single_eval_call_ms"], d["eval_poses_call_ms"]))
print("kernel_ms %.5f frac %.3f floor %.5f ceiling %.3f" % (r["kernel_ms"], r["frac"], r["launch_floor_ms"], r["frac_ceiling_at_floor"]))
print("lm it/s", d.get("lm_iters_per_s_at_1e5_pts"))
PY
timeout -k 10 600 python bench.py --gpus 2 --dist-backend gloo --force-device 0 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03a_bench_2ranks_gloo.json 2> gpurun_out/r03a_bench_2ranks_gloo.err
echo "2-rank rehearsal rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03a_bench_2ranks_gloo.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "incomplete:", d.get("extras_incomplete"))
PY
