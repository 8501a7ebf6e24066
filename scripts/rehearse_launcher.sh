#!/bin/bash
# round 3, second GPU visit: GPU suite with the new entry points (eval_poses, ea_comm_*), bench line, 2-rank rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03b_tests.txt 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03b_tests.txt
tail -5 gpurun_out/r03b_tests.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03b_bench_c2_steps20.json 2> gpurun_out/r03b_bench_c2_steps20.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03b_bench_c2_steps20.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.5f serial-dependent %.4g single_eval %.4f eval_poses_call %.4f" % (d["value"], d["ms_per_step"], d["value_serial_dependent_steps"], d["single_eval_call_ms"], d["eval_poses_call_ms"]))
print("kernel_ms %.5f frac %.3f floor %.5f ceiling %.3f" % (r["kernel_ms"], r["frac"], r["launch_floor_ms"], r["frac_ceiling_at_floor"]))
print("lm it/s", d.get("lm_iters_per_s_at_1e5_pts"), "gather ms", d.get("pose_gather_ms"))
print("sharded", d.get("lm_point_sharded_device_1e5_pts"))
print("c4", {k: d["c4_batch_32_pairs_per_gpu"].get(k) for k in ("solve_ms", "pose_gather_ms", "pose_gather", "lm_iters_per_s")})
print("comm_error", d.get("comm_error"))
PY
timeout -k 10 600 python bench.py --gpus 2 --dist-backend gloo --force-device 0 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03b_bench_2ranks_gloo.json 2> gpurun_out/r03b_bench_2ranks_gloo.err
echo "2-rank rehearsal rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03b_bench_2ranks_gloo.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "incomplete:", d.get("extras_incomplete"))
print("sharded", d.get("lm_point_sharded_device_1e5_pts"))
PY
timeout -k 10 600 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03b_bench_force_dist_rccl.json 2> gpurun_out/r03b_bench_force_dist_rccl.err
echo "one-rank RCCL group rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03b_bench_force_dist_rccl.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "incomplete:", d.get("extras_incomplete"), "comm_error", d.get("comm_error"))
print("sharded", d.get("lm_point_sharded_device_1e5_pts"))
print("c4", {k: d["c4_batch_32_pairs_per_gpu"].get(k) for k in ("solve_ms", "pose_gather_ms", "pose_gather", "lm_iters_per_s")})
PY
