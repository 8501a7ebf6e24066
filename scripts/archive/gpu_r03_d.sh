#!/bin/bash
# round 3, fourth GPU visit: the whole suite, one soak run of each kind after the criteria change, the driver-shaped line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03d_tests.txt 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03d_tests.txt
tail -5 gpurun_out/r03d_tests.txt
(timeout -k 10 260 python scripts/soak_variants.py 150 4321 2>&1 | tail -40) > gpurun_out/r03d_soak_variants.txt
echo "soak variants rc=$?"; tail -3 gpurun_out/r03d_soak_variants.txt
(timeout -k 10 260 python scripts/soak.py 150 777 2>&1 | tail -20) > gpurun_out/r03d_soak.txt
echo "soak rc=$?"; tail -2 gpurun_out/r03d_soak.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03d_bench_c2_steps20.json 2> gpurun_out/r03d_bench_c2_steps20.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03d_bench_c2_steps20.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.5f serial-dependent %.4g single_eval %.4f" % (d["value"], d["ms_per_step"], d["value_serial_dependent_steps"], d["single_eval_call_ms"]))
print("kernel_ms %.5f frac %.3f floor %.5f ceiling %.3f counter", r["kernel_ms"], r["frac"], r["launch_floor_ms"], r["frac_ceiling_at_floor"], r["counter_busy"])
print("lm it/s", d.get("lm_iters_per_s_at_1e5_pts"), "sharded", d.get("lm_point_sharded_device_1e5_pts", {}).get("iters_per_s"))
for k, v in d["other_workloads"].items():
    print(k, "kernel_us %.2f frac %.3f tex %s" % (v.get("kernel_us", 0), v.get("roofline_frac", 0), v.get("image_texel_bytes_read")), "mat", v.get("materialised_mode", {}).get("frac"), v.get("materialised_mode", {}).get("frac_bytes_moved"))
PY
