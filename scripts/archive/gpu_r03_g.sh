#!/bin/bash
# round 3: smoke(), the plain-command 2-rank rehearsal (gloo, both ranks on device 0), the one-rank RCCL group
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
echo "smoke rc=$?"
timeout -k 10 600 python bench.py --gpus 2 --dist-backend gloo --force-device 0 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03g_bench_2ranks_gloo.json 2> gpurun_out/r03g_bench_2ranks_gloo.err
echo "2-rank rehearsal rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03g_bench_2ranks_gloo.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "incomplete:", d.get("extras_incomplete"), "comm_error", d.get("comm_error"))
print("sharded", d.get("lm_point_sharded_device_1e5_pts"))
print("c4", {k: d["c4_batch_32_pairs_per_gpu"].get(k) for k in ("solve_ms", "pose_gather_ms", "pose_gather", "lm_iters_per_s")})
PY
timeout -k 10 600 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03g_bench_force_dist_rccl.json 2> gpurun_out/r03g_bench_force_dist_rccl.err
echo "one-rank RCCL group rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03g_bench_force_dist_rccl.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "incomplete:", d.get("extras_incomplete"), "comm_error", d.get("comm_error"))
print("sharded", d.get("lm_point_sharded_device_1e5_pts"))
print("c4", {k: d["c4_batch_32_pairs_per_gpu"].get(k) for k in ("solve_ms", "pose_gather_ms", "pose_gather", "lm_iters_per_s")})
PY
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03g_bench_torchrun1.json 2> gpurun_out/r03g_bench_torchrun1.err
echo "torchrun N=1 rc=$?"; head -c 300 gpurun_out/r03g_bench_torchrun1.json
