#!/bin/bash
# bench.py control-flow checks: watchdog, one-rank RCCL group through every N>1 branch, two gloo ranks, plain run
cd /tmp && export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --extras-timeout 1.5 > $O/b6_watchdog.json 2> $O/b6.err; echo "watchdog rc=$?"
timeout -k 10 300 python bench.py --force-dist --steps 200 --warmup 20 --no-cpu-baseline > $O/b6_forcedist.json 2>> $O/b6.err; echo "forcedist rc=$?" &&
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 200 --warmup 20 --dist-backend gloo --force-device 0 > $O/b6_gloo2.json 2>> $O/b6.err; echo "gloo2 rc=$?" &&
timeout -k 10 400 python bench.py > $O/b6_plain.json 2>> $O/b6.err; echo "plain rc=$?"
python - <<'PY'
import json
for f in ("b6_watchdog", "b6_forcedist", "b6_gloo2", "b6_plain"):
    try:
        lines = [l for l in open("gpurun_out/%s.json" % f).read().strip().splitlines() if l.startswith("{")]
        d = json.loads(lines[-1])
        print(f, "lines", len(lines), "value %.3e n_gpus %d" % (d["value"], d["n_gpus"]), "incomplete:", d.get("extras_incomplete"),
              "| sharded host", (d.get("lm_point_sharded_1e5_pts") or {}).get("iters_per_s"), "device", (d.get("lm_point_sharded_device_1e5_pts") or {}),
              "| c4", {k: (d.get("c4_batch_32_pairs_per_gpu") or {}).get(k) for k in ("solve_ms", "pose_gather_ms", "converged", "error")},
              "| cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", repr(e))
PY
tail -5 $O/b6.err
