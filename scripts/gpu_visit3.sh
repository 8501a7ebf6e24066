#!/bin/bash
# round 2, visit 3: GPU tests, driver-shaped and default bench, 2-rank gloo rehearsal of the N>1 control flow, rocprof
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest_exit=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; echo "bench_s20=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_s20b.json 2>> $O/bench_s20.err; echo "bench_s20b=$?"
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_s2000.json 2> $O/bench_s2000.err; echo "bench_s2000=$?"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --force-device 0 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "bench_2rank=$?"
python - <<'PY'
import json
for f in ("bench_s20", "bench_s20b", "bench_s2000", "bench_2rank_gloo"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.3e ms/step %.5f kernel_ms %.5f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]),
              {k: d[k] for k in d if k.startswith("c4") or k.startswith("lm_iters")})
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 $O/bench_2rank_gloo.err
