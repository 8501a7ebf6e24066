#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest_exit=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 200 python scripts/bench_bracket.py > $O/bench_bracket.txt 2>&1; echo "bracket=$?"; cat $O/bench_bracket.txt
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --force-device 0 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "bench_2rank=$?"
tail -c 1500 $O/bench_2rank_gloo.json; tail -3 $O/bench_2rank_gloo.err
