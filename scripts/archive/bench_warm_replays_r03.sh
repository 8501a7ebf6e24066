#!/bin/bash
# the driver's 20-step line against the number of untimed runs of the timed call before the clock starts (EA_BENCH_WARM_REPLAYS)
cd $GRAFT_REPO_ROOT
for w in 0 2 5 10 20 50; do
  for rep in 1 2 3; do
    EA_BENCH_WARM_REPLAYS=$w timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('warm runs $w: region %.1f us, %.3g evals/s' % (d['ms_per_step'] * 20 * 1e3, d['value']))"
  done
done
