#!/bin/bash
# bench.py's 20-step line with 0, 1, 2, 4 and 8 extra untimed replays of the region's graph before the clock starts (EA_BENCH_WARM_REPLAYS),
# three runs each, interleaved.  Output -> profiles/r02_bench_warm_replays.txt
set -o pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/bench_warm_replays.txt
: > $out
for rep in 1 2 3; do
  for w in 0 1 2 4 8; do
    EA_BENCH_WARM_REPLAYS=$w timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/_bw.json 2> gpurun_out/_bw.err || { echo "bench failed (warm $w)"; tail -5 gpurun_out/_bw.err; exit 1; }
    python - $w $rep >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_bw.json").read().strip().splitlines()[-1])
print("warm replays %s run %s: value %.4g evals/s  %.3f us/step (region %.1f us)" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"] * 1e3, d["ms_per_step"] * 1e3 * d["steps"]))
PY
  done
done
cat $out
