#!/bin/bash
# HBM-side traffic of the fused evaluation for large batches: FETCH_SIZE and WRITE_SIZE in separate passes (guide: TCC budget)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_traffic_batch; mkdir -p $O
for pairs in 64 256; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $c -d $O/${c}_$pairs -o p --output-format csv -- python3 $R/scripts/prof_run.py batchf32 pairs=$pairs tile=16 > $O/${c}_$pairs.log 2>&1); echo "$c pairs=$pairs rc=$?"
  done
done
python3 - $O <<'PY'
import csv, glob, sys, statistics
o = sys.argv[1]
for pairs in (64, 256):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        xs = []
        for f in glob.glob("%s/%s_%d/**/*counter_collection.csv" % (o, c, pairs), recursive=True):
            for r in csv.DictReader(open(f)):
                if "ea_eval_fused" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    xs.append(float(r["Counter_Value"]))
        v[c] = statistics.median(xs) if xs else float("nan")
    alg = pairs * (3 * 4 * 50000 + 480 * 640 * 4)
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    print("batch of %3d C2 pairs fp32 tile16: FETCH_SIZE %.0f KB WRITE_SIZE %.0f KB -> HBM-side bytes/launch (2*FETCH+WRITE) %.1f MB; algorithmic %.1f MB; ratio %.2f" % (
        pairs, v["FETCH_SIZE"], v["WRITE_SIZE"], hbm / 1e6, alg / 1e6, hbm / alg))
PY
