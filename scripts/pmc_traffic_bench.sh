#!/bin/bash
# HBM-side traffic of the bench command's own dominant kernel (the evaluation with the previous step's fold riding in it):
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --workload W --steps 200`, counters only.
# bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request; guide, HBM section).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bench; mkdir -p $O
for w in c2 c5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $ctr -d $O/${w}_$ctr -o p --output-format csv -- python3 $R/bench.py --workload $w --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $O/${w}_$ctr.log 2>&1)
    rc=$?; echo "$w $ctr rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
done
python3 - "$O" "$R" <<'PY'
import csv, glob, json, statistics, sys
out, root = sys.argv[1], sys.argv[2]
res = {}
for w in ("c2", "c5"):
    med = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        by = {}
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, w, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    by.setdefault(r["Kernel_Name"].split("(")[0][-60:], []).append(float(r["Counter_Value"]))
        for k, v in by.items():
            print(w, ctr, k, "dispatches", len(v), "median_KB", statistics.median(v), "min", min(v), "max", max(v))
            if "ea_eval_fold_kernel" in k:
                med[ctr] = statistics.median(v)
    if len(med) == 2:
        res[w + "_riding_fold"] = {"hbm_bytes_per_launch": (2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024,
                                   "FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"], "round": 2,
                                   "note": "scripts/pmc_traffic_bench.sh: the bench command's own kernel (ea_eval_fold_kernel: evaluation k + the fold of step k-1), "
                                           "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, median over its launches; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"}
json.dump(res, open(out + "/traffic_riding.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
