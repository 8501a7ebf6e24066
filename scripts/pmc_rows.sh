#!/bin/bash
# Materialised mode: rocprofv3 kernel stats and HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of
# ea_eval_rows_kernel on C5 (fp32) and the 32 x C2 batch (fp32).  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (guide, HBM section).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_rows; mkdir -p $O
for w in c5 batch32f32; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o prof -- python3 $R/scripts/prof_rows.py $w > $O/stats_$w.log 2>&1); echo "stats $w rc=$?"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $ctr -d $O/${w}_$ctr -o p --output-format csv -- python3 $R/scripts/prof_rows.py $w > $O/${w}_$ctr.log 2>&1)
    rc=$?; echo "$w $ctr rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
done
python3 - "$O" <<'PY'
import csv, glob, json, statistics, sys
out = sys.argv[1]
res = {}
alg = {"c5": 10 * 4 * 1000000 + 2048 * 1536 * 4, "batch32f32": 32 * (10 * 4 * 50000 + 640 * 480 * 4)}
for w in ("c5", "batch32f32"):
    med = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        xs = []
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, w, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr and "ea_eval_rows_kernel" in r["Kernel_Name"]:
                    xs.append(float(r["Counter_Value"]))
        if xs:
            med[ctr] = statistics.median(xs)
            print(w, ctr, "dispatches", len(xs), "median_KB", med[ctr], "min", min(xs), "max", max(xs))
    if len(med) == 2:
        hb = (2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024
        res["rows_" + w] = {"hbm_bytes_per_launch": hb, "FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"],
                            "algorithmic_bytes_per_launch": alg[w], "ratio": hb / alg[w], "round": 2,
                            "note": "scripts/pmc_rows.sh: ea_eval_rows_kernel (materialised mode, J row-major through LDS), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, median over its launches"}
    for f in glob.glob("%s/stats_%s/**/*kernel_stats.csv" % (out, w), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ea_eval_rows_kernel" in r["Name"]:
                print(w, "kernel stats: calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
json.dump(res, open(out + "/traffic_rows.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
