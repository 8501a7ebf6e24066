#!/bin/bash
# HBM-side traffic of the bench command's own dominant kernel -- the evaluation launch of G poses (ea_eval_poses_kernel: the
# evaluation kernel under the name its pose-batched launches carry; full launches = the largest grid of the run) --:
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --workload W --steps S`, counters only.
# bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request; guide, HBM section).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bench; mkdir -p $O
STEPS=${STEPS:-20}
for w in c2 c5; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${w}_$ctr
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $ctr -d $O/${w}_$ctr -o p --output-format csv -- python3 $R/bench.py --workload $w --steps $STEPS --warmup 5 --no-extras --no-cpu-baseline > $O/${w}_$ctr.log 2>&1)
    rc=$?; echo "$w $ctr rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
done
python3 - "$O" "$R" "$STEPS" <<'PY'
import csv, glob, json, statistics, sys
out, root, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
res = {}
for w in ("c2", "c5"):
    med = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        by = {}
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, w, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    by.setdefault((r["Kernel_Name"].split("(")[0][-60:], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
        for k, v in sorted(by.items()):
            print(w, ctr, k, "dispatches", len(v), "median_KB", statistics.median(v), "min", min(v), "max", max(v))
        fused = [k for k in by if "ea_eval_poses_kernel" in k[0]]
        if fused:
            big = max(fused, key=lambda k: k[1])   # the launch of G poses: the largest grid of that kernel in the run
            med[ctr] = statistics.median(by[big])
            med["grid"] = big[1]
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        res["%s_poses_%d" % (w, steps)] = {"hbm_bytes_per_launch": (2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024,
                                   "FETCH_SIZE_KB": med["FETCH_SIZE"], "WRITE_SIZE_KB": med["WRITE_SIZE"], "grid_size": med["grid"], "round": 3,
                                   "note": "scripts/pmc_traffic_bench.sh: the bench command's own dominant kernel (ea_eval_poses_kernel: the launch of G poses, of %d in the timed region), "
                                           "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps %d, median over its launches; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. "
                                           "The poses of one launch share the points and the image: the later ones find them in L2 / Infinity Cache, so the HBM-side bytes are far BELOW the algorithmic bytes (one pass per pose)" % (steps, steps)}
json.dump(res, open(out + "/traffic_poses_%d.json" % steps, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
