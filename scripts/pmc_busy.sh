#!/bin/bash
# A duration for the bench command's dominant kernel that is neither bench.py's hipEvent clock nor the tracer's bracket
# (VERDICT r02 item 5): GPU-busy cycles per launch from the counters, over the shader clock.
#   pass 1: SQ_BUSY_CYCLES SQ_WAVES   (SQ_BUSY_CYCLES is summed over the 32 shader engines' sequencers: / 32)
#   pass 2: GRBM_GUI_ACTIVE           (summed over the 8 XCDs: / 8; contains the command processor's share of a dispatch)
# counters only (no trace domains), separate passes, over `bench.py --workload W --steps 200 --no-extras --no-cpu-baseline`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_busy; mkdir -p $O
for w in ${WORKLOADS:-c2 c5}; do
  for pass in "SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rm -rf $O/${w}_$tag
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $pass -d $O/${w}_$tag -o p --output-format csv -- python3 $R/bench.py --workload $w --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $O/${w}_$tag.log 2>&1)
    rc=$?; echo "$w $tag rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
done
python3 - "$O" <<'PY'
import csv, glob, json, statistics, sys
out = sys.argv[1]
CLOCK = 2.4e9   # shader clock the guide quotes; the chip holds it on these short, mostly waiting launches
res = {}
for w in ("c2", "c5"):
    med = {}
    for tag, ctrs in (("SQ_BUSY_CYCLES", ("SQ_BUSY_CYCLES", "SQ_WAVES")), ("GRBM_GUI_ACTIVE", ("GRBM_GUI_ACTIVE",))):
        by = {}
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, w, tag), recursive=True):
            for r in csv.DictReader(open(f)):
                # the one-pose launch of the evaluation kernel (an LM iteration's launch; the smallest grid of that kernel in the run)
                if r["Counter_Name"] in ctrs and "ea_eval_fused_kernel" in r["Kernel_Name"]:
                    by.setdefault((r["Counter_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
        small = min((g for (_, g) in by), default=None)
        by = {c: v for (c, g), v in by.items() if g == small}
        for k, v in by.items():
            med[k] = statistics.median(v)
            print(w, k, "dispatches", len(v), "median", med[k], "min", min(v), "max", max(v))
    if "SQ_BUSY_CYCLES" in med:
        res[w] = {"kernel": "ea_eval_fused_kernel, one pose per launch", "sq_busy_cycles_per_launch": med["SQ_BUSY_CYCLES"], "sq_waves_per_launch": med.get("SQ_WAVES"),
                  "kernel_ms_from_counters": med["SQ_BUSY_CYCLES"] / 32.0 / CLOCK * 1e3,
                  "grbm_gui_active_per_launch": med.get("GRBM_GUI_ACTIVE"),
                  "kernel_ms_from_grbm_gui_active": (med["GRBM_GUI_ACTIVE"] / 8.0 / CLOCK * 1e3) if "GRBM_GUI_ACTIVE" in med else None,
                  "round": 3,
                  "note": "scripts/pmc_busy.sh: rocprofv3 --pmc over bench.py --workload %s --steps 200; SQ_BUSY_CYCLES / 32 sequencers / 2.4 GHz = time some wave of the "
                          "launch was resident; GRBM_GUI_ACTIVE / 8 XCDs / 2.4 GHz also holds the command processor's share of the dispatch" % w}
json.dump(res, open(out + "/pmc_busy.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
