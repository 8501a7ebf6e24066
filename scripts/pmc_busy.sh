#!/bin/bash
# A duration for the bench command's dominant kernel that is neither bench.py's hipEvent clock nor the tracer's bracket
# (VERDICT r02 item 5): GPU-busy cycles per launch from the counters, over the shader clock.
#   pass 1: SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU   (SQ_BUSY_CYCLES is summed over the 32 shader engines' sequencers: / 32;
#           SQ_INSTS_VALU of the pose-batched launch feeds the bench line's vector-issue bound)
#   pass 2: GRBM_GUI_ACTIVE           (summed over the 8 XCDs: / 8; contains the command processor's share of a dispatch)
# counters only (no trace domains), separate passes, over `bench.py --workload W --steps 200 --no-extras --no-cpu-baseline`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_busy; mkdir -p $O
STEPS=${STEPS:-20}
for w in ${WORKLOADS:-c2 c5}; do
  for pass in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rm -rf $O/${w}_$tag
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $pass -d $O/${w}_$tag -o p --output-format csv -- python3 $R/bench.py --workload $w --steps $STEPS --warmup 5 --no-extras --no-cpu-baseline > $O/${w}_$tag.log 2>&1)
    rc=$?; echo "$w $tag rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
done
python3 - "$O" "$STEPS" <<'PY'
import csv, glob, json, statistics, sys
out, steps = sys.argv[1], int(sys.argv[2])
CLOCK = 2.4e9   # shader clock the guide quotes
res = {}
for w in ("c2", "c5"):
    # "ea_eval_poses_kernel": the timed region's launch of G poses (the dominant kernel); "ea_eval_fused_kernel": the same
    # kernel at one pose per launch (an LM iteration's launch, the secondary measurements of the same command)
    for kname, key in (("ea_eval_poses_kernel", "%s_poses_%d" % (w, steps)), ("ea_eval_fused_kernel", w)):
        med = {}
        for tag, ctrs in (("SQ_BUSY_CYCLES", ("SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU")), ("GRBM_GUI_ACTIVE", ("GRBM_GUI_ACTIVE",))):
            by = {}
            for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, w, tag), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] in ctrs and kname in r["Kernel_Name"]:
                        by.setdefault((r["Counter_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
            grid = max((g for (_, g) in by), default=None)   # (poses kernel: the full launches of G poses)
            for (c, g), v in by.items():
                if g == grid:
                    med[c] = statistics.median(v)
                    print(w, kname, c, "grid", g, "dispatches", len(v), "median", med[c], "min", min(v), "max", max(v))
            if grid:
                med["grid"] = grid
        if "SQ_BUSY_CYCLES" in med:
            res[key] = {"kernel": kname, "grid_size": med.get("grid"), "sq_busy_cycles_per_launch": med["SQ_BUSY_CYCLES"], "sq_waves_per_launch": med.get("SQ_WAVES"), "sq_insts_valu_per_launch": med.get("SQ_INSTS_VALU"),
                        "kernel_ms_from_counters": med["SQ_BUSY_CYCLES"] / 32.0 / CLOCK * 1e3,
                        "grbm_gui_active_per_launch": med.get("GRBM_GUI_ACTIVE"),
                        "kernel_ms_from_grbm_gui_active": (med["GRBM_GUI_ACTIVE"] / 8.0 / CLOCK * 1e3) if "GRBM_GUI_ACTIVE" in med else None,
                        "round": 3,
                        "note": "scripts/pmc_busy.sh: rocprofv3 --pmc over bench.py --workload %s --steps %d; SQ_BUSY_CYCLES / 32 sequencers / 2.4 GHz = time some wave of the "
                                "launch was resident; GRBM_GUI_ACTIVE / 8 XCDs / 2.4 GHz also holds the command processor's share of the dispatch" % (w, steps)}
json.dump(res, open(out + "/pmc_busy_%d.json" % steps, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
