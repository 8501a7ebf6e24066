#!/bin/bash
# Regenerates profiles/pmc_valu.json (what bench.py's roofline.secondary divides by the live kernel time) from two SQ counter
# passes per workload (scripts/pmc_eval.sh with PASSES=2).  cheap_class_fraction_static is kept (scripts/isa_breakdown.py).
R=$GRAFT_REPO_ROOT
for w in c2 c5 batch32f32 batch32f64; do
  PASSES=2 bash $R/scripts/pmc_eval.sh $w valu_$w > /dev/null 2>&1; echo "$w done"
done
python3 - <<'PY'
import json, os, re
R = os.environ["GRAFT_REPO_ROOT"]
p = os.path.join(R, "profiles", "pmc_valu.json")
d = json.load(open(p))
names = {"c2": "c2", "c5": "c5", "batch32f32": "batch32_c2_fp32", "batch32f64": "batch32_c2_fp64"}
keys = {"SQ_INSTS_VALU": "sq_insts_valu_per_launch", "SQ_WAVES": "sq_waves_per_launch", "SQ_INSTS_SALU": "sq_insts_salu_per_launch",
        "SQ_INSTS_VMEM_RD": "sq_insts_vmem_rd_per_launch", "SQ_WAIT_ANY": "sq_wait_any", "SQ_WAIT_INST_ANY": "sq_wait_inst_any", "SQ_WAVE_CYCLES": "sq_wave_cycles"}
for w, k in names.items():
    txt = open(os.path.join(R, "gpurun_out", "pmc_valu_%s" % w, "summary.txt")).read()
    for c, f in keys.items():
        m = re.search(r"^%s\s+launches\s+\d+\s+mean per launch (\S+)" % c, txt, re.M)
        if m:
            d.setdefault(k, {})[f] = float(m.group(1))
    print(k, d[k])
json.dump(d, open(os.path.join(R, "gpurun_out", "pmc_valu.json"), "w"), indent=1)
PY
