#!/bin/bash
# Is the LM-step kernel's first-pass penalty instruction fetch?  SQC instruction- and scalar-data-cache requests / misses of
# the step kernel over 40 solves of the 1e5-point problem, one small counter set per pass, counters only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lm_icache; mkdir -p $O
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  (cd /tmp && timeout -k 10 150 rocprofv3 --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/scripts/prof_lm_solves.py > $O/p$i.log 2>&1)
  rc=$?; echo "pass $i ($set) rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; break; fi
done <<'SETS'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM
SETS
python3 - "$O" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = 'step' if 'ea_lm_step' in r['Kernel_Name'] else ('eval' if 'ea_eval_fused' in r['Kernel_Name'] else None)
        if k:
            agg.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, c), v in agg.items():
    print('%-5s %-22s launches %5d  mean per launch %.6g  min %.6g max %.6g' % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
PY
