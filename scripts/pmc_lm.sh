#!/bin/bash
# PMC passes (counters only, SQ block) for the kernels of the 1e5-point solve -- ea_lm_iter_kernel (one launch per iteration) and,
# for problems that do not qualify, ea_lm_step_kernel + the evaluation; summary under gpurun_out/pmc_lm
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lm
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_LDS"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $R/scripts/prof_run.py lm > $OUT/p$i.log 2>&1) || echo "pass $i failed" >> $OUT/fail.txt
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_lm'
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        kn = r.get('Kernel_Name', '')
        tag = 'lm_iter' if 'ea_lm_iter' in kn else 'lm_step' if 'ea_lm_step' in kn else ('reduce' if 'ea_reduce' in kn else ('eval' if 'ea_eval_fused' in kn else None))
        if tag is None: continue
        agg.setdefault((tag, r['Counter_Name']), []).append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for (tag, k), v in agg.items():
        fh.write('%-8s %-28s launches %4d  mean per launch %.6g\n' % (tag, k, len(v), sum(v) / len(v)))
print(open(out + '/summary.txt').read())
PY
