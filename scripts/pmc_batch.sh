#!/bin/bash
# PMC passes (counters only, no tracing; SQ block only: a TA_* pass hung on this pool) for the fused kernel on the 32 x C2 batch; summaries under gpurun_out/pmc_batch
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_batch
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
cd $R
W=${1:-batch32f32}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" \
           "SQ_THREAD_CYCLES_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $R/scripts/prof_run.py $W "${@:2}" > $OUT/p$i.log 2>&1) || echo "pass $i failed" >> $OUT/fail.txt
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_batch'
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if 'ea_eval_fused' not in r.get('Kernel_Name', ''): continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for k, v in agg.items():
        fh.write('%-36s launches %4d  mean per launch %.6g\n' % (k, len(v), sum(v) / len(v)))
print(open(out + '/summary.txt').read())
PY
