#!/bin/bash
# PMC passes for the fused evaluation kernel, a few counters per pass (a wide TA/TCP set makes rocprofiler abort with
# "Request exceeds the capabilities of the hardware to collect": round 1's "hung" pass).  Counters only, no tracing.
# usage: scripts/pmc_eval.sh <workload of scripts/prof_run.py> <tag> [tuning k=v ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-batch32f32}
TAG=${2:-$W}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
while read -r set; do
  [ -z "$set" ] && continue
  [ -n "$PASSES" ] && [ $i -ge $PASSES ] && break
  i=$((i+1))
  (cd /tmp && timeout -k 10 150 rocprofv3 --pmc $set -d $OUT/p$i -o p$i --output-format csv -- python3 $R/scripts/prof_run.py $W "${@:3}" > $OUT/p$i.log 2>&1)
  rc=$?
  echo "pass $i ($set) rc=$rc" >> $OUT/passes.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out: stopping" >> $OUT/passes.txt; break; fi
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD
GRBM_GUI_ACTIVE GRBM_TA_BUSY
TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_READ_sum
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'ea_eval' not in r.get('Kernel_Name', ''): continue
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    fh.write(open(out + '/passes.txt').read())
    for k, v in agg.items():
        fh.write('%-36s launches %4d  mean per launch %.6g\n' % (k, len(v), sum(v) / len(v)))
print(open(out + '/summary.txt').read())
PY
