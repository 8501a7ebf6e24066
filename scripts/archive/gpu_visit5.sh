#!/bin/bash
# round 2, visit 5: rocprofv3 kernel stats + HBM traffic + SQ instruction counters for the bench workloads, outlier probe
cd /tmp && export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out; mkdir -p $O
for w in c2 c5; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_prof_$w -o prof -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-extras --steps 2000 --warmup 200 > $O/r02_prof_$w.log 2>&1); echo "prof_$w=$?"
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r02_pmc_fetch_$w -o p -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/r02_pmc_fetch_$w.log 2>&1); echo "fetch_$w=$?"
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r02_pmc_write_$w -o p -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/r02_pmc_write_$w.log 2>&1); echo "write_$w=$?"
done
export PASSES=2
for w in c2 c5 batch32f32 batch32f64; do bash scripts/pmc_eval.sh $w r02_$w > $O/pmc_r02_$w.log 2>&1; echo "pmc_$w=$?"; done
timeout -k 10 200 python scripts/outlier_probe.py > $O/outlier_probe.txt 2>&1; echo "outlier=$?"; cat $O/outlier_probe.txt
timeout -k 10 200 python scripts/bench_bracket.py > $O/bench_bracket.txt 2>&1; echo "bracket=$?"
find $O/r02_prof_c2 $O/r02_prof_c5 -name "*kernel_stats.csv" | head
