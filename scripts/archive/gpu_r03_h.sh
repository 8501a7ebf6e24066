#!/bin/bash
# quick check: c5 bench + PMC refresh for the pose path's launch shape
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --workload c5 --no-extras --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"; tail -2 $O/bench_c5.err
STEPS=20 bash scripts/pmc_busy.sh > $O/pmc_busy.txt 2>&1; echo "pmc busy (20) rc=$?"
STEPS=2000 bash scripts/pmc_busy.sh >> $O/pmc_busy.txt 2>&1; echo "pmc busy (2000) rc=$?"; grep -n "kernel_ms_from_counters\|sq_insts_valu\|\"c[25]" $O/pmc_busy.txt | tail -24
STEPS=20 bash scripts/pmc_traffic_bench.sh > $O/pmc_traffic_bench.txt 2>&1; echo "pmc traffic (20 poses) rc=$?"
STEPS=2000 bash scripts/pmc_traffic_bench.sh >> $O/pmc_traffic_bench.txt 2>&1; echo "pmc traffic (2000 poses) rc=$?"; grep -n "hbm_bytes_per_launch" $O/pmc_traffic_bench.txt | tail -8
timeout -k 10 300 python scripts/ab_poses_shape.py > $O/ab_poses_shape.txt 2>/dev/null; echo "shape sweep rc=$?"
