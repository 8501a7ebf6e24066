#!/bin/bash
# The round's committed evidence in one GPU visit: the GPU suite, bench lines, rocprofv3 kernel-trace stats of the same commands,
# PMC passes, the same-box A/Bs of the LM loop's forms, one run of each soak.  Summaries land in gpurun_out/r03e/ and are
# copied into profiles/ by hand (profiles/README.md says which).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
timeout -k 10 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2_steps20.json 2> $O/bench_c2_steps20.err; echo "bench c2 steps20 rc=$?"
timeout -k 10 300 python bench.py --serial-steps --no-extras --no-cpu-baseline > $O/bench_c2_serial_steps.json 2> /dev/null; echo "bench serial rc=$?"
timeout -k 10 300 python bench.py --workload c5 --no-extras --no-cpu-baseline > $O/bench_c5.json 2> /dev/null; echo "bench c5 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2_steps20 -o k -- python3 $R/bench.py --steps 20 --warmup 5 > $O/kt_c2_steps20.log 2>&1; echo "kernel-trace steps20 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2 -o k -- python3 $R/bench.py --no-extras --no-cpu-baseline > $O/kt_c2.log 2>&1; echo "kernel-trace c2 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -o k -- python3 $R/bench.py --workload c5 --no-extras --no-cpu-baseline > $O/kt_c5.log 2>&1; echo "kernel-trace c5 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sharded -o k -- python3 $R/scripts/prof_sharded.py > $O/kt_sharded.log 2>&1; echo "kernel-trace sharded rc=$?"
tail -3 $O/kt_sharded.log
cd $R
timeout -k 10 300 python scripts/ab_fused_iterations.py 5 > $O/ab_fused_iterations.txt 2>&1; echo "ab fused iterations rc=$?"; grep -v "^{" $O/ab_fused_iterations.txt | cut -c1-260
timeout -k 10 300 python scripts/ab_sharded_rows.py > $O/ab_sharded_rows.txt 2>&1; echo "ab sharded rows rc=$?"; grep "it/s" $O/ab_sharded_rows.txt
timeout -k 10 200 python scripts/stamps.py > $O/lm_stamps.txt 2>&1; echo "stamps rc=$?"; tail -16 $O/lm_stamps.txt
timeout -k 10 400 python scripts/soak.py 120 777 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -2 $O/soak.txt | cut -c1-400
timeout -k 10 400 python scripts/soak_variants.py 100 4321 > $O/soak_variants.txt 2>&1; echo "soak variants rc=$?"; tail -2 $O/soak_variants.txt | cut -c1-400
timeout -k 10 400 python scripts/soak_fused.py 100 2025 > $O/soak_fused.txt 2>&1; echo "soak fused rc=$?"; tail -1 $O/soak_fused.txt | cut -c1-400
STEPS=20 bash scripts/pmc_traffic_bench.sh > $O/pmc_traffic_bench.txt 2>&1; echo "pmc traffic (20 poses) rc=$?"
STEPS=2000 bash scripts/pmc_traffic_bench.sh >> $O/pmc_traffic_bench.txt 2>&1; echo "pmc traffic (2000 poses) rc=$?"; grep -n "hbm_bytes_per_launch\|_poses_" $O/pmc_traffic_bench.txt | tail -12
STEPS=20 bash scripts/pmc_busy.sh > $O/pmc_busy.txt 2>&1; echo "pmc busy (20) rc=$?"
STEPS=2000 bash scripts/pmc_busy.sh >> $O/pmc_busy.txt 2>&1; echo "pmc busy (2000) rc=$?"; grep -n "kernel_ms_from_counters\|\"c[25]" $O/pmc_busy.txt | tail -16
bash scripts/pmc_valu_refresh.sh > $O/pmc_valu_refresh.txt 2>&1; echo "pmc valu rc=$?"; tail -5 $O/pmc_valu_refresh.txt
find $O -name "*kernel_stats.csv" | head; 
for d in kt_c2_steps20 kt_c2 kt_c5 kt_sharded; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -8 $f | cut -c1-200; done
# the tracer's durations per (kernel, grid): --stats averages every shape the kernel name was launched in
for d in kt_c2_steps20 kt_c2 kt_c5; do echo "== $d"; python3 scripts/kt_by_grid.py $O/$d ea_eval_poses; done > $O/kt_by_grid.txt 2>&1; cat $O/kt_by_grid.txt | cut -c1-140
# the raw per-dispatch tables are large (gpurun_out/ travels back only below 64 MiB): keep the summaries
find $R/gpurun_out -name "*kernel_trace.csv" -delete; find $R/gpurun_out -name "*counter_collection.csv" -delete; find $R/gpurun_out -name "*.log" -size +200k -delete
du -sh $R/gpurun_out
