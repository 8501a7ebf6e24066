#!/bin/bash
# round 3, third GPU visit: suite, dt_f32 A/B, busy counters, kernel-trace stats of the driver-shaped bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_tests.txt 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03c_tests.txt
tail -5 gpurun_out/r03c_tests.txt
timeout -k 10 600 python scripts/ab_dt_f32.py > gpurun_out/r03_ab_dt_f32.txt 2> gpurun_out/r03_ab_dt_f32.err
echo "ab rc=$?"; cat gpurun_out/r03_ab_dt_f32.txt
bash scripts/pmc_busy.sh 2>&1 | tail -30
