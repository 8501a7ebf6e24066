#!/bin/bash
# round 3, ninth GPU visit: one launch per LM iteration (ea_lm_iter_kernel) -- parity with the pair form, then the A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_fused_iterations.py -x -q > $O/tests_fused.txt 2>&1; rc=$?; echo "fused tests rc=$rc"; tail -15 $O/tests_fused.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/ab_fused_iterations.py 5 > $O/ab_fused_iterations.txt 2>&1; echo "ab rc=$?"; cat $O/ab_fused_iterations.txt | cut -c1-400
