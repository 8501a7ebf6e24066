#!/bin/bash
# round 2, visit 2: parity tests, same-box A/B of the r01 build vs the current one, PMC passes
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest_exit=$?"; tail -3 $O/pytest_gpu.log
EA_HIP_LIB=$PWD/edge_alignment_amd/lib/libea_hip_r01.so timeout -k 10 300 python scripts/ab_build.py > $O/ab_r01.txt 2>&1; echo "ab_r01=$?"
timeout -k 10 300 python scripts/ab_build.py > $O/ab_cur.txt 2>&1; echo "ab_cur=$?"
timeout -k 10 300 python scripts/ab_build.py buffer_loads=1 > $O/ab_cur_buf.txt 2>&1; echo "ab_buf=$?"
cat $O/ab_r01.txt $O/ab_cur.txt $O/ab_cur_buf.txt
