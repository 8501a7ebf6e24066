#!/bin/bash
# same-box A/B of the fp64 occupancy target of the materialised-mode kernel: builds made in the build container
# (python scripts/ab_rows_build.py), run here one after the other, twice
cd /tmp && export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd $R
for rep in 1 2; do
  for w in 4 5 6; do
    lib=$R/edge_alignment_amd/lib/alt/libea_hip_rw$w.so
    [ -f $lib ] || continue
    echo "== EA_ROWS_WAVES_F64=$w (pass $rep)"
    EA_HIP_LIB=$lib timeout -k 10 200 python scripts/rows_sweep.py f64only 2>&1 | grep -v amdgpu.ids | grep "layout 0 staged  \|soa  "
  done
done
