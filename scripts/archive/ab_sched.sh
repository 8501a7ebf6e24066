#!/bin/bash
# Same-box A/B of a machine-scheduling strategy for the device compile.  Build the alternative HERE first (no GPU needed):
#   scripts/ab_sched.sh build max-ilp        -> scripts/_ab/max-ilp.so  (every source compiled with the strategy)
# then on the GPU box:  scripts/ab_sched.sh run max-ilp   (shipped library and alternative, interleaved twice)
cd "$(dirname "$0")/.."
S=${2:-max-ilp}
if [ "$1" = build ]; then
  mkdir -p scripts/_ab
  (cd edge_alignment_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=16 \
     -mllvm -amdgpu-sched-strategy=$S -Wall -Wno-unused-function -o ../scripts/_ab/$S.so csrc/ea_kernels.hip csrc/ea_kernels_var.hip csrc/ea_preprocess.hip csrc/ea_capi.hip)
else
  for rep in 1 2; do
    for lib in "" scripts/_ab/$S.so; do
      EA_HIP_LIB=$lib timeout -k 10 300 python scripts/ab_build.py 2>&1 | grep -v amdgpu.ids
      EA_HIP_LIB=$lib timeout -k 10 200 python scripts/ab_variant.py 2>&1 | grep -v amdgpu.ids
    done
  done
fi
