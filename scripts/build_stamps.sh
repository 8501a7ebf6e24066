#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (never shipped, never timed): lib/libea_hip_stamps.so
cd "$(dirname "$0")/../edge_alignment_amd" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=16 -DEA_STAMPS -Wall -Wno-unused-function -o lib/libea_hip_stamps.so csrc/ea_kernels.hip csrc/ea_preprocess.hip csrc/ea_capi.hip
