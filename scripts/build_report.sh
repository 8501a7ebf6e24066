#!/bin/bash
# rebuild libea_hip.so and print VGPR / scratch / occupancy per kernel
cd "$(dirname "$0")/../edge_alignment_amd" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage -o lib/libea_hip.so csrc/ea_kernels.hip csrc/ea_preprocess.hip csrc/ea_capi.hip 2>&1 | grep -E "error|warning:|Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: //' | paste - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | sed 's/_ZN2ea//; s/EEvPKNS.*iiii//; s/EPKNS.*Pi//; s/EPKNS.*OutE//' | cut -c1-170
