#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (never shipped, never timed): lib/libea_hip_stamps.so
cd "$(dirname "$0")/.." && python -c "
from edge_alignment_amd import build
import os
print(build.build_library(force=True, out=os.path.join(os.path.dirname(build.LIB), 'libea_hip_stamps.so'), defines=('EA_STAMPS',)))"
