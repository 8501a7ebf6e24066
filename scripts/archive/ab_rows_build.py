"""Builds of the library with the fp64 occupancy target of the rows kernel set to 4 / 5 / 6 wavefronts per SIMD
(lib/alt/libea_hip_rw<N>.so), for scripts/ab_rows.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import build
alt = os.path.join(os.path.dirname(build.LIB), "alt")
os.makedirs(alt, exist_ok=True)
for w in (4, 5, 6):
    print(build.build_library(force=True, out=os.path.join(alt, "libea_hip_rw%d.so" % w), defines=("EA_ROWS_WAVES_F64=%d" % w,)))
