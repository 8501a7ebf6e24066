"""rocprofv3 --kernel-trace workload: 20 x set_now_frame on a bundled 640x480 frame (Laplacian flavour)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi
from oracle import preprocess_np as pp
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/rgbd')
rgb3 = pp.load_rgb_as_bgr(os.path.join(G, 'rgb_3.png'))
P = capi.Problem(525.0, 525.0, 319.5, 239.5, dtype=capi.EA_F64)
for _ in range(5): P.set_now_frame(rgb3)
t0 = time.perf_counter()
for _ in range(20): P.set_now_frame(rgb3)
print('set_now_frame %.3f ms per call' % ((time.perf_counter() - t0) / 20 * 1e3))
P.close()
