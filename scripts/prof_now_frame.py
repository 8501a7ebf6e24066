"""rocprofv3 --kernel-trace --memory-copy-trace workload: N x one frame producer on a bundled 640x480 frame.
usage: prof_now_frame.py now|ref|now_canny|ref_canny|now_ros|ref_ros"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi
from oracle import preprocess_np as pp
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/rgbd')
rgb3 = pp.load_rgb_as_bgr(os.path.join(G, 'rgb_3.png')); rgb1 = pp.load_rgb_as_bgr(os.path.join(G, 'rgb_1.png')); d1 = pp.load_depth_u16(os.path.join(G, 'depth_1.png'))
half = rgb3[::2, ::2].copy(); dhalf = (d1[::2, ::2].astype(np.float32) / np.float32(5000.0)).copy()
which = sys.argv[1] if len(sys.argv) > 1 else 'now'
P = capi.Problem(525.0, 525.0, 319.5, 239.5, dtype=capi.EA_F64)
fn = {'now': lambda: P.set_now_frame(rgb3), 'ref': lambda: P.set_ref_frame(rgb1, d1),
      'now_canny': lambda: P.set_now_frame_canny(rgb3), 'ref_canny': lambda: P.set_ref_frame_canny(rgb1, d1),
      'now_ros': lambda: P.set_now_frame_ros(half), 'ref_ros': lambda: P.set_ref_frame_ros(half, dhalf)}[which]
for _ in range(5): fn()
t0 = time.perf_counter()
for _ in range(10): fn()
print('%s %.3f ms per call' % (which, (time.perf_counter() - t0) / 10 * 1e3))
P.close()
