"""Times the GPU frame producers (edge-point extractor, DT producer) on the bundled 640x480 frames and on a
2048x1536 synthetic frame, next to the numpy restatement on the host."""
import os, sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi
from oracle import preprocess_np as pp
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/rgbd')
K = (525.0, 525.0, 319.5, 239.5)
rgb1 = pp.load_rgb_as_bgr(os.path.join(G, 'rgb_1.png')); d1 = pp.load_depth_u16(os.path.join(G, 'depth_1.png')); rgb3 = pp.load_rgb_as_bgr(os.path.join(G, 'rgb_3.png'))
def timeit(fn, n=20):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
P = capi.Problem(*K, dtype=capi.EA_F64)
print('640x480  set_ref_frame (H2D + kernels + count readback) %.3f ms' % timeit(lambda: P.set_ref_frame(rgb1, d1)))
print('640x480  set_now_frame (H2D + kernels)                  %.3f ms' % timeit(lambda: P.set_now_frame(rgb3)))
t0 = time.perf_counter(); pp.get_aX(rgb1, d1, *K); t1 = time.perf_counter(); pp.get_distance_transform(rgb3); t2 = time.perf_counter()
print('numpy restatement on the host: get_aX %.1f ms, get_distance_transform %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
q, t, s = P.solve([1, 0, 0, 0], [0, 0, 0]); print('solve', s['why'], s['num_iterations'], '%.3f ms' % s['total_time_ms'])
print('640x480  set_ref_frame_canny (blur, gray, Canny 30/90, compaction)   %.3f ms  (n=%d)' % (timeit(lambda: P.set_ref_frame_canny(rgb1, d1)), P.num_points))
print('640x480  set_now_frame_canny (... + chamfer DT + normalise)          %.3f ms  (hysteresis launches: %d)' % (timeit(lambda: P.set_now_frame_canny(rgb3)), P.set_now_frame_canny(rgb3, debug=True)['hysteresis_launches']))
t0 = time.perf_counter(); pp.get_aX_canny(rgb1, d1, *K); t1 = time.perf_counter(); pp.get_distance_transform2(rgb3); t2 = time.perf_counter()
print('numpy/scipy restatement on the host: get_aX_canny %.1f ms, get_distance_transform2 %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
q, t, s = P.solve([1, 0, 0, 0], [0, 0, 0]); print('solve (Canny inputs)', s['why'], s['num_iterations'], '%.3f ms' % s['total_time_ms'])
half = rgb3[::2, ::2].copy(); dhalf = (d1[::2, ::2].astype(np.float32) / np.float32(5000.0)).copy()
Pr = capi.Problem(262.5, 262.5, 159.75, 119.75, dtype=capi.EA_F64)
print('320x240  set_ref_frame_ros (3-channel L2 Canny 150/100, compaction)      %.3f ms  (n=%d)' % (timeit(lambda: Pr.set_ref_frame_ros(half, dhalf)), Pr.num_points))
print('320x240  set_now_frame_ros (... + exact Euclidean DT + [0,255])          %.3f ms' % timeit(lambda: Pr.set_now_frame_ros(half)))
Pr.close()
rng = np.random.default_rng(0)
big = np.kron(rgb3, np.ones((4, 4, 1), np.uint8))[:1536, :2048].copy()
bigd = np.kron(d1, np.ones((4, 4), np.uint16))[:1536, :2048].copy()
P2 = capi.Problem(1680., 1680., 1023.5, 767.5, dtype=capi.EA_F32)
print('2048x1536 set_ref_frame %.3f ms  (n=%d)' % (timeit(lambda: P2.set_ref_frame(big, bigd), 5), P2.num_points))
print('2048x1536 set_now_frame %.3f ms' % timeit(lambda: P2.set_now_frame(big), 5))
print('2048x1536 set_ref_frame_canny %.3f ms  (n=%d)' % (timeit(lambda: P2.set_ref_frame_canny(big, bigd), 5), P2.num_points))
print('2048x1536 set_now_frame_canny %.3f ms  (hysteresis launches: %d)' % (timeit(lambda: P2.set_now_frame_canny(big), 5), P2.set_now_frame_canny(big, debug=True)['hysteresis_launches']))
P.close(); P2.close()
# frame-to-frame tracker (ea_tracker_*): the whole per-frame pipeline of a sequence -- DT of the new frame, solve of the
# previous frame's points from the last relative pose, the new frame's points -- on the five bundled grabs, cycled
frames = [(pp.load_rgb_as_bgr(os.path.join(G, 'rgb_%d.png' % i)), pp.load_depth_u16(os.path.join(G, 'depth_%d.png' % i))) for i in range(1, 6)]
for flavour, name in ((0, 'Laplacian'), (1, 'Canny')):
    for dtype, dn in ((capi.EA_F64, 'fp64'), (capi.EA_F32, 'fp32')):
        T = capi.Tracker(*K, dtype=dtype, flavour=flavour, loss=(capi.LOSS_CAUCHY, 1.0))
        for bgr, dep in frames: T.push_frame(bgr, dep)
        n, its, t0 = 0, 0, time.perf_counter()
        for rep in range(6):
            for bgr, dep in (frames if rep % 2 == 0 else frames[::-1]):
                q, t, s = T.push_frame(bgr, dep); n += 1; its += s['num_iterations'] if s else 0
        el = (time.perf_counter() - t0) / n
        print('640x480  tracker push_frame, %s flavour, %s: %.3f ms per frame (%.0f frames/s, %.1f LM iterations per frame)' % (name, dn, el * 1e3, 1.0 / el, its / n))
        T.close()
