"""rocprofv3 --kernel-trace --memory-copy-trace workload: the frame-to-frame tracker on the bundled frames."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_alignment_amd import capi
from oracle import preprocess_np as pp
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/rgbd')
frames = [(pp.load_rgb_as_bgr(os.path.join(G, 'rgb_%d.png' % i)), pp.load_depth_u16(os.path.join(G, 'depth_%d.png' % i))) for i in range(1, 6)]
T = capi.Tracker(525.0, 525.0, 319.5, 239.5, dtype=capi.EA_F64, flavour=int(sys.argv[1]) if len(sys.argv) > 1 else 0, loss=(capi.LOSS_CAUCHY, 1.0))
for bgr, dep in frames: T.push_frame(bgr, dep)
t0 = time.perf_counter(); n = 0
for rep in range(2):
    for bgr, dep in (frames if rep % 2 == 0 else frames[::-1]):
        T.push_frame(bgr, dep); n += 1
print('push_frame %.3f ms per frame (frames in pageable memory)' % ((time.perf_counter() - t0) / n * 1e3))
# the same frames in page-locked memory (ea_host_alloc): the two uploads per frame become direct DMA
pinned = []
for bgr, dep in frames:
    b = capi.pinned_array(bgr.shape, bgr.dtype); b[...] = bgr
    d = capi.pinned_array(dep.shape, dep.dtype); d[...] = dep
    pinned.append((b, d))
for bgr, dep in pinned: T.push_frame(bgr, dep)
best = 1e9
for rnd in range(3):
    t0 = time.perf_counter(); n = 0
    for rep in range(2):
        for bgr, dep in (pinned if rep % 2 == 0 else pinned[::-1]):
            T.push_frame(bgr, dep); n += 1
    best = min(best, (time.perf_counter() - t0) / n)
print('push_frame %.3f ms per frame (frames in ea_host_alloc memory)' % (best * 1e3))
best = 1e9
for rnd in range(3):
    t0 = time.perf_counter(); n = 0
    for rep in range(2):
        for bgr, dep in (frames if rep % 2 == 0 else frames[::-1]):
            T.push_frame(bgr, dep); n += 1
    best = min(best, (time.perf_counter() - t0) / n)
print('push_frame %.3f ms per frame (pageable again, best of 3 rounds)' % (best * 1e3))
T.close()
