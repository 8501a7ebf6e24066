"""Writes the bundled pair (frame 1 points, frame 3 DT) in the examples' binary format and runs examples/facade_timing on it:
the reference's call sequence per frame pair, repeated.   python scripts/facade_timing.py [stride]"""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import preprocess_np as pp   # (image loaders + the restated pre-processing: inputs only)
G = os.path.join(ROOT, "tests", "golden", "rgbd")
K = (525.0, 525.0, 319.5, 239.5)
aX, _ = pp.get_aX(pp.load_rgb_as_bgr(os.path.join(G, "rgb_1.png")), pp.load_depth_u16(os.path.join(G, "depth_1.png")), *K)
grid = pp.grid_view_of_image(pp.get_distance_transform(pp.load_rgb_as_bgr(os.path.join(G, "rgb_3.png"))))
W, H = grid.shape
path = os.path.join(tempfile.gettempdir(), "pair13.bin")
with open(path, "wb") as f:
    f.write(struct.pack("<iii", aX.shape[1], H, W)); f.write(struct.pack("<dddd", *K))
    f.write(np.ascontiguousarray(aX.T, dtype=np.float64).tobytes()); f.write(np.ascontiguousarray(grid, dtype=np.float64).tobytes())
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
for stride in ([int(sys.argv[1])] if len(sys.argv) > 1 else [30, 1]):
    print("stride", stride, flush=True)
    subprocess.check_call([os.path.join(ROOT, "examples", "facade_timing"), path, str(stride), "8"])
