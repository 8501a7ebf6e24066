"""Stress: create / use / destroy cycles must give device memory back (problems with tile order, batches with
sub-batches, frame producers, tracker)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edge_alignment_amd import capi, synth
cfgs = [synth.config_c2_twin(seed=700 + i, n_points=4000) for i in range(18)]
big = synth.make_problem(480, 640, 210000, 900, 5, 525.0, 525.0, 319.5, 239.5)
rng = np.random.default_rng(0)
bgr = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8); depth = rng.integers(1, 30000, (240, 320)).astype(np.uint16)
def cycle():
    Ps = []
    for cfg in cfgs:
        P = capi.Problem(*cfg['K'], dtype=capi.EA_F32); P.set_points(cfg['xyz']); P.set_dt_grid(cfg['grid']); Ps.append(P)
    B = capi.Batch(Ps)
    q0 = np.tile([1., 0, 0, 0], (18, 1)); t0 = np.zeros((18, 3))
    B.solve(q0, t0); B.set_tuning('solve_streams', 3); B.solve(q0, t0); B.eval(q0, t0)
    B.close()
    for P in Ps: P.close()
    P = capi.Problem(*big['K'], dtype=capi.EA_F32); P.set_points(big['xyz']); P.set_dt_grid(big['grid'])
    P.eval([1., 0, 0, 0], [0, 0, 0]); P.eval_points([1., 0, 0, 0], [0, 0, 0]); P.get_points(); P.close()
    P = capi.Problem(525.0, 525.0, 319.5, 239.5, dtype=capi.EA_F64)
    P.set_ref_frame(bgr, depth); P.set_now_frame(bgr); P.set_ref_frame_canny(bgr, depth); P.set_now_frame_canny(bgr); P.solve([1., 0, 0, 0], [0, 0, 0], max_num_iterations=3)
    P.pixel_cost([1., 0, 0, 0], [0, 0, 0]); P.close()
    # round 2: scaled ROS producers (staging buffer), bench graph, sharded device solve (event + one-row descriptor)
    P = capi.Problem(262.5, 262.5, 159.75, 119.75, dtype=capi.EA_F64)
    P.set_flavour(z_guard=0.0, z_eps=0.001, rot_transposed=True)
    big_bgr = np.repeat(np.repeat(bgr, 2, axis=0), 2, axis=1); big_depth = np.repeat(np.repeat(depth, 2, axis=0), 2, axis=1).astype(np.float32) / 5000.0
    P.set_ref_frame_ros(big_bgr, big_depth, halvings=1); P.set_now_frame_ros(big_bgr, halvings=1)
    sums = torch.zeros(32, dtype=torch.float64, device='cuda')
    P.solve_sharded_device([1., 0, 0, 0], [0, 0, 0], lambda stream: None, sums.data_ptr(), max_num_iterations=3)
    B = capi.Batch([P]); B.eval([1., 0, 0, 0], [0, 0, 0]); B.bench_capture(4); B.bench_steps(4); B.close(); P.close()
for _ in range(3): cycle()
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
for i in range(40): cycle()
torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
print('free before %d MiB, after 40 cycles %d MiB, delta %.1f MiB' % (free0 >> 20, free1 >> 20, (free0 - free1) / 2**20))
# the library keeps freed blocks for the next problem (resource cache): steady state above, everything back on request
capi.load().ea_release_cached_memory()
torch.cuda.synchronize(); free2 = torch.cuda.mem_get_info()[0]
print('after ea_release_cached_memory: %d MiB free (%.1f MiB more than in steady state)' % (free2 >> 20, (free2 - free1) / 2**20))
sys.exit(1 if (free0 - free1 > 64 * 2**20 or free2 < free1) else 0)
