"""ctypes binding of libea_hip.so (include/ea_hip.h) — the Python stub a maintainer of a Python
harness would write; the tests and bench.py drive the C-ABI through it.

There is no CPU fallback: if the shared library is missing or no gfx950 device is present the
calls raise `EAError` (never a silent numpy path).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EA_HIP_LIB") or os.path.join(_HERE, "lib", "libea_hip.so")  # EA_HIP_LIB: A/B against another build

EA_F64, EA_F32 = 0, 1
EA_OK, EA_ERR_INVALID_ARG, EA_ERR_HIP, EA_ERR_NO_DEVICE, EA_ERR_STATE, EA_ERR_ALLOC = 0, -1, -2, -3, -4, -5
LOSS_TRIVIAL, LOSS_CAUCHY, LOSS_HUBER = 0, 1, 2
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
STRATEGY_LM, STRATEGY_DOGLEG = 0, 1
WHY = ["none", "function_tolerance", "gradient_tolerance", "parameter_tolerance",
       "max_iterations", "min_radius", "initial_eval_failed", "too_many_invalid_steps",
       "eval_failed"]
MAX_TRACE = 128

EXPORTED = [
    "ea_last_error", "ea_version", "ea_device_count", "ea_default_options",
    "ea_problem_create", "ea_problem_destroy", "ea_problem_set_points", "ea_problem_set_point_order", "ea_problem_get_point_order",
    "ea_problem_set_points_device", "ea_problem_set_dt", "ea_problem_set_dt_image_device",
    "ea_problem_set_loss", "ea_problem_set_flavour", "ea_problem_num_points",
    "ea_eval", "ea_eval_points", "ea_cost", "ea_problem_pixel_cost", "ea_solve",
    "ea_release_cached_memory", "ea_host_alloc", "ea_host_free", "ea_batch_create", "ea_batch_destroy", "ea_batch_count", "ea_batch_eval", "ea_batch_solve",
    "ea_batch_eval_poses", "ea_batch_set_poses", "ea_batch_eval_resident_poses",
    "ea_solve_pyramid", "ea_solve_sharded", "ea_solve_sharded_device",
    "ea_comm_get_unique_id", "ea_comm_create", "ea_comm_create_all", "ea_comm_destroy", "ea_comm_rank", "ea_comm_size",
    "ea_comm_gather_poses", "ea_solve_sharded_comm", "ea_comm_get_info", "ea_hip_runtime_copies", "ea_tracker_create", "ea_tracker_destroy", "ea_tracker_problem", "ea_tracker_push_frame",
    "ea_batch_row_offsets", "ea_problem_num_rows", "ea_eval_rows", "ea_eval_rows_device", "ea_batch_eval_rows_device", "ea_batch_eval_rows",
    "ea_batch_set_tuning", "ea_batch_get_info", "ea_selftest_wave_reduce",
    "ea_problem_set_ref_frame", "ea_problem_set_ref_frame_masked", "ea_problem_set_now_frame", "ea_problem_debug_now_frame",
    "ea_problem_set_ref_frame_canny", "ea_problem_set_now_frame_canny", "ea_problem_debug_now_frame_canny",
    "ea_problem_set_ref_frame_ros", "ea_problem_set_now_frame_ros", "ea_problem_debug_now_frame_ros",
    "ea_problem_set_ref_frame_ros_scaled", "ea_problem_set_now_frame_ros_scaled", "ea_resize_half",
    "ea_problem_get_points", "ea_problem_get_dt",
    "ea_problem_set_distortion", "ea_problem_set_second_camera", "ea_problem_add_term", "ea_problem_clear_terms",
]

# measurement hooks (edge_alignment_amd/csrc/ea_hip_dev.h): bound by bench.py, the A/B scripts and the tests that pin the
# launch patterns; not part of the drop-in boundary
EXPORTED_DEV = [
    "ea_batch_bench_eval", "ea_batch_bench_steps", "ea_batch_bench_capture", "ea_batch_bench_capture_pipelined", "ea_batch_bench_steps_riding",
    "ea_batch_bench_result", "ea_batch_bench_result_riding", "ea_batch_bench_kernel", "ea_batch_bench_rows", "ea_batch_bench_fold",
    "ea_batch_bench_resident_poses", "ea_bench_graph_floor",
]


class EAError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libea_hip error %d: %s" % (code, msg))
        self.code = code


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
DEVICE_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("max_num_consecutive_invalid_steps", C.c_int),
                ("jacobi_scaling", C.c_int), ("strategy", C.c_int),
                ("minimizer_progress_to_stdout", C.c_int), ("iterations_per_sync", C.c_int),
                ("solve_timeout_ms", C.c_double)]


class PixelCost(C.Structure):
    _fields_ = [("total_cost", C.c_double), ("mean_cost", C.c_double), ("max_cost", C.c_double),
                ("max_pixel", C.c_double * 2), ("count", C.c_int64), ("outside", C.c_int64)]


class Summary(C.Structure):
    _fields_ = [("termination", C.c_int), ("why", C.c_int), ("num_iterations", C.c_int),
                ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("num_point_evals", C.c_int64), ("total_time_ms", C.c_double),
                ("it_cost", C.c_double * MAX_TRACE), ("it_cost_change", C.c_double * MAX_TRACE),
                ("it_gradient_max_norm", C.c_double * MAX_TRACE),
                ("it_step_norm", C.c_double * MAX_TRACE),
                ("it_relative_decrease", C.c_double * MAX_TRACE),
                ("it_radius", C.c_double * MAX_TRACE), ("it_successful", C.c_int * MAX_TRACE)]


_lib = None


def load():
    """dlopen libea_hip.so and declare every prototype of include/ea_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EAError(-3, "%s not built — run `python -c 'import __graft_entry__ as g; g.build()'`; "
                          "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    dp, vp = C.POINTER(C.c_double), C.c_void_p
    i64p = C.POINTER(C.c_int64)
    L.ea_last_error.restype = C.c_char_p
    L.ea_version.restype = C.c_char_p
    L.ea_device_count.argtypes = [C.POINTER(C.c_int)]
    L.ea_default_options.argtypes = [C.POINTER(Options)]
    L.ea_default_options.restype = None
    L.ea_problem_create.argtypes = [C.POINTER(vp), C.POINTER(Camera), C.c_int, C.c_int]
    L.ea_problem_destroy.argtypes = [vp]
    L.ea_problem_destroy.restype = None
    L.ea_problem_set_points.argtypes = [vp, dp, C.c_int64, C.c_int64]
    L.ea_problem_set_points_device.argtypes = [vp, vp, vp, vp, C.c_int64]
    L.ea_problem_set_point_order.argtypes = [vp, C.c_int]
    L.ea_problem_get_point_order.argtypes = [vp, C.POINTER(C.c_int)]
    L.ea_problem_set_dt.argtypes = [vp, dp, C.c_int, C.c_int]
    L.ea_problem_set_dt_image_device.argtypes = [vp, vp, C.c_int, C.c_int]
    L.ea_problem_set_loss.argtypes = [vp, C.c_int, C.c_double]
    L.ea_problem_set_flavour.argtypes = [vp, C.c_double, C.c_double, C.c_int]
    L.ea_problem_num_points.argtypes = [vp]
    L.ea_problem_num_points.restype = C.c_int64
    L.ea_eval.argtypes = [vp, dp, dp, dp, dp, dp, i64p]
    L.ea_eval_points.argtypes = [vp, dp, dp, dp, dp, C.c_int]
    L.ea_cost.argtypes = [vp, dp, dp, dp, i64p]
    L.ea_solve.argtypes = [vp, C.POINTER(Options), dp, dp, C.POINTER(Summary)]
    L.ea_batch_create.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int]
    L.ea_batch_destroy.argtypes = [vp]
    L.ea_batch_destroy.restype = None
    L.ea_batch_count.argtypes = [vp]
    L.ea_batch_eval.argtypes = [vp, dp, dp, dp, dp, dp, i64p]
    L.ea_batch_solve.argtypes = [vp, C.POINTER(Options), dp, dp, C.POINTER(Summary)]
    L.ea_batch_eval_poses.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, i64p]
    L.ea_batch_set_poses.argtypes = [vp, C.c_int, dp, dp]
    L.ea_batch_eval_resident_poses.argtypes = [vp, dp, dp, dp, i64p]
    L.ea_batch_bench_eval.argtypes = [vp, dp, dp, C.c_int, C.c_int, dp, dp]
    L.ea_batch_bench_steps.argtypes = [vp, C.c_int, dp]
    L.ea_batch_bench_capture.argtypes = [vp, C.c_int]
    L.ea_batch_bench_capture_pipelined.argtypes = [vp, C.c_int]
    L.ea_batch_bench_steps_riding.argtypes = [vp, C.c_int, dp]
    L.ea_batch_bench_result.argtypes = [vp, dp, dp, dp, i64p]
    L.ea_batch_bench_result_riding.argtypes = [vp, dp, dp, dp, i64p]
    L.ea_batch_bench_kernel.argtypes = [vp, dp, dp, C.c_int, C.c_int, dp]
    L.ea_batch_bench_fold.argtypes = [vp, C.c_int, C.c_int, dp]
    L.ea_batch_bench_resident_poses.argtypes = [vp, C.c_int, C.c_int, dp, C.POINTER(C.c_int)]
    L.ea_bench_graph_floor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp]
    L.ea_batch_row_offsets.argtypes = [vp, i64p]
    L.ea_problem_num_rows.argtypes = [vp, i64p]
    L.ea_eval_rows.argtypes = [vp, dp, dp, C.c_int, C.c_int, vp, vp, C.c_int64, i64p]
    L.ea_eval_rows_device.argtypes = [vp, dp, dp, C.c_int, C.c_int, vp, vp, C.c_int64, i64p]
    L.ea_batch_eval_rows_device.argtypes = [vp, dp, dp, C.c_int, C.c_int, vp, vp, C.c_int64, i64p]
    L.ea_batch_eval_rows.argtypes = [vp, dp, dp, C.c_int, C.c_int, vp, vp, C.c_int64, i64p]
    L.ea_batch_bench_rows.argtypes = [vp, dp, dp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int64, C.c_int, C.c_int, dp]
    L.ea_problem_pixel_cost.argtypes = [vp, dp, dp, C.POINTER(PixelCost)]
    L.ea_solve_sharded.argtypes = [vp, C.POINTER(Options), ALLREDUCE_FN, vp, dp, dp, C.POINTER(Summary)]
    L.ea_solve_sharded_device.argtypes = [vp, C.POINTER(Options), DEVICE_ALLREDUCE_FN, vp, vp, dp, dp, C.POINTER(Summary)]
    L.ea_solve_pyramid.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Options), dp, dp, C.POINTER(Summary)]
    L.ea_comm_get_unique_id.argtypes = [C.c_char_p]
    L.ea_comm_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.ea_comm_create_all.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    L.ea_comm_destroy.argtypes = [vp]
    L.ea_comm_destroy.restype = None
    L.ea_comm_rank.argtypes = [vp]
    L.ea_comm_size.argtypes = [vp]
    L.ea_comm_gather_poses.argtypes = [vp, vp, dp, dp, C.POINTER(C.c_int), C.c_int, dp, dp, C.POINTER(C.c_int)]
    L.ea_solve_sharded_comm.argtypes = [vp, C.POINTER(Options), vp, dp, dp, C.POINTER(Summary)]
    L.ea_comm_get_info.argtypes = [vp, C.c_char_p, i64p]
    L.ea_batch_set_tuning.argtypes = [vp, C.c_char_p, C.c_int]
    L.ea_batch_get_info.argtypes = [vp, C.c_char_p, i64p]
    L.ea_selftest_wave_reduce.argtypes = [C.c_int, C.POINTER(C.c_float), dp, dp, C.POINTER(C.c_float)]
    u8p, u16p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)
    L.ea_tracker_create.argtypes = [C.POINTER(vp), C.POINTER(Camera), C.c_int, C.c_int, C.c_int]
    L.ea_tracker_destroy.argtypes = [vp]
    L.ea_tracker_destroy.restype = None
    L.ea_tracker_problem.argtypes = [vp]
    L.ea_tracker_problem.restype = vp
    L.ea_tracker_push_frame.argtypes = [vp, u8p, u16p, C.c_int, C.c_int, C.c_double, C.POINTER(Options), dp, dp,
                                        C.POINTER(Summary), C.POINTER(C.c_int)]
    L.ea_problem_set_ref_frame.argtypes = [vp, u8p, u16p, C.c_int, C.c_int, C.c_double, C.c_int]
    L.ea_problem_set_now_frame.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ea_problem_set_ref_frame_masked.argtypes = [vp, u8p, u8p, u16p, C.c_int, C.c_int, C.c_double, C.c_int]
    L.ea_problem_set_ref_frame_ros.argtypes = [vp, u8p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double, C.c_double]
    L.ea_problem_set_now_frame_ros.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_double, C.c_double]
    L.ea_problem_set_ref_frame_ros_scaled.argtypes = [vp, u8p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    L.ea_problem_set_now_frame_ros_scaled.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    L.ea_resize_half.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]
    L.ea_problem_debug_now_frame_ros.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_double, C.c_double, u8p, C.POINTER(C.c_float)]
    L.ea_problem_set_ref_frame_canny.argtypes = [vp, u8p, C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    L.ea_problem_set_now_frame_canny.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]
    L.ea_problem_debug_now_frame_canny.argtypes = [vp, u8p, u8p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double,
                                                   C.c_double, u8p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.ea_problem_debug_now_frame.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    L.ea_problem_get_points.argtypes = [vp, dp, C.c_int64]
    L.ea_problem_get_dt.argtypes = [vp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ea_problem_set_distortion.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]
    L.ea_problem_set_second_camera.argtypes = [vp, dp, dp]
    L.ea_problem_add_term.argtypes = [vp, vp]
    L.ea_problem_clear_terms.argtypes = [vp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise EAError(rc, load().ea_last_error().decode())


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int(0)
    rc = load().ea_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def default_options(**kw):
    o = Options()
    load().ea_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


_TRACE_FIELDS = ("it_cost", "it_cost_change", "it_gradient_max_norm", "it_step_norm", "it_relative_decrease", "it_radius")
_TRACE_OFF = Summary.it_cost.offset // 8


def summary_to_dict(s):
    ni = min(s.num_iterations + 1, MAX_TRACE)
    # one zero-copy view of the struct's six double[MAX_TRACE] arrays instead of six ctypes slices
    tr = np.frombuffer(s, dtype=np.float64, count=6 * MAX_TRACE, offset=8 * _TRACE_OFF).reshape(6, MAX_TRACE)[:, :ni].copy()
    d = dict(termination=s.termination, why=WHY[s.why], num_iterations=s.num_iterations,
             num_successful_steps=s.num_successful_steps,
             num_unsuccessful_steps=s.num_unsuccessful_steps,
             initial_cost=s.initial_cost, final_cost=s.final_cost,
             num_point_evals=s.num_point_evals, total_time_ms=s.total_time_ms,
             it_successful=np.frombuffer(s, dtype=np.int32, count=ni, offset=Summary.it_successful.offset).copy())
    for k, name in enumerate(_TRACE_FIELDS):
        d[name] = tr[k]
    return d


class Problem:
    """One frame pair: the N residual blocks + interpolator + loss of the reference's set-up
    block (standalone_edge_align.cpp:256-278), resident in HBM."""

    def __init__(self, fx, fy, cx, cy, dtype=EA_F64, device=0):
        self._h = C.c_void_p()
        cam = Camera(fx, fy, cx, cy)
        _check(load().ea_problem_create(C.byref(self._h), C.byref(cam), dtype, device))
        self.dtype = dtype
        self.device = device
        self._keep = []

    def close(self):
        if self._h:
            load().ea_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_points(self, xyz):
        xyz = _f64(xyz)
        if xyz.ndim != 2 or xyz.shape[1] < 3:
            raise ValueError("xyz must be (n, >=3)")
        _check(load().ea_problem_set_points(self._h, _dp(xyz), xyz.shape[0], xyz.shape[1]))

    def set_point_order(self, tile_px):
        """storage order for the next set_points: tiles of tile_px pixels (> 0), the caller's order (0), automatic (< 0)"""
        _check(load().ea_problem_set_point_order(self._h, int(tile_px)))

    @property
    def point_order(self):
        v = C.c_int()
        _check(load().ea_problem_get_point_order(self._h, C.byref(v)))
        return v.value

    def set_points_device(self, x_ptr, y_ptr, z_ptr, n):
        _check(load().ea_problem_set_points_device(self._h, x_ptr, y_ptr, z_ptr, n))

    def set_dt_grid(self, grid):
        """grid: the Grid2D view (rows = u extent, cols = v extent), row-major float64."""
        grid = _f64(grid)
        _check(load().ea_problem_set_dt(self._h, _dp(grid), grid.shape[0], grid.shape[1]))

    def set_dt_image_device(self, ptr, height, width):
        _check(load().ea_problem_set_dt_image_device(self._h, ptr, height, width))

    def set_ref_frame(self, bgr, depth_u16, z_scaling=5000.0, threshold=35, mask=None):
        """get_aX (get_aX_mask with a mask) on the GPU: bgr (H,W,3) uint8 as cv::imread returns it, depth (H,W) uint16"""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth_u16 = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        H, W = depth_u16.shape
        assert bgr.shape == (H, W, 3)
        if mask is not None:
            mk = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mk.shape == (H, W)
            _check(load().ea_problem_set_ref_frame_masked(self._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                          mk.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                          depth_u16.ctypes.data_as(C.POINTER(C.c_uint16)), H, W, z_scaling, threshold))
            return
        _check(load().ea_problem_set_ref_frame(self._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               depth_u16.ctypes.data_as(C.POINTER(C.c_uint16)), H, W, z_scaling, threshold))

    def set_now_frame(self, bgr, threshold=35, median=True, normalize=True, debug=False):
        """get_distance_transform on the GPU, written straight into the problem's DT image"""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        H, W = bgr.shape[:2]
        u8 = C.POINTER(C.c_uint8)
        if not debug:
            _check(load().ea_problem_set_now_frame(self._h, bgr.ctypes.data_as(u8), H, W, threshold, int(median), int(normalize)))
            return None
        lap = np.zeros((H, W), np.uint8); mask = np.zeros((H, W), np.uint8)
        cham = np.zeros((H, W), np.int32); dt = np.zeros((H, W), np.float32)
        _check(load().ea_problem_debug_now_frame(self._h, bgr.ctypes.data_as(u8), H, W, threshold, int(median), int(normalize),
                                                 lap.ctypes.data_as(u8), mask.ctypes.data_as(u8),
                                                 cham.ctypes.data_as(C.POINTER(C.c_int32)), dt.ctypes.data_as(C.POINTER(C.c_float))))
        return dict(lap=lap, mask=mask, chamfer=cham, dt=dt)

    def set_ref_frame_canny(self, bgr, depth_u16, z_scaling=5000.0, low=30.0, high=90.0):
        """get_aX_canny on the GPU (ref: utils.cpp:371-462)"""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth_u16 = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        H, W = depth_u16.shape
        assert bgr.shape == (H, W, 3)
        _check(load().ea_problem_set_ref_frame_canny(self._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                     depth_u16.ctypes.data_as(C.POINTER(C.c_uint16)), H, W, z_scaling, low, high))

    def set_now_frame_canny(self, bgr, mask=None, low=30.0, high=90.0, normalize=(0.0, 1.0), debug=False):
        """get_distance_transform2[_masked][_NoNormalize] on the GPU (ref: utils.cpp:85-199); normalize: None or (lo, hi)"""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        H, W = bgr.shape[:2]
        u8 = C.POINTER(C.c_uint8)
        mk = None
        if mask is not None:
            mk = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mk.shape == (H, W)
        mp = mk.ctypes.data_as(u8) if mk is not None else None
        do_norm, (lo, hi) = (0, (0.0, 1.0)) if normalize is None else (1, normalize)
        if not debug:
            _check(load().ea_problem_set_now_frame_canny(self._h, bgr.ctypes.data_as(u8), mp, H, W, low, high, do_norm, lo, hi))
            return None
        edges = np.zeros((H, W), np.uint8); cham = np.zeros((H, W), np.int32); dt = np.zeros((H, W), np.float32)
        rounds = C.c_int()
        _check(load().ea_problem_debug_now_frame_canny(self._h, bgr.ctypes.data_as(u8), mp, H, W, low, high, do_norm, lo, hi,
                                                       edges.ctypes.data_as(u8), cham.ctypes.data_as(C.POINTER(C.c_int32)),
                                                       dt.ctypes.data_as(C.POINTER(C.c_float)), C.byref(rounds)))
        return dict(edges=edges, chamfer=cham, dt=dt, hysteresis_launches=rounds.value)

    def set_ref_frame_ros(self, bgr, depth_f32, t1=150.0, t2=100.0, halvings=0):
        """SolveEA::setRefFrame on the GPU (ref: src/SolveEA.cpp:29-82); halvings > 0: the frames are full resolution and
        the node's cv::resize x0.5 (NaN -> 0 on depth first, src/ea.cpp:38, :56-62) runs on the device that many times"""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth_f32 = np.ascontiguousarray(depth_f32, dtype=np.float32)
        H, W = depth_f32.shape
        assert bgr.shape == (H, W, 3)
        _check(load().ea_problem_set_ref_frame_ros_scaled(self._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                          depth_f32.ctypes.data_as(C.POINTER(C.c_float)), H, W, int(halvings), t1, t2))

    def set_now_frame_ros(self, bgr, t1=150.0, t2=100.0, debug=False, halvings=0):
        """SolveEA::setNowFrame on the GPU (ref: src/SolveEA.cpp:86-119); halvings as in set_ref_frame_ros"""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        H, W = bgr.shape[:2]
        u8 = C.POINTER(C.c_uint8)
        if halvings:
            assert not debug
            _check(load().ea_problem_set_now_frame_ros_scaled(self._h, bgr.ctypes.data_as(u8), H, W, int(halvings), t1, t2))
            return None
        if not debug:
            _check(load().ea_problem_set_now_frame_ros(self._h, bgr.ctypes.data_as(u8), H, W, t1, t2))
            return None
        edges = np.zeros((H, W), np.uint8); dt = np.zeros((H, W), np.float32)
        _check(load().ea_problem_debug_now_frame_ros(self._h, bgr.ctypes.data_as(u8), H, W, t1, t2, edges.ctypes.data_as(u8),
                                                     dt.ctypes.data_as(C.POINTER(C.c_float))))
        return dict(edges=edges, dt=dt)

    def solve_sharded(self, q, t, allreduce, **opts):
        """ea_solve_sharded: `allreduce(array_of_32_doubles)` sums its argument in place over all ranks"""
        q, t = _f64(q).copy(), _f64(t).copy()
        o = default_options(**opts)
        s = Summary()

        def _cb(buf, count, _user):
            try:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                allreduce(a)
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLREDUCE_FN(_cb)
        _check(load().ea_solve_sharded(self._h, C.byref(o), cb, None, _dp(q), _dp(t), C.byref(s)))
        return q, t, summary_to_dict(s)

    def solve_sharded_device(self, q, t, enqueue_allreduce, device_sums_ptr, **opts):
        """ea_solve_sharded_device: `device_sums_ptr` = address of 32 doubles of device memory the caller owns;
        `enqueue_allreduce(stream_ptr)` must enqueue, on that HIP stream, the in-place sum of those 32 doubles over all
        ranks and return without waiting (edge_alignment_amd.dist.make_device_allreduce)."""
        q = _f64(q).reshape(4).copy()
        t = _f64(t).reshape(3).copy()
        o = default_options(**opts)
        s = Summary()

        def _cb(_buf, _count, stream, _user):
            try:
                enqueue_allreduce(stream or 0)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = DEVICE_ALLREDUCE_FN(_cb)
        _check(load().ea_solve_sharded_device(self._h, C.byref(o), cb, None, C.c_void_p(int(device_sums_ptr)), _dp(q), _dp(t),
                                              C.byref(s)))
        return q, t, summary_to_dict(s)

    def solve_sharded_rows(self, q, t, enqueue_allreduce, agree, **opts):
        """The one-launch-per-iteration form of the point-sharded solve with the exchange supplied by the caller (what
        ea_solve_sharded_comm does with RCCL; this binding exists so that the protocol can be rehearsed over gloo):
        `enqueue_allreduce(ptr, count, stream_ptr)` enqueues the in-place sum over all ranks of `count` doubles of device
        memory at `ptr`; `agree([cannot, rows]) -> [max, max]` takes two ints to their maximum over the ranks.
        Returns (q, t, summary, used): used == 0 means some rank's shard does not qualify and nothing was solved."""
        q = _f64(q).reshape(4).copy()
        t = _f64(t).reshape(3).copy()
        o = default_options(**opts)
        s = Summary()
        used = C.c_int()

        def _cb(buf, count, stream, _user):
            try:
                enqueue_allreduce(int(buf or 0), int(count), int(stream or 0))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def _agree(vals, _user):
            try:
                out = agree([int(vals[0]), int(vals[1])])
                vals[0], vals[1] = int(out[0]), int(out[1])
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        cb = DEVICE_ALLREDUCE_FN(_cb)
        ag = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int), C.c_void_p)(_agree)
        L = load()
        L.ea_internal_solve_sharded_rows.restype = C.c_int
        L.ea_internal_solve_sharded_rows.argtypes = [C.c_void_p, C.c_void_p, DEVICE_ALLREDUCE_FN, type(ag), C.c_void_p, C.POINTER(C.c_double),
                                                     C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int)]
        _check(L.ea_internal_solve_sharded_rows(self._h, C.cast(C.byref(o), C.c_void_p), cb, ag, None, _dp(q), _dp(t),
                                                C.cast(C.byref(s), C.c_void_p), C.byref(used)))
        return q, t, summary_to_dict(s), used.value

    def solve_sharded_comm(self, q, t, comm, **opts):
        """ea_solve_sharded_comm: the point-sharded solve with ncclAllReduce enqueued by the library itself (no callback)"""
        q = _f64(q).reshape(4).copy()
        t = _f64(t).reshape(3).copy()
        o = default_options(**opts)
        s = Summary()
        _check(load().ea_solve_sharded_comm(self._h, C.byref(o), comm._h, _dp(q), _dp(t), C.byref(s)))
        return q, t, summary_to_dict(s)

    def pixel_cost(self, q, t):
        """the reference's integer-pixel cost report (standalone_edge_align.cpp:2494-2567) at pose (q, t)"""
        q, t = _f64(q).reshape(4), _f64(t).reshape(3)
        pc = PixelCost()
        _check(load().ea_problem_pixel_cost(self._h, _dp(q), _dp(t), C.byref(pc)))
        return dict(total_cost=pc.total_cost, mean_cost=pc.mean_cost, max_cost=pc.max_cost,
                    max_pixel=(pc.max_pixel[0], pc.max_pixel[1]), count=pc.count, outside=pc.outside)

    def get_points(self):
        n = self.num_points
        xyz = np.zeros((n, 3))
        _check(load().ea_problem_get_points(self._h, _dp(xyz), n))
        return xyz

    def get_dt(self):
        h, w = C.c_int(), C.c_int()
        _check(load().ea_problem_get_dt(self._h, None, C.byref(h), C.byref(w)))
        img = np.zeros((h.value, w.value))
        _check(load().ea_problem_get_dt(self._h, _dp(img), None, None))
        return img

    def set_distortion(self, k1, k2, p1, p2, k3):
        _check(load().ea_problem_set_distortion(self._h, k1, k2, p1, p2, k3))

    def set_second_camera(self, T12, T12inv=None):
        T12 = _f64(T12).reshape(16)
        T12inv = _f64(T12inv if T12inv is not None else np.linalg.inv(T12.reshape(4, 4))).reshape(16)
        _check(load().ea_problem_set_second_camera(self._h, _dp(T12), _dp(T12inv)))

    def add_term(self, term):
        self._keep.append(term)
        _check(load().ea_problem_add_term(self._h, term.handle))

    def clear_terms(self):
        _check(load().ea_problem_clear_terms(self._h))
        self._keep = []

    def set_loss(self, kind, a=1.0):
        _check(load().ea_problem_set_loss(self._h, kind, a))

    def set_flavour(self, z_guard=0.01, z_eps=0.0, rot_transposed=False):
        _check(load().ea_problem_set_flavour(self._h, z_guard, z_eps, int(rot_transposed)))

    @property
    def num_points(self):
        return load().ea_problem_num_points(self._h)

    def eval(self, q, t):
        q, t = _f64(q), _f64(t)
        cost = C.c_double()
        JtJ, Jtr = np.zeros((6, 6)), np.zeros(6)
        bad = C.c_int64()
        _check(load().ea_eval(self._h, _dp(q), _dp(t), C.byref(cost), _dp(JtJ), _dp(Jtr),
                              C.byref(bad)))
        return dict(cost=cost.value, JtJ=JtJ, Jtr=Jtr, n_invalid=bad.value)

    def eval_points(self, q, t, corrected=True):
        q, t = _f64(q), _f64(t)
        n = self.num_points
        r, J = np.zeros(n), np.zeros((n, 6))
        _check(load().ea_eval_points(self._h, _dp(q), _dp(t), _dp(r), _dp(J), int(corrected)))
        return r, J

    def eval_rows(self, q, t, corrected=True, layout=0):
        """materialised mode: r [rows], J [rows, 6] (layout 0) or [6, rows] (layout 1) in the problem's dtype, n_invalid"""
        q, t = _f64(q), _f64(t)
        n = C.c_int64()
        _check(load().ea_problem_num_rows(self._h, C.byref(n)))
        dt = np.float32 if self.dtype == EA_F32 else np.float64
        r = np.zeros(n.value, dtype=dt)
        J = np.zeros((n.value, 6) if layout == 0 else (6, n.value), dtype=dt)
        bad = C.c_int64()
        _check(load().ea_eval_rows(self._h, _dp(q), _dp(t), int(corrected), int(layout), r.ctypes.data_as(C.c_void_p),
                                   J.ctypes.data_as(C.c_void_p), n.value, C.byref(bad)))
        return r, J, bad.value

    def cost(self, q, t):
        q, t = _f64(q), _f64(t)
        cost, bad = C.c_double(), C.c_int64()
        _check(load().ea_cost(self._h, _dp(q), _dp(t), C.byref(cost), C.byref(bad)))
        return cost.value, bad.value

    def solve(self, q, t, **opts):
        q, t = _f64(q).copy(), _f64(t).copy()
        o = default_options(**opts)
        s = Summary()
        _check(load().ea_solve(self._h, C.byref(o), _dp(q), _dp(t), C.byref(s)))
        return q, t, summary_to_dict(s)


class Tracker:
    """frame-to-frame driver: push_frame aligns the previous frame's edge points against the new frame"""

    def __init__(self, fx, fy, cx, cy, dtype=EA_F64, device=0, flavour=0, loss=None):
        self._h = C.c_void_p()
        cam = Camera(fx, fy, cx, cy)
        _check(load().ea_tracker_create(C.byref(self._h), C.byref(cam), dtype, device, flavour))
        if loss is not None:
            _check(load().ea_problem_set_loss(load().ea_tracker_problem(self._h), loss[0], loss[1]))

    def push_frame(self, bgr, depth_u16, z_scaling=5000.0, **opts):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth_u16 = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        H, W = depth_u16.shape
        q, t = np.zeros(4), np.zeros(3)
        o = default_options(**opts)
        s = Summary()
        aligned = C.c_int()
        _check(load().ea_tracker_push_frame(self._h, bgr.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            depth_u16.ctypes.data_as(C.POINTER(C.c_uint16)), H, W, z_scaling, C.byref(o),
                                            _dp(q), _dp(t), C.byref(s), C.byref(aligned)))
        return q, t, (summary_to_dict(s) if aligned.value else None)

    def close(self):
        if self._h:
            load().ea_tracker_destroy(self._h)
            self._h = C.c_void_p()


def solve_pyramid(levels, q, t, **opts):
    """coarse-to-fine over complete per-level problems, levels[0] = finest; returns q, t, [summary per level]"""
    q, t = _f64(q).copy(), _f64(t).copy()
    o = default_options(**opts)
    hs = (C.c_void_p * len(levels))(*[p._h for p in levels])
    s = (Summary * len(levels))()
    _check(load().ea_solve_pyramid(hs, len(levels), C.byref(o), _dp(q), _dp(t), s))
    return q, t, [summary_to_dict(x) for x in s]


def runtime_copies():
    """number of HIP runtimes mapped in this process (ea_hip_runtime_copies): 1 when torch was imported before this library"""
    return load().ea_hip_runtime_copies()


COMM_ID_BYTES = 128


def comm_unique_id():
    """the 128 bytes rank 0 hands to the other ranks (ncclGetUniqueId)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().ea_comm_get_unique_id(buf))
    return buf.raw


class Comm:
    """One RCCL communicator rank (ea_comm): the pose gather of the batch mode and the all-reduce of the point-sharded solve,
    issued by the library from librccl directly."""

    def __init__(self, unique_id, nranks, rank, device=0):
        self._h = C.c_void_p()
        assert len(unique_id) == COMM_ID_BYTES
        _check(load().ea_comm_create(C.byref(self._h), unique_id, int(nranks), int(rank), int(device)))
        self.nranks, self.rank = int(nranks), int(rank)

    @classmethod
    def from_process_group(cls, device=0):
        """inside an initialised torch.distributed group (any backend): rank 0's id broadcast as an object"""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(box[0], world, rank, device)

    def close(self):
        if self._h:
            load().ea_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather_poses(self, q, t, status=None, after=None):
        """-> (q (nranks * m, 4), t (nranks * m, 3), status (nranks * m,)) in rank order; `after`: the Batch that solved them"""
        q, t = _f64(q).reshape(-1, 4), _f64(t).reshape(-1, 3)
        m = q.shape[0]
        st = np.ascontiguousarray(status if status is not None else np.zeros(m), dtype=np.int32)
        qa, ta, sa = np.zeros((self.nranks * m, 4)), np.zeros((self.nranks * m, 3)), np.zeros(self.nranks * m, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        _check(load().ea_comm_gather_poses(self._h, after._h if after is not None else None, _dp(q), _dp(t), st.ctypes.data_as(ip), m,
                                           _dp(qa), _dp(ta), sa.ctypes.data_as(ip)))
        return qa, ta, sa

    def info(self, key):
        v = C.c_int64()
        _check(load().ea_comm_get_info(self._h, key.encode(), C.byref(v)))
        return v.value


class _Pinned:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            load().ea_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_array(shape, dtype, device=0):
    """a numpy array in page-locked host memory (ea_host_alloc): frames handed to the producers / the tracker from such an
    array go up as direct DMA.  The memory is freed when the array (and every view of it) is gone."""
    L = load()
    L.ea_host_alloc.restype = C.c_void_p
    L.ea_host_alloc.argtypes = [C.c_size_t, C.c_int]
    L.ea_host_free.restype = None
    L.ea_host_free.argtypes = [C.c_void_p]
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = L.ea_host_alloc(max(n, 1), int(device))
    if not ptr:
        raise EAError(-1, L.ea_last_error().decode())
    owner = _Pinned(ptr)
    buf = (C.c_uint8 * max(n, 1)).from_address(ptr)
    buf._owner = owner   # keeps the block alive as long as the ctypes buffer (the array's base) lives
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def graph_floor_ms(device=0, nodes=200, grid=196, block=256):
    """ms per kernel node of a replayed hipGraph of EMPTY kernels (ea_bench_graph_floor): the launch mechanism's floor"""
    ms = C.c_double()
    _check(load().ea_bench_graph_floor(int(device), int(nodes), int(grid), int(block), C.byref(ms)))
    return ms.value


class Batch:
    """Several independent frame pairs evaluated / solved by the same launch sequence."""

    def __init__(self, problems):
        self.problems = list(problems)
        self.dtype = self.problems[0].dtype if self.problems else EA_F64
        arr = (C.c_void_p * len(self.problems))(*[p.handle for p in self.problems])
        self._h = C.c_void_p()
        _check(load().ea_batch_create(C.byref(self._h), arr, len(self.problems)))

    def close(self):
        if self._h:
            load().ea_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.problems)

    def eval(self, q, t):
        q, t = _f64(q).reshape(-1, 4), _f64(t).reshape(-1, 3)
        n = len(self)
        cost, JtJ, Jtr = np.zeros(n), np.zeros((n, 6, 6)), np.zeros((n, 6))
        bad = np.zeros(n, dtype=np.int64)
        _check(load().ea_batch_eval(self._h, _dp(q), _dp(t), _dp(cost), _dp(JtJ), _dp(Jtr),
                                    bad.ctypes.data_as(C.POINTER(C.c_int64))))
        return dict(cost=cost, JtJ=JtJ, Jtr=Jtr, n_invalid=bad)

    def _pose_outputs(self, K):
        n = len(self)
        return (np.zeros((K, n)), np.zeros((K, n, 6, 6)), np.zeros((K, n, 6)), np.zeros((K, n), dtype=np.int64))

    def eval_poses(self, q, t):
        """K evaluations of every problem at K different poses in one call (ea_batch_eval_poses): q (K, n, 4), t (K, n, 3)
        -> dict of cost (K, n), JtJ (K, n, 6, 6), Jtr (K, n, 6), n_invalid (K, n)"""
        n = len(self)
        q, t = _f64(q).reshape(-1, n, 4), _f64(t).reshape(-1, n, 3)
        K = q.shape[0]
        assert t.shape[0] == K
        cost, JtJ, Jtr, bad = self._pose_outputs(K)
        _check(load().ea_batch_eval_poses(self._h, K, _dp(q), _dp(t), _dp(cost), _dp(JtJ), _dp(Jtr),
                                          bad.ctypes.data_as(C.POINTER(C.c_int64))))
        self._resident_K = K
        return dict(cost=cost, JtJ=JtJ, Jtr=Jtr, n_invalid=bad)

    def set_poses(self, q, t):
        """make K poses per problem resident in HBM (ea_batch_set_poses); returns K"""
        n = len(self)
        q, t = _f64(q).reshape(-1, n, 4), _f64(t).reshape(-1, n, 3)
        assert q.shape[0] == t.shape[0]
        _check(load().ea_batch_set_poses(self._h, q.shape[0], _dp(q), _dp(t)))
        self._resident_K = q.shape[0]
        return q.shape[0]

    def eval_resident_poses(self, out=None, fetch=True):
        """the K evaluations of the resident poses (ea_batch_eval_resident_poses); `out`: arrays of a previous call to fill
        again (no allocation inside a timed region); fetch=False: run and synchronise, hand nothing back"""
        if not fetch:
            _check(load().ea_batch_eval_resident_poses(self._h, None, None, None, None))
            return None
        if out is None:
            cost, JtJ, Jtr, bad = self._pose_outputs(self._resident_K)
            out = dict(cost=cost, JtJ=JtJ, Jtr=Jtr, n_invalid=bad)
        ptrs = out.get("_ptrs")
        if ptrs is None:   # (the ctypes pointer objects of the four arrays, built once: ~2 us each -- as much as two 5e4-point evaluations)
            ptrs = out["_ptrs"] = (_dp(out["cost"]), _dp(out["JtJ"]), _dp(out["Jtr"]), out["n_invalid"].ctypes.data_as(C.POINTER(C.c_int64)))
        _check(load().ea_batch_eval_resident_poses(self._h, *ptrs))
        return out

    def bench_resident_poses(self, reps, evaluations_only=False):
        """(ms per run of the resident poses' launches -- one event pair on the batch's stream around `reps` runs --,
        evaluation launches per run); evaluations_only: without the fold launches"""
        ms, n = C.c_double(), C.c_int()
        _check(load().ea_batch_bench_resident_poses(self._h, int(reps), int(evaluations_only), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def solve(self, q, t, **opts):
        q = _f64(q).reshape(-1, 4).copy()
        t = _f64(t).reshape(-1, 3).copy()
        o = default_options(**opts)
        s = (Summary * len(self))()
        _check(load().ea_batch_solve(self._h, C.byref(o), _dp(q), _dp(t), s))
        return q, t, [summary_to_dict(x) for x in s]

    def bench_eval(self, q, t, warmup, steps, kernel_pass=True):
        """(ms over `steps` fused evaluations, mean ms of the per-point kernel alone | None)"""
        q, t = _f64(q).reshape(-1, 4), _f64(t).reshape(-1, 3)
        ms_total, ms_kernel = C.c_double(), C.c_double()
        _check(load().ea_batch_bench_eval(self._h, _dp(q), _dp(t), warmup, steps, C.byref(ms_total),
                                          C.byref(ms_kernel) if kernel_pass else None))
        return ms_total.value, (ms_kernel.value if kernel_pass else None)

    def bench_capture(self, steps):
        """untimed: capture `steps` steps into a hipGraph that bench_steps(steps) replays"""
        _check(load().ea_batch_bench_capture(self._h, int(steps)))

    def bench_capture_pipelined(self, steps):
        """untimed: the same steps with the fold of step k-1 riding in the launch of evaluation k (K launches + 1)"""
        _check(load().ea_batch_bench_capture_pipelined(self._h, int(steps)))

    def bench_result(self, riding=False):
        """what the last step of the last bench_steps left in the batch's result array (layout of eval);
        riding=True: the last-but-one step's result of a pipelined sequence (folded inside an evaluation launch)"""
        n = len(self)
        cost, JtJ, Jtr = np.zeros(n), np.zeros((n, 6, 6)), np.zeros((n, 6))
        bad = np.zeros(n, dtype=np.int64)
        fn = load().ea_batch_bench_result_riding if riding else load().ea_batch_bench_result
        _check(fn(self._h, _dp(cost), _dp(JtJ), _dp(Jtr), bad.ctypes.data_as(C.POINTER(C.c_int64))))
        return dict(cost=cost, JtJ=JtJ, Jtr=Jtr, n_invalid=bad)

    def bench_steps(self, steps, host_times=False, riding=False):
        """`steps` x (fused evaluation + fold) at the poses already on the device, then a stream sync: the timed region.
        riding=True: launch by launch with the fold of step k-1 riding in evaluation k (no graph)"""
        fn = load().ea_batch_bench_steps_riding if riding else load().ea_batch_bench_steps
        if host_times:
            us = np.zeros(3)
            _check(fn(self._h, int(steps), _dp(us)))
            return us
        _check(fn(self._h, int(steps), None))

    def bench_kernel(self, q, t, warmup, launches):
        """mean ms of the per-point kernel over `launches` back-to-back launches (one event pair)"""
        q, t = _f64(q).reshape(-1, 4), _f64(t).reshape(-1, 3)
        ms = C.c_double()
        _check(load().ea_batch_bench_kernel(self._h, _dp(q), _dp(t), warmup, launches, C.byref(ms)))
        return ms.value

    def row_offsets(self):
        """rows of problem i in the materialised outputs: [offsets[i], offsets[i + 1])"""
        off = np.zeros(len(self) + 1, dtype=np.int64)
        _check(load().ea_batch_row_offsets(self._h, off.ctypes.data_as(C.POINTER(C.c_int64))))
        return off

    def eval_rows(self, q, t, corrected=True, layout=0):
        """materialised mode into host arrays of the batch's dtype: r [rows], J [rows, 6] (layout 0) or [6, rows] (layout 1)"""
        q, t = _f64(q).reshape(-1, 4), _f64(t).reshape(-1, 3)
        n = int(self.row_offsets()[-1])
        dt = np.float32 if self.dtype == EA_F32 else np.float64
        r = np.zeros(n, dtype=dt)
        J = np.zeros((n, 6) if layout == 0 else (6, n), dtype=dt)
        bad = C.c_int64()
        _check(load().ea_batch_eval_rows(self._h, _dp(q), _dp(t), int(corrected), int(layout), r.ctypes.data_as(C.c_void_p),
                                         J.ctypes.data_as(C.c_void_p), n, C.byref(bad)))
        return r, J, bad.value

    def eval_rows_device(self, q, t, r_ptr, J_ptr, capacity_rows=None, corrected=True, layout=0):
        """materialised mode into the caller's device arrays; returns n_invalid.  r_ptr / J_ptr: torch tensors on the batch's
        GPU (checked here: device, dtype, contiguity, size) or raw device addresses (then capacity_rows is required and
        the addresses are taken on trust)"""
        q, t = _f64(q).reshape(-1, 4), _f64(t).reshape(-1, 3)
        if hasattr(r_ptr, "data_ptr") or hasattr(J_ptr, "data_ptr"):
            import torch
            want = torch.float32 if self.dtype == EA_F32 else torch.float64
            rows = None
            for name, x, per_row in (("r", r_ptr, 1), ("J", J_ptr, 6)):
                if not hasattr(x, "data_ptr"):
                    raise TypeError("%s: pass torch tensors for both outputs or raw addresses for both" % name)
                if not x.is_cuda or x.dtype != want or not x.is_contiguous():
                    raise ValueError("%s must be a contiguous %s tensor on the batch's GPU" % (name, want))
                if x.device.index is not None and x.device.index != self.problems[0].device:
                    raise ValueError("%s lives on cuda:%d, the batch on device %d" % (name, x.device.index, self.problems[0].device))
                n = x.numel() // per_row
                rows = n if rows is None else min(rows, n)
            capacity_rows = rows if capacity_rows is None else min(int(capacity_rows), rows)
            torch.cuda.current_stream(r_ptr.device).synchronize()   # the tensors' producers are done before the library writes
            r_ptr, J_ptr = r_ptr.data_ptr(), J_ptr.data_ptr()
        if capacity_rows is None:
            raise ValueError("capacity_rows is required with raw addresses")
        bad = C.c_int64()
        _check(load().ea_batch_eval_rows_device(self._h, _dp(q), _dp(t), int(corrected), int(layout), C.c_void_p(r_ptr),
                                                C.c_void_p(J_ptr), int(capacity_rows), C.byref(bad)))
        return bad.value

    def bench_rows(self, q, t, warmup, launches, corrected=True, layout=0, mode=1, r_ptr=None, J_ptr=None, capacity_rows=0):
        """mean ms of the materialised-mode kernel over `launches` back-to-back launches (one event pair)"""
        q, t = _f64(q).reshape(-1, 4), _f64(t).reshape(-1, 3)
        ms = C.c_double()
        _check(load().ea_batch_bench_rows(self._h, _dp(q), _dp(t), int(corrected), int(layout), int(mode),
                                          C.c_void_p(r_ptr) if r_ptr else None, C.c_void_p(J_ptr) if J_ptr else None,
                                          int(capacity_rows), warmup, launches, C.byref(ms)))
        return ms.value

    def bench_fold(self, warmup, launches):
        """mean ms of the fold kernel over `launches` back-to-back launches (one event pair)"""
        ms = C.c_double()
        _check(load().ea_batch_bench_fold(self._h, warmup, launches, C.byref(ms)))
        return ms.value

    def set_tuning(self, key, value):
        _check(load().ea_batch_set_tuning(self._h, key.encode(), int(value)))

    def info(self, key):
        v = C.c_int64()
        _check(load().ea_batch_get_info(self._h, key.encode(), C.byref(v)))
        return v.value


def resize_half(img, nan_to_zero=False, device=0):
    """cv::resize(img, dst, Size(), 0.5, 0.5) on the device (ea_resize_half): uint8 H x W x 3 or float32 H x W"""
    img = np.ascontiguousarray(img)
    if img.dtype == np.uint8:
        assert img.ndim == 3 and img.shape[2] == 3
        kind, out = 0, np.zeros((img.shape[0] // 2, img.shape[1] // 2, 3), np.uint8)
    else:
        img = np.ascontiguousarray(img, dtype=np.float32)
        assert img.ndim == 2
        kind, out = (1 if nan_to_zero else 2), np.zeros((img.shape[0] // 2, img.shape[1] // 2), np.float32)
    _check(load().ea_resize_half(device, kind, img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1],
                                 out.ctypes.data_as(C.c_void_p)))
    return out


def selftest_wave_reduce(values, device=0):
    """values: (32, 64) float32 -> (totals from the fp32 reduction, totals from the fp64 reduction, stages (30,64))"""
    v = np.ascontiguousarray(values, dtype=np.float32)
    assert v.shape == (32, 64)
    o32, o64 = np.zeros(32), np.zeros(32)
    st = np.zeros((30, 64), dtype=np.float32)
    _check(load().ea_selftest_wave_reduce(device, v.ctypes.data_as(C.POINTER(C.c_float)), _dp(o32), _dp(o64),
                                          st.ctypes.data_as(C.POINTER(C.c_float))))
    return o32, o64, st
