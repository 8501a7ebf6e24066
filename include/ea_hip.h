/*
 * ea_hip.h — C-ABI of the MI355X-native edge-alignment hot path (libea_hip.so).
 *
 * This is the drop-in boundary for ONE path of kuwt/edge_alignment: the per-edge-point
 * EAResidue evaluation (SE(3) warp -> pinhole -> bicubic DT sample -> 1x6 Jacobian), the
 * JtJ / Jtr / cost reduction, and the trust-region loop the reference hands to ceres::Solve.
 * The reference has no FFI layer (it is a C++ source-level API on top of Ceres); each entry
 * point below names the reference code it replaces (paths relative to the reference root).
 * Host C++ shims that keep the reference's own spellings (EAResidue, SolveEA, a minimal
 * ceres:: facade) sit on top of this header in edge_alignment_amd/include/.
 *
 * Conventions
 *   - plain C types only; every function returns an ea_status (0 = ok, < 0 = error) and
 *     never throws.  ea_last_error() gives a thread-local message for the last failure.
 *   - poses are q = (w,x,y,z), t = (tx,ty,tz), b_T_a (maps frame-A points into frame B),
 *     exactly the raw arrays of PoseManipUtils::eigenmat_to_raw
 *     (standalone/PoseManipUtils.cpp:16-27); ea_solve updates them in place like ceres::Solve.
 *   - host buffers handed to ea_problem_set_* are copied to HBM before the call returns and
 *     may be freed afterwards (the reference's Grid2D merely borrows, utils.h:95).
 *   - a handle is not thread-safe; use one handle per host thread / per GPU.
 *   - there is NO CPU fallback: every compute entry point fails with EA_ERR_NO_DEVICE when
 *     no gfx950 device is usable.
 */
#ifndef EA_HIP_H
#define EA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  EA_OK = 0,
  EA_ERR_INVALID_ARG = -1,
  EA_ERR_HIP = -2,
  EA_ERR_NO_DEVICE = -3,
  EA_ERR_STATE = -4,
  EA_ERR_ALLOC = -5
} ea_status;

/* arithmetic type of the per-point evaluation (reductions are always fp64) */
typedef enum { EA_F64 = 0, EA_F32 = 1 } ea_dtype;

/* ceres::TrivialLoss / CauchyLoss(a) / HuberLoss(a)
 * (standalone_edge_align.cpp:272, :2604; src/SolveEA.cpp:144) */
typedef enum { EA_LOSS_TRIVIAL = 0, EA_LOSS_CAUCHY = 1, EA_LOSS_HUBER = 2 } ea_loss_kind;

/* ceres::TerminationType as far as this path can produce it */
typedef enum { EA_CONVERGENCE = 0, EA_NO_CONVERGENCE = 1, EA_FAILURE = 2 } ea_termination;

typedef enum {
  EA_WHY_NONE = 0,
  EA_WHY_FUNCTION_TOL = 1,
  EA_WHY_GRADIENT_TOL = 2,
  EA_WHY_PARAMETER_TOL = 3,
  EA_WHY_MAX_ITERATIONS = 4,
  EA_WHY_MIN_RADIUS = 5,
  EA_WHY_INITIAL_EVAL_FAILED = 6,     /* a functor returned false at the start (utils.h:70-73) */
  EA_WHY_TOO_MANY_INVALID_STEPS = 7,
  EA_WHY_EVAL_FAILED = 8
} ea_why;

typedef enum { EA_STRATEGY_LM = 0, EA_STRATEGY_DOGLEG = 1 } ea_strategy;

/* pinhole intrinsics, the four doubles every EAResidue carries (utils.h:41-44,96) */
typedef struct {
  double fx, fy, cx, cy;
} ea_camera;

/* the fields of ceres::Solver::Options the reference touches or relies on by default
 * (standalone_edge_align.cpp:282-284, :2112-2116; src/SolveEA.cpp:184-192) */
typedef struct {
  int max_num_iterations;             /* 50 */
  double function_tolerance;          /* 1e-6 */
  double gradient_tolerance;          /* 1e-10 */
  double parameter_tolerance;         /* 1e-8 */
  double initial_trust_region_radius; /* 1e4 */
  double max_trust_region_radius;     /* 1e16 */
  double min_trust_region_radius;     /* 1e-32 */
  double min_relative_decrease;       /* 1e-3 */
  double min_lm_diagonal;             /* 1e-6 */
  double max_lm_diagonal;             /* 1e32 */
  int max_num_consecutive_invalid_steps; /* 5 */
  int jacobi_scaling;                 /* 1 */
  int strategy;                       /* EA_STRATEGY_LM (LEVENBERG_MARQUARDT) | EA_STRATEGY_DOGLEG */
  int minimizer_progress_to_stdout;   /* 0 */
  int iterations_per_sync;            /* device LM iterations enqueued between host checks; 0 = default */
  double solve_timeout_ms;            /* a solve whose device side shows no progress for this long returns EA_ERR_HIP
                                         instead of spinning for ever; 0 = default (5000 ms), < 0 = no deadline */
} ea_options;

#define EA_MAX_TRACE 128

/* what the drivers read back from ceres::Solver::Summary (FullReport(), :293) */
typedef struct {
  int termination;  /* ea_termination */
  int why;          /* ea_why */
  int num_iterations;          /* successful + unsuccessful steps (+1 if the last one ended the solve) */
  int num_successful_steps;
  int num_unsuccessful_steps;
  double initial_cost, final_cost;
  int64_t num_point_evals;     /* edge-point residual+Jacobian evaluations performed */
  double total_time_ms;        /* host wall time of the call */
  /* per-iteration trace, entries [0 .. min(num_iterations, EA_MAX_TRACE-1)] */
  double it_cost[EA_MAX_TRACE];
  double it_cost_change[EA_MAX_TRACE];
  double it_gradient_max_norm[EA_MAX_TRACE];
  double it_step_norm[EA_MAX_TRACE];
  double it_relative_decrease[EA_MAX_TRACE];
  double it_radius[EA_MAX_TRACE];
  int it_successful[EA_MAX_TRACE];
} ea_summary;

typedef struct ea_problem ea_problem; /* one frame pair: edge points of A + DT image of B */
typedef struct ea_batch ea_batch;     /* several problems evaluated / solved by the same launches */

/* ---- library ------------------------------------------------------------------------- */
const char *ea_last_error(void);
/* Device blocks, pinned host blocks and streams freed by ea_problem_destroy / ea_batch_destroy (and by re-sizing calls) are
 * kept by the library and handed out again (bounded: 1 GiB of device memory, 256 MiB pinned): creating and destroying an
 * ea_problem per solve -- what the ceres facade does per ceres::Solve -- then costs no driver allocation.  This returns
 * everything that is cached to the driver. */
int ea_release_cached_memory(void);
/* Page-locked host memory for the caller's frames (the buffers handed to ea_problem_set_*_frame*, ea_tracker_push_frame,
 * ea_problem_set_points / _set_dt): the uploads of those calls then run as direct DMA instead of being staged through the
 * runtime's bounce buffers -- ea_tracker_push_frame 0.31 -> 0.28 ms per VGA frame (profiles/r03_tracker_pinned.txt).
 * Plain memory to the host (cv::Mat can wrap it); NULL on failure (ea_last_error).  Free with ea_host_free only. */
void *ea_host_alloc(size_t bytes, int device);
void ea_host_free(void *p);
const char *ea_version(void);
int ea_device_count(int *count);            /* number of gfx950 devices visible */
void ea_default_options(ea_options *opt);   /* ceres::Solver::Options() defaults used by test1 */

/* ---- problem = ceres::Problem filled by the loop at standalone_edge_align.cpp:264-278 --- */
int ea_problem_create(ea_problem **out, const ea_camera *cam, int dtype, int device);
void ea_problem_destroy(ea_problem *p);

/* Edge points, replacing the N `EAResidue::Create(fx,fy,cx,cy, X,Y,Z, interp)` +
 * `AddResidualBlock` calls (standalone_edge_align.cpp:267-274; src/SolveEA.cpp:163-175).
 * xyz: host, point i at xyz[i*stride_elems + {0,1,2}] — stride 4 for the 4xN column-major
 * a_X of get_aX (utils.cpp:268-280) or 3 for list_edge_ref (SolveEA.cpp:55).  Converted once
 * to SoA x[],y[],z[] of the problem dtype in HBM. */
int ea_problem_set_points(ea_problem *p, const double *xyz, int64_t n, int64_t stride_elems);
/* Storage order of the points ea_problem_set_points uploads after this call.  tile_px > 0: tiles of
 * tile_px x tile_px pixels of the reference frame (identity-pose projection), tiles in raster order,
 * the caller's order inside a tile -- a wavefront's points then sample a compact patch of the DT image;
 * 0: the caller's order (the reference's: residual blocks in the order of the a_X columns);
 * < 0 (default): automatic, tiles of 16 px from 200 000 points up.  Only the order of summation
 * changes (results agree to rounding); per-point outputs (ea_eval_points) and ea_problem_get_points
 * stay in the caller's order. */
int ea_problem_set_point_order(ea_problem *p, int tile_px);
/* the tile size the stored points are ordered by (0 = the caller's order) */
int ea_problem_get_point_order(const ea_problem *p, int *tile_px);
/* same, from SoA arrays of the problem's dtype already resident in HBM (borrowed, must
 * outlive the problem) */
int ea_problem_set_points_device(ea_problem *p, const void *x, const void *y, const void *z,
                                 int64_t n);

/* Distance-transform image, replacing
 *   ceres::Grid2D<double,1> grid(data, 0, grid_rows, 0, grid_cols);
 *   ceres::BiCubicInterpolator<...> interp(grid);        (standalone_edge_align.cpp:258-259)
 * data: host, row-major, value(r,c) = data[r*grid_cols + c]; the functor samples it at
 * (r = u, c = v) (utils.h:77), so grid_rows is the u extent (image width) and grid_cols the
 * v extent (image height) exactly as the reference passes e_disTrans.cols()/rows(). */
int ea_problem_set_dt(ea_problem *p, const double *data, int grid_rows, int grid_cols);
/* same, from an image already in HBM: row-major [height][width] (u contiguous) of the problem's
 * dtype; copied into the library's padded layout by a device kernel */
int ea_problem_set_dt_image_device(ea_problem *p, const void *image, int height, int width);

/* loss_function argument of AddResidualBlock (:272 `new CauchyLoss(1.)`) */
int ea_problem_set_loss(ea_problem *p, int loss_kind, double a);
/* functor flavour: standalone (utils.h:48-80) = {z_guard 0.01, z_eps 0, rot_transposed 0}
 * [default]; ROS flavour (include/EAResidue.h:86-118) = {0, 0.001, 1} */
int ea_problem_set_flavour(ea_problem *p, double z_guard, double z_eps, int rot_transposed);
int64_t ea_problem_num_points(const ea_problem *p);

/* Residual variants of standalone/utils.h.
 * Distortion: EAResidueEx / EAResidueSecondCamEx (utils.h:102-177, :295-421), coefficients in the order
 * of EAResidueEx::Create(fx,fy,cx,cy, k1,k2,p1,p2,k3, ...) (utils.h:156-160).  All zero = off. */
int ea_problem_set_distortion(ea_problem *p, double k1, double k2, double p1, double p2, double k3);
/* Second camera of a rigid rig: EAResidueSecondCam[Ex] (utils.h:179-292) evaluates
 * b_T_a_SecCam = trans_1to2 * b_T_a * trans_1to2_inv; both row-major 4x4 with last row 0 0 0 1, exactly the
 * ptrans_1to2 / ptrans_1to2_inv arrays.  NULL, NULL = first camera again. */
int ea_problem_set_second_camera(ea_problem *p, const double trans_1to2[16], const double trans_1to2_inv[16]);
/* Several residual families on ONE pose — the reference adds camera-1 and camera-2 blocks to the same
 * ceres::Problem with the same (q,t) (standalone_edge_align.cpp:791-803, :3205-3218).  `term` keeps its own
 * points, DT image, intrinsics, loss and variant; it must outlive `p` (borrowed).  ea_eval / ea_solve on `p`
 * then cover p and all its terms. */
int ea_problem_add_term(ea_problem *p, ea_problem *term);
int ea_problem_clear_terms(ea_problem *p);

/* One evaluation of the whole problem at pose (q,t) — what ceres' evaluator computes from the
 * N AutoDiffCostFunction<EAResidue,1,4,3> blocks + QuaternionParameterization + loss:
 *   cost = 1/2 sum rho(r_i^2);  JtJ (6x6 row-major) and Jtr (6) of the loss-corrected 1x6 rows
 *   in the tangent ordering [delta(3) | t(3)];  n_invalid = blocks whose functor returned false
 *   (sums then cover the valid blocks only).  Any output pointer may be NULL. */
int ea_eval(ea_problem *p, const double q[4], const double t[3], double *cost, double JtJ[36],
            double Jtr[6], int64_t *n_invalid);
/* per-point outputs (host, n and n*6 row-major; NaN for failed blocks).  corrected != 0:
 * sqrt(rho') r and sqrt(rho') J as handed to the minimiser; 0: raw r_i and raw row. */
int ea_eval_points(ea_problem *p, const double q[4], const double t[3], double *r, double *J,
                   int corrected);
/* residual-only evaluation (the candidate-cost evaluation inside the trust-region loop) */
int ea_cost(ea_problem *p, const double q[4], const double t[3], double *cost,
            int64_t *n_invalid);

/* The reference's own quantitative self-check, printed before and after its solves (standalone_edge_align.cpp:2494-2567
 * "initialCost / InitialMaxCost / Initialmaxpixel", :2704-2776 the same after): every point warped by the pose, x / z and
 * y / z pushed through K, truncated `(int)` to a pixel, the distance transform read there; total, mean over the points,
 * maximum and the (untruncated) pixel of the first point that attains it.  Computed on the device from the points and the
 * DT image the problem already holds.  Upstream reads outside the image unchecked (undefined behaviour): such points are
 * skipped and counted in `outside`; upstream accumulates in float in point order, this sums in double. */
typedef struct {
  double total_cost, mean_cost, max_cost;
  double max_pixel[2];      /* (u, v) before truncation */
  int64_t count, outside;   /* points read / points projecting outside the image */
} ea_pixel_cost;
int ea_problem_pixel_cost(ea_problem *p, const double q[4], const double t[3], ea_pixel_cost *out);

/* ceres::Solve(options, &problem, &summary) (standalone_edge_align.cpp:286; SolveEA.cpp:198).
 * q,t in/out.  The whole trust-region loop runs on the device. */
int ea_solve(ea_problem *p, const ea_options *opt, double q[4], double t[3], ea_summary *summary);

/* One problem whose points are sharded over several processes / GPUs: every rank holds a shard of
 * the points (ea_problem_set_points with its slice) and the whole DT image.  Per iteration: local fused evaluation ->
 * 32 accumulator slots -> `allreduce` (in-place sum over all ranks; 0 = success) -> the same trust-region step on every
 * rank (deterministic: no broadcast).  Every rank must call it with the same options and initial pose.  summary->
 * num_point_evals counts the local shard. */
typedef int (*ea_allreduce_fn)(double *buf, int count, void *user);
int ea_solve_sharded(ea_problem *p, const ea_options *opt, ea_allreduce_fn allreduce, void *user, double q[4], double t[3],
                     ea_summary *summary);
/* The same solve with the exchange kept ON THE STREAM ("no host round-trip"): per iteration
 * the library enqueues, on one HIP stream, fused evaluation -> fold into `device_sums` (32 doubles of 16-byte aligned DEVICE memory owned
 * by the caller, e.g. the storage of a torch tensor) -> `allreduce(device_sums, 32, stream, user)`, which must ENQUEUE an
 * in-place sum over all ranks on that stream (RCCL: ncclAllReduce on it, or torch.distributed under an ExternalStream)
 * and return without waiting -> the trust-region step kernel, which reads `device_sums`.  Iterations are enqueued in
 * rounds of opt->iterations_per_sync (default 4); the host looks at the device's progress word once per round, and since
 * every rank computes the same state from the same sums, every rank stops after the same round: the number of
 * collectives enqueued is identical on all ranks.  `hip_stream` is the hipStream_t as a void pointer. */
typedef int (*ea_device_allreduce_fn)(void *device_buf, int count, void *hip_stream, void *user);
int ea_solve_sharded_device(ea_problem *p, const ea_options *opt, ea_device_allreduce_fn allreduce, void *user,
                            double *device_sums, double q[4], double t[3], ea_summary *summary);

/* ---- several GPUs: one communicator rank per GPU, RCCL over xGMI, called from librccl directly -------------------------
 * The path shards by INDEPENDENT frame pairs -- the unit is one ceres::Solve per pair (standalone_edge_align.cpp:286,
 * src/SolveEA.cpp:198): every GPU solves its own ea_batch, no data-path collective -- and the one exchange step is the
 * all-gather of the solved poses.  librccl is opened on first use (dlopen): callers that stay on one GPU never load it.
 * One rank per GPU, either as one process per GPU (ea_comm_get_unique_id on rank 0, the 128 bytes handed to the others
 * by whatever launched them, ea_comm_create on every rank) or as host threads of one process (ea_comm_create_all, one
 * communicator per device, each used by the thread that drives that device). */
typedef struct ea_comm ea_comm;
#define EA_COMM_ID_BYTES 128
int ea_comm_get_unique_id(unsigned char id[EA_COMM_ID_BYTES]);
int ea_comm_create(ea_comm **out, const unsigned char id[EA_COMM_ID_BYTES], int nranks, int rank, int device);
int ea_comm_create_all(ea_comm **out /* ndev entries */, const int *devices /* NULL = 0 .. ndev-1 */, int ndev);
void ea_comm_destroy(ea_comm *c);
int ea_comm_rank(const ea_comm *c);
int ea_comm_size(const ea_comm *c);
/* THE collective of the batch mode: every rank hands in the `count` poses it solved (q: count x 4, t: count x 3, status:
 * count ints such as ea_summary.termination, NULL = zeros) and receives all nranks x count of them in rank order (any of
 * the three outputs may be NULL).  ONE ncclAllGather of count x 8 doubles, enqueued on the stream of `after` -- the batch
 * whose ea_batch_solve produced the poses; NULL = the communicator's own stream -- one synchronisation.  `count` must be
 * the same on every rank (BASELINE config C4: 32). */
int ea_comm_gather_poses(ea_comm *c, ea_batch *after, const double *q, const double *t, const int *status, int count,
                         double *all_q, double *all_t, int *all_status);
/* The point-sharded solve with the exchange issued by the library itself, nothing leaving the device.  Every rank calls it
 * with its shard of the points (and the whole DT image), the same options and the same start pose.  When every rank's shard
 * is small enough for one workgroup per CU (settled by one small MAX all-reduce up front) an iteration is ONE kernel launch
 * + ONE in-place ncclAllReduce of the shard's partial rows (a few tens of KB): the next launch folds the summed rows, takes
 * the trust-region step in every workgroup and evaluates at the new pose.  Otherwise: evaluation -> fold ->
 * ncclAllReduce(32 doubles, sum) -> step kernel per iteration (ea_solve_sharded_device's sequence). */
int ea_solve_sharded_comm(ea_problem *p, const ea_options *opt, ea_comm *c, double q[4], double t[3], ea_summary *summary);
/* key in {"allreduces", "allgathers", "row_solves", "device"}: collectives enqueued so far (tests: equal on every rank);
 * sharded solves that took the one-launch-per-iteration form */
int ea_comm_get_info(const ea_comm *c, const char *key, int64_t *value);
/* number of HIP runtimes (libamdhip64) mapped in this process.  1 is the only healthy answer: PyTorch-ROCm ships its own
 * copy under the same SONAME, so a process that imports torch FIRST shares one runtime with this library, while loading
 * this library first maps two -- streams and device pointers of one are then not objects of the other (ea_comm_* and the
 * on-stream collectives refuse to run). */
int ea_hip_runtime_copies(void);

/* Coarse-to-fine driver (BASELINE config C3; the reference has no pyramid): levels[0] = finest.
 * Solves levels[nlevels-1] first and carries q, t down level by level; every level is a complete problem with its own
 * points, DT image and (caller-scaled) intrinsics.  summaries: nlevels entries or NULL. */
int ea_solve_pyramid(ea_problem *const *levels, int nlevels, const ea_options *opt, double q[4], double t[3],
                     ea_summary *summaries);

/* Frame-to-frame driver (upstream aligns one stored pair, src/ea.cpp:155-200): every pushed frame
 * is aligned against the previous one, starting from the last relative pose; pre-processing (flavour 0: get_aX /
 * get_distance_transform, 1: the Canny forms) and solve stay on the device.  q_rel, t_rel: pose of the previous frame
 * in the new frame's coordinates (identity for the first frame); *aligned (nullable) = 1 when a solve took place. */
typedef struct ea_tracker ea_tracker;
int ea_tracker_create(ea_tracker **out, const ea_camera *cam, int dtype, int device, int flavour);
void ea_tracker_destroy(ea_tracker *tr);
ea_problem *ea_tracker_problem(ea_tracker *tr); /* the problem it drives (set loss / flavour knobs on it) */
int ea_tracker_push_frame(ea_tracker *tr, const uint8_t *bgr, const uint16_t *depth, int height, int width,
                          double z_scaling, const ea_options *opt, double q_rel[4], double t_rel[3],
                          ea_summary *summary, int *aligned);

/* ---- batches of independent frame pairs (one launch sequence for all of them) --------- */
int ea_batch_create(ea_batch **out, ea_problem *const *problems, int count);
void ea_batch_destroy(ea_batch *b);
int ea_batch_count(const ea_batch *b);
/* q: count x 4, t: count x 3, cost: count, JtJ: count x 36, Jtr: count x 6, n_invalid: count */
int ea_batch_eval(ea_batch *b, const double *q, const double *t, double *cost, double *JtJ,
                  double *Jtr, int64_t *n_invalid);
int ea_batch_solve(ea_batch *b, const ea_options *opt, double *q, double *t,
                   ea_summary *summaries);

/* K evaluations of every problem of the batch at K DIFFERENT poses, one call -- what a caller that runs its own optimiser,
 * a line search, multi-start or a cost-surface probe asks of the evaluator: ceres::Problem::Evaluate once per pose (the
 * reference's own call of it: src/SolveEA.cpp:241).  q: K x count x 4, t: K x count x 3 (pose k of problem i at
 * [k * count + i]); outputs in the layout of ea_batch_eval with the same leading index: cost K x count, JtJ K x count x 36,
 * Jtr K x count x 6, n_invalid K x count; any of them may be NULL.  Independent evaluations do not queue up behind each
 * other: the pose is a batch dimension of the launch -- G poses per evaluation launch (grid = chunks x G x terms, every
 * (point, pose) pair through the whole per-point arithmetic) and one fold launch for their partial rows, ceil(K / G) such
 * pairs, the results folded straight into pinned host memory, ONE synchronisation: well under 1 us per evaluation of a
 * 5e4-point pair from a few dozen poses on, against ~27 us through ea_batch_eval.  The sums of pose k are those
 * ea_batch_eval returns at pose k up to rounding (the pose-batched launch takes a throughput shape -- more points per
 * lane -- so the partial rows are cut differently, and the pose-dependent constants are built on the device); any split of
 * the K poses over launches gives the same bits.  Every kind of batch is covered (variant functors, shared-pose terms,
 * LDS staging). */
int ea_batch_eval_poses(ea_batch *b, int K, const double *q, const double *t, double *cost, double *JtJ, double *Jtr,
                        int64_t *n_invalid);
/* The same in two halves, for a caller that evaluates the same poses again (or wants the upload off its critical path):
 * ea_batch_set_poses makes K poses per problem resident in HBM (56 bytes per pose go up; the pose-dependent constants
 * the kernels read are built by a device kernel); ea_batch_eval_resident_poses runs their K evaluations.  A change of
 * the batch's problems (points, DT image, loss, tuning) drops the resident poses: EA_ERR_STATE until they are set again. */
int ea_batch_set_poses(ea_batch *b, int K, const double *q, const double *t);
int ea_batch_eval_resident_poses(ea_batch *b, double *cost, double *JtJ, double *Jtr, int64_t *n_invalid);

/* ---- materialised mode: the "EAResidue batch Evaluate" view -------------------------------------------------------
 * Replaces N calls of ceres::AutoDiffCostFunction<EAResidue,1,4,3>::Evaluate followed by the parameterisation's 4x3
 * plus-Jacobian (standalone/utils.h:48-92, standalone_edge_align.cpp:261-278): residual r and the effective 1x6 row
 * [d r / d delta | d r / d t] of EVERY point, written in the batch's dtype (float / double) -- for a caller that runs its
 * own solver on the rows.  Rows of problem i are [offsets[i], offsets[i+1]) (terms of a problem adjacent, points in
 * their storage order: ea_problem_get_points).  layout 0: J row-major [rows][6]; 1: column-major [6][rows].
 * corrected != 0: rows scaled by sqrt(rho'(r^2)) (Ceres' Corrector, alpha = 0).  A functor that returns false
 * (utils.h:70-73) leaves a NaN row and is counted in *n_invalid (Ceres fails the whole evaluation then).
 * This mode is bandwidth-bound: 3 s bytes in, 7 s bytes out per point plus one pass over the DT image. */
int ea_batch_row_offsets(ea_batch *b, int64_t *offsets /* count + 1 */);
/* into DEVICE memory of the batch's GPU owned by the caller (e.g. the storage of a torch tensor), 16-byte aligned,
 * capacity_rows >= offsets[count]; complete and visible when the call returns (the batch's stream is synchronised).
 * r_dev == J_dev == NULL: into the library's own arrays (then read them with ea_batch_eval_rows). */
int ea_batch_eval_rows_device(ea_batch *b, const double *q, const double *t, int corrected, int layout, void *r_dev,
                              void *J_dev, int64_t capacity_rows, int64_t *n_invalid);
/* the same into host arrays of the batch's dtype holding capacity_rows >= offsets[count] rows (either may be NULL) */
int ea_batch_eval_rows(ea_batch *b, const double *q, const double *t, int corrected, int layout, void *r_host,
                       void *J_host, int64_t capacity_rows, int64_t *n_invalid);

/* the same for one problem (its terms included, in term order; rows = ea_problem_num_rows) */
int ea_problem_num_rows(ea_problem *p, int64_t *rows);
int ea_eval_rows(ea_problem *p, const double q[4], const double t[3], int corrected, int layout, void *r_host, void *J_host,
                 int64_t capacity_rows, int64_t *n_invalid);
int ea_eval_rows_device(ea_problem *p, const double q[4], const double t[3], int corrected, int layout, void *r_dev, void *J_dev,
                        int64_t capacity_rows, int64_t *n_invalid);

/* ---- tuning ---------------------------------------------------------------------------- */
/* tuning knobs: key in {"lds_bytes", "points_per_thread", "use_lds", "xcd_remap", "threads", "buffer_loads",
 * "solve_streams", "rows_staged", "rows_nontemporal", "wide_accumulate", "dt_f32", "poses_per_launch", "poll_results",
 * "fused_iterations", "zero_copy_poses"};
 * value < 0 restores the default.
 * "dt_f32" = 0: an fp64 batch reads its fp64 images even where a float32 mirror holds them exactly (default: the mirror
 * when every term has one; results are bit-identical either way).  "poses_per_launch" = g > 0 caps the poses one
 * evaluation launch of ea_batch_eval_poses covers (default: enough to fill the chip, ~32k workgroups).  "poll_results" = 0:
 * ea_batch_eval / ea_batch_eval_poses wait for the stream's completion signal instead of returning on the flag their last
 * fold workgroup raises in pinned memory ~6 us earlier (default 1; a caller that synchronises the whole device right behind
 * the call is better off with 0: profiles/r03_ab_poll.txt).
 * "zero_copy_poses" = 0: ea_batch_eval / ea_eval always upload the pose constants before the launch (default: for up to four
 * problems the kernel reads them from the pinned host block they were written to -- one transfer less in front of a lone
 * evaluation, 26 -> 24 us per call).
 * "fused_iterations" = 0: ea_solve / ea_batch_solve always run (evaluate, step) pairs; default: a solve of problems small
 * enough for one workgroup per CU (LM strategy, one plain residual family each) runs ONE launch per iteration, every workgroup
 * taking the LM step itself before it evaluates -- the same iterates bit for bit (ea_batch_get_info "fused_iterations"
 * reports which form the last solve took).
 * "wide_accumulate" = 1: an fp32 batch sums in fp64 from a lane's sum of <= points_per_thread products on (default: a
 * lane's and a wavefront's sums are fp32, everything above fp64).  Plain functor on the L2 path; ignored for fp64
 * batches, variant functors and the LDS-staged form (ea_batch_get_info "wide_accumulate" reports what is in effect).
 * Applies to ea_batch_eval and the solves; cost: profiles/r02_ab_wide_accumulate.txt. */
int ea_batch_set_tuning(ea_batch *b, const char *key, int value);
int ea_batch_get_info(const ea_batch *b, const char *key, int64_t *value);

/* ---- producers either side of the hot path, on the device ----------------
 * Reference frame: replaces get_aX (standalone/utils.cpp:201-281) + the residual-block loop at stride 1:
 * GaussianBlur 3x3 -> CV_RGB2GRAY -> Laplacian k3 -> |.| > threshold && depth > 0 -> back-projection with the
 * problem's intrinsics, points kept in raster order.  bgr: H x W x 3 bytes as cv::imread returns them,
 * depth: H x W uint16 (TUM: z_scaling 5000).  Host pointers. */
int ea_problem_set_ref_frame(ea_problem *p, const uint8_t *bgr, const uint16_t *depth, int height, int width,
                             double z_scaling, int threshold);
/* get_aX_mask (utils.cpp:283-369; call sites standalone_edge_align.cpp:1039, :1081): the same, and mask > 0
 * (mask: H x W bytes) */
int ea_problem_set_ref_frame_masked(ea_problem *p, const uint8_t *bgr, const uint8_t *mask, const uint16_t *depth,
                                    int height, int width, double z_scaling, int threshold);
/* Current frame: replaces get_distance_transform (utils.cpp:38-83) + cv2eigen + Grid2D (:201-206, :258):
 * edge map (same gradient, > threshold) -> [3x3 median] -> 3x3 chamfer DT (DIST_L2, mask 3) -> [min-max
 * normalise to [0,1]] written straight into the problem's DT image in HBM. */
int ea_problem_set_now_frame(ea_problem *p, const uint8_t *bgr, int height, int width, int threshold, int median,
                             int normalize);
/* same, and copies the intermediate stages back (any may be NULL): |Laplacian| (H*W bytes), edge mask after
 * the median (0 = edge), chamfer distance in 16.16 fixed point, final float32 DT */
int ea_problem_debug_now_frame(ea_problem *p, const uint8_t *bgr, int height, int width, int threshold, int median,
                               int normalize, uint8_t *lap_out, uint8_t *mask_out, int32_t *chamfer_fix_out,
                               float *dt_out);
/* Canny flavour of the two producers -- what every standalone test after edge_align_test1 uses:
 * blur 3x3 -> CV_RGB2GRAY -> Canny(low, high) (aperture 3, L1 gradient; the reference passes 30, 90).
 * Reference frame: get_aX_canny (utils.cpp:371-462): edge pixel && depth > 0 -> back-projection, raster order. */
int ea_problem_set_ref_frame_canny(ea_problem *p, const uint8_t *bgr, const uint16_t *depth, int height, int width,
                                   double z_scaling, double low_threshold, double high_threshold);
/* Current frame: get_distance_transform2 (utils.cpp:85-106), _masked (:108-141; mask: H x W bytes, edges survive where
 * mask > 1), _NoNormalize (:142-165), _masked_NoNormalize (:166-199): edges -> 3x3 chamfer DT -> [min-max normalise to
 * [norm_lo, norm_hi]: (0,1) at :103, (0,255) at :138] written into the problem's DT image.  mask may be NULL. */
int ea_problem_set_now_frame_canny(ea_problem *p, const uint8_t *bgr, const uint8_t *mask, int height, int width,
                                   double low_threshold, double high_threshold, int normalize, double norm_lo,
                                   double norm_hi);
/* same, and copies the stages back (any may be NULL): edge map (0/255), chamfer distance in 16.16 fixed point,
 * final float32 DT, number of hysteresis launches */
int ea_problem_debug_now_frame_canny(ea_problem *p, const uint8_t *bgr, const uint8_t *mask, int height, int width,
                                     double low_threshold, double high_threshold, int normalize, double norm_lo,
                                     double norm_hi, uint8_t *edges_out, int32_t *chamfer_fix_out, float *dt_out,
                                     int *hysteresis_launches);
/* ROS flavour of the two producers (src/SolveEA.cpp:29-119): cv::Canny(rgb, 150, 100, 3, true) on the 3-channel image
 * (per pixel the channel with the largest dx^2 + dy^2; squared thresholds).
 * setRefFrame (:29-82): every edge pixel gives a point; depth: H x W float32 in metres, Z == 0 -> 1.0. */
int ea_problem_set_ref_frame_ros(ea_problem *p, const uint8_t *bgr, const float *depth, int height, int width,
                                 double threshold1, double threshold2);
/* setNowFrame (:86-119): 255 - edges -> distanceTransform(DIST_L2, DIST_MASK_PRECISE) (exact Euclidean) -> normalize to
 * [0, 255].  A frame without any edge returns EA_ERR_STATE.  The debug form copies the edge map / float32 DT back. */
int ea_problem_set_now_frame_ros(ea_problem *p, const uint8_t *bgr, int height, int width, double threshold1,
                                 double threshold2);
int ea_problem_debug_now_frame_ros(ea_problem *p, const uint8_t *bgr, int height, int width, double threshold1,
                                   double threshold2, uint8_t *edges_out, float *dt_out);
/* The same two producers fed with the frames as the ROS callbacks RECEIVE them (full resolution), the node's reduction to
 * its working resolution done on the device first: `halvings` times cv::resize(src, dst, cv::Size(), 0.5, 0.5) -- the
 * 2 x 2 area mean OpenCV's linear resize is at an exact factor of two -- on the bgr8 frame and on the float32 depth frame
 * after NaN -> 0 (src/ea.cpp:38, :56-62).  full_height / full_width must be divisible by 2^halvings; the problem's
 * intrinsics are those of the working resolution (SolveEA's constructor: TUM x 0.5, src/SolveEA.cpp:15-18).  With
 * halvings = 0..L these calls build the levels of ea_solve_pyramid from one pair of full-resolution frames. */
int ea_problem_set_ref_frame_ros_scaled(ea_problem *p, const uint8_t *bgr, const float *depth, int full_height,
                                        int full_width, int halvings, double threshold1, double threshold2);
int ea_problem_set_now_frame_ros_scaled(ea_problem *p, const uint8_t *bgr, int full_height, int full_width, int halvings,
                                        double threshold1, double threshold2);
/* the half-resolution step by itself, host in / host out: kind 0 = bgr8 (height x width x 3 bytes), 1 = float32 with
 * NaN -> 0 first, 2 = float32 as is; dst = (height / 2) x (width / 2); height and width even */
int ea_resize_half(int device, int kind, const void *src, int height, int width, void *dst);
/* read back what the problem holds in HBM: points (n x 3 doubles), DT image (H x W doubles, [v][u]) */
int ea_problem_get_points(ea_problem *p, double *xyz, int64_t capacity);
int ea_problem_get_dt(ea_problem *p, double *image, int *height, int *width);

/* Self-test of the wavefront reduction primitives the kernels rely on (write-masked DPP adds,
 * v_permlane16/32_swap, quad_perm): in = 32 slots x 64 lanes (fp32, slot-major); out32 / out64 = the 32
 * wave totals from the fp32 and the fp64 reduction; stages (nullable, 30 x 64 floats) = the
 * intermediate levels of the fp32 reduction. */
int ea_selftest_wave_reduce(int device, const float *in, double *out32, double *out64, float *stages);

#ifdef __cplusplus
}
#endif
#endif /* EA_HIP_H */
