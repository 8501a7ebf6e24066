/*
 * ea_hip_dev.h -- measurement hooks of libea_hip.so.  NOT part of the drop-in boundary (include/ea_hip.h): bench.py, the
 * A/B scripts and the tests that pin the launch patterns bind these through ctypes; a caller of the library has no use
 * for them.  Timing is done with HIP events on the stream the kernels are launched on.
 */
#ifndef EA_HIP_DEV_H
#define EA_HIP_DEV_H

#include "../../include/ea_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Upload the poses once, run `warmup` untimed then `steps` timed fused evaluations
 * (residual + Jacobian + JtJ/Jtr/cost reduction) back to back with the inputs resident in HBM.
 * ms_total: event time over the timed region; ms_eval_kernel: average duration of the
 * dominant per-point kernel alone (events around each launch in a second pass). */
int ea_batch_bench_eval(ea_batch *b, const double *q, const double *t, int warmup, int steps,
                        double *ms_total, double *ms_eval_kernel);
/* The timed region of bench.py and nothing else: `steps` x (fused evaluation + fold) enqueued on the batch's stream at the
 * poses the last ea_batch_bench_eval / ea_batch_eval uploaded, then a stream synchronisation.  No pose upload, no event
 * creation, no allocation inside: whoever brackets this call with a wall clock times exactly K steps (round 1's bracket
 * contained ~70 us of setup, a third of a 20-step run).  EA_ERR_STATE when no poses have been uploaded yet. */
int ea_batch_bench_steps(ea_batch *b, int steps, double *host_us /* nullable, 3 doubles: [0] enqueue and [1] wait in
                         microseconds of host time, [2] milliseconds between a HIP event pair around the region on the stream */);
/* Untimed set-up for it: capture `steps` x (evaluation + fold) into a hipGraph once; ea_batch_bench_steps with the same
 * `steps` then replays the graph (one host call, the launches execute back to back from the queue) instead of enqueueing
 * 2 x steps launches at ~3 us of host time each.  Dropped when the batch's problems or tuning change. */
int ea_batch_bench_capture(ea_batch *b, int steps);
/* The same K steps with the fold of step k-1 riding in the launch of evaluation k (one extra workgroup per problem; the K
 * passes of the timed region are independent evaluations at the resident poses, the fold's result is not an input of the
 * next one): K launches + one closing fold instead of 2 K dependent launches.  Every step still runs its evaluation and
 * its fold in full and the evaluation kernels execute one after the other.  The folds sum in the order of a workgroup of
 * the evaluation's size (for 256-thread launches not the order of ea_batch_eval's 1024-thread fold: equal to rounding).  Plain single-family problems on the L2 path; EA_ERR_STATE otherwise. */
int ea_batch_bench_capture_pipelined(ea_batch *b, int steps);
/* The same K launches + 1 enqueued launch by launch (no graph) and synchronised: the first evaluation runs while the host
 * enqueues the others.  host_us as in ea_batch_bench_steps.  Same restrictions as the captured form. */
int ea_batch_bench_steps_riding(ea_batch *b, int steps, double *host_us /* nullable, 3 doubles */);
/* cost / JtJ / Jtr / invalid count (layout of ea_batch_eval) that the LAST step of the last ea_batch_bench_steps left in
 * the batch's result array: a check that the timed launches compute what ea_batch_eval computes. */
int ea_batch_bench_result(ea_batch *b, double *cost, double *JtJ, double *Jtr, int64_t *n_invalid);
/* the last-but-one step's result of a pipelined sequence (a riding fold; ea_batch_bench_result reads the closing one) */
int ea_batch_bench_result_riding(ea_batch *b, double *cost, double *JtJ, double *Jtr, int64_t *n_invalid);
/* `launches` of the per-point kernel queued back to back between ONE event pair: average execution window per
 * launch (dispatch of the next launch overlaps the running one) -- the figure rocprofv3 --kernel-trace reports. */
int ea_batch_bench_kernel(ea_batch *b, const double *q, const double *t, int warmup, int launches,
                          double *ms_per_launch);
/* the same for the materialised-mode kernel; mode bit 0: LDS-staged row-major stores, bit 1: non-temporal stores;
 * r_dev / J_dev NULL: the library's own arrays */
int ea_batch_bench_rows(ea_batch *b, const double *q, const double *t, int corrected, int layout, int mode, void *r_dev,
                        void *J_dev, int64_t capacity_rows, int warmup, int launches, double *ms_per_launch);
/* the same for the fold kernel of ea_batch_eval, over the partial rows the last evaluation left */
int ea_batch_bench_fold(ea_batch *b, int warmup, int launches, double *ms_per_launch);
/* `reps` runs of the launches behind ea_batch_eval_resident_poses between one event pair on the batch's stream (held while
 * they are enqueued): milliseconds per run of the K resident poses; evaluations_only: the fold launches left out;
 * launches (nullable): evaluation launches per run */
int ea_batch_bench_resident_poses(ea_batch *b, int reps, int evaluations_only, double *ms_per_run, int *launches);
/* the floor of the launch mechanism: a hipGraph of `nodes` EMPTY kernels of grid x block threads replayed between one event
 * pair, milliseconds per node (best of four replays after the uploading one) */
int ea_bench_graph_floor(int device, int nodes, int grid, int block, double *ms_per_node);

#ifdef __cplusplus
}
#endif
#endif /* EA_HIP_DEV_H */
