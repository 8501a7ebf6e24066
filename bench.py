#!/usr/bin/env python3
"""bench.py — edge-point residual+Jacobian evaluations per second on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one frame pair's edge cloud with the inputs already
resident in HBM: the fused per-point kernel (SE(3) warp, pinhole, bicubic DT sample, analytic 1x6
row, IRLS weight, JtJ/Jtr/cost partials) plus the fixed-order fold of the partials.  The K timed steps
are ONE call of the product API ea_batch_eval_resident_poses (include/ea_hip.h): K INDEPENDENT
evaluations at K DIFFERENT poses -- the throughput a caller gets who asks for many evaluations at
once (cost-surface probes, a line search, multi-start, its own optimiser).  Independent evaluations
need not queue up behind each other: the pose is a batch dimension of the launch (G poses per
evaluation launch, every (point, pose) pair evaluated in full, every pose's partial rows folded).
It is NOT the cost of an evaluation inside the trust-region loop, where step k+1 depends on the
fold and the LM step of k: that dependent form (evaluation -> fold, two launches per step, one
pose per launch) is carried beside the headline as `value_serial_dependent_steps` /
roofline.one_pose_per_launch, and --serial-steps makes it the timed region; the loop itself is
measured by lm_iters_per_s_at_1e5_pts.  Default workload = BASELINE.json configs[1]
(C2): single 640x480 frame pair, 5e4 edge points, fp64.  With N GPUs every rank evaluates its own
independent frame pair (weak scaling, no data-path collective); the one collective is the pose
all-gather (RCCL) after the per-rank LM solves, reported separately.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_workload(name, seed_offset):
    from edge_alignment_amd import capi, synth
    if name == "c2":
        cfg = synth.config_c2_twin(seed=2 + seed_offset, n_points=50000)
        dtype, tag, loss = capi.EA_F64, "f64", (capi.LOSS_CAUCHY, 1.0)
        desc = "c2: single 640x480 frame pair, 50000 edge points, fp64, CauchyLoss(1)"
    elif name == "c5":
        cfg = synth.config_c5(seed=5 + seed_offset, n_points=1000000)
        dtype, tag, loss = capi.EA_F32, "f32", (capi.LOSS_TRIVIAL, 1.0)
        desc = "c5: 1e6-point edge cloud into a 2048x1536 DT image, fp32 (fp64 accumulation)"
    elif name == "lm1e5":
        cfg = synth.config_c2_twin(seed=7 + seed_offset, n_points=100000)
        dtype, tag, loss = capi.EA_F64, "f64", (capi.LOSS_CAUCHY, 1.0)
        desc = "lm1e5: 640x480 frame pair, 1e5 edge points, fp64"
    else:
        raise ValueError(name)
    return cfg, dtype, tag, loss, desc


SIMDS = 256 * 4                # MI355X: 256 CUs x 4 SIMDs
VALU_PEAK_IPS = SIMDS * 2.4e9 / 2.0  # one wave64 vector instruction per 2 cycles and SIMD at 2.4 GHz (guide: v_fma_f32 2 cyc)


def valu_issue(workload_key, kernel_ms, insts=None, source=None):
    """Second bound of the same kernel: vector-instruction issue.  SQ_INSTS_VALU per launch comes from
    the committed PMC pass (profiles/pmc_valu.json), the duration is the live one.  `frac` is against the 2-cycle issue
    peak of the guide; `frac_of_measured_ceiling` is against what this machine sustains on the kernel's mix of instruction
    classes (profiles/r01_issue_rates_2.txt: 1.05 ns for two-VGPR-operand forms, 1.75 ns for three-operand FMAs, DPP,
    conversions and every fp64 instruction, per wave64 instruction and SIMD at eight waves per SIMD)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))[workload_key]
    except Exception:
        return None
    n = insts if insts else d["sq_insts_valu_per_launch"]   # (insts: counted on the launch itself, e.g. a pose-batched one)
    achieved = n / (kernel_ms * 1e-3)
    c = d.get("cheap_class_fraction_static", 0.5)
    ceiling = SIMDS / ((c * 1.05 + (1.0 - c) * 1.75) * 1e-9)
    return {"bound": "valu_issue", "achieved": achieved, "peak": VALU_PEAK_IPS, "unit": "wave64 instr/s",
            "frac": achieved / VALU_PEAK_IPS, "measured_ceiling": ceiling, "frac_of_measured_ceiling": achieved / ceiling,
            "sq_insts_valu_per_launch": n, "source": source or "profiles/pmc_valu.json (rocprofv3 --pmc SQ_INSTS_VALU) / live kernel time"}


def step_poses(K, seed):
    """K different poses for the K timed steps: seeded perturbations of the identity (rotation <= 0.2 deg about a random
    axis, translation <= 5 mm per axis) -- every step samples other stencil fractions, no two steps repeat a launch"""
    from edge_alignment_amd import synth
    rng = np.random.default_rng(seed)
    Q, T = np.zeros((K, 1, 4)), np.zeros((K, 1, 3))
    for k in range(K):
        Q[k, 0] = synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0.0, 0.2)))
        T[k, 0] = rng.uniform(-0.005, 0.005, size=3)
    return Q, T


def counter_busy(workload_key, kernel_ms):
    """A kernel duration that is neither this script's hipEvent clock nor the tracer's bracket: GPU-busy cycles per launch
    from a committed PMC pass (profiles/pmc_busy.json: rocprofv3 --pmc GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES on this command) over
    the shader clock.  Reported beside the live figure with their ratio."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_busy.json")))[workload_key]
        d = {k: v for k, v in d.items() if k != "note"}
    except Exception:
        return None
    out = dict(d)
    if d.get("kernel_ms_from_counters"):
        out["live_over_counters"] = kernel_ms / d["kernel_ms_from_counters"]
    return out


def algorithmic_bytes(n_points, H, W, esize):
    """Fused mode: 3*s per point + the DT image once per launch."""
    return 3 * esize * n_points + H * W * esize


def cpu_baseline(cfg, loss, budget_s=12.0):
    """The oracle's Ceres-style path (Jet<7> autodiff of the functor + local parameterisation +
    loss correction + normal-equation accumulation) on ONE host core, on a bounded sample of the
    same workload."""
    from oracle import ea_oracle as eo
    O = eo.OracleProblem(cfg["grid"], *cfg["K"], loss=loss[0], loss_a=loss[1])
    n = min(cfg["xyz"].shape[0], 50000)
    X = np.ascontiguousarray(cfg["xyz"][:n])
    q, t = np.array([1.0, 0, 0, 0]), np.zeros(3)
    O.eval(X, q, t, eo.JAC_JET)
    reps, t0 = 0, time.perf_counter()
    while True:
        O.eval(X, q, t, eo.JAC_JET)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {"value": n * reps / el, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "%d passes over the first %d points of the same cloud, Jet<7> autodiff (Ceres-style) "
                      "fp64, %.1f s" % (reps, n, el)}


def cpu_baseline_all_cores(cfg, loss, budget_s=6.0):
    """Second CPU figure: the oracle's analytic-Jacobian evaluation -- what a hand-optimised CPU port would
    run, not what the reference runs -- on ALL host cores: one thread per core, each over its own contiguous shard of
    the same points (the C library releases the GIL)."""
    import threading
    from oracle import ea_oracle as eo
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    try:  # a container's CPU quota, when there is one, is the number of cores this process really has
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    cores = min(cores, 16 * max(1, int(os.environ.get("EA_BENCH_GPUS_ON_BOX", "1"))))  # the box's CPU share per GPU
    n = min(cfg["xyz"].shape[0], 50000)
    q, t = np.array([1.0, 0, 0, 0]), np.zeros(3)
    shards = [np.ascontiguousarray(cfg["xyz"][(k * n) // cores:((k + 1) * n) // cores]) for k in range(cores)]
    probs = [eo.OracleProblem(cfg["grid"], *cfg["K"], loss=loss[0], loss_a=loss[1]) for _ in range(cores)]
    for O, X in zip(probs, shards):
        O.eval(X, q, t, eo.JAC_ANALYTIC)
    stop = time.perf_counter() + budget_s
    done = [0] * cores

    def work(k):
        while time.perf_counter() < stop:
            probs[k].eval(shards[k], q, t, eo.JAC_ANALYTIC)
            done[k] += len(shards[k])
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    el = time.perf_counter() - t0
    return {"value": sum(done) / el, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "analytic-Jacobian oracle, %d threads x shards of the first %d points, fp64, %.1f s" % (cores, n, el)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """`python3 bench.py --gpus N` with no launcher in the environment: THIS process becomes the parent of N fresh rank
    processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU) and never touches the GPU itself -- no torch
    import, no HIP call, nothing re-executed in a process that has initialised a device.  Rank 0's stdout (the ONE JSON
    line) is relayed; every rank's stderr is inherited; the exit code is non-zero if any rank fails, and the others are
    terminated once one has failed (they would wait for it in a collective)."""
    import signal
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    env["EA_BENCH_GPUS_ON_BOX"] = env.get("EA_BENCH_GPUS_ON_BOX", str(n))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, cwd=ROOT))
    line = []

    def relay():
        for raw in procs[0].stdout:
            line.append(raw)
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc, failed_at = 0, None
    try:
        while any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                c = p.poll()
                if c not in (None, 0) and rc == 0:
                    rc, failed_at = c, time.perf_counter()
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, c))
            if failed_at is not None and time.perf_counter() - failed_at > 10.0:
                for p in procs:
                    if p.poll() is None:
                        p.send_signal(signal.SIGTERM)
                failed_at = time.perf_counter() + 1e9
            time.sleep(0.05)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        rc = 130
    for r, p in enumerate(procs):
        c = p.wait()
        if c != 0 and rc == 0:
            rc = c
    th.join(timeout=5.0)
    # the contract is ONE JSON line on stdout: anything else rank 0 (or a library under it: gloo announces its peers on
    # stdout) printed goes to stderr
    for raw in line:
        txt = raw.decode(errors="replace")
        (sys.stdout if txt.lstrip().startswith("{") else sys.stderr).write(txt)
    sys.stdout.flush()
    return rc if rc >= 0 else 128 - rc


def launcher_selftest(rank, world, args, line_out):
    """tests/test_bench_launcher.py (CPU): what a rank does with the environment launch_ranks() gave it, minus the GPU --
    rendezvous over gloo, one all-reduce, rank 0 prints a line carrying the world size every rank saw."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1), 1.0], dtype=torch.float64)
    dist.all_reduce(t)
    fail_rank = os.environ.get("EA_BENCH_LAUNCHER_SELFTEST_FAIL")
    if fail_rank is not None and int(fail_rank) == rank:
        os._exit(7)   # (a rank that dies: the parent must report it and stop the others)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        line_out.write(json.dumps({"selftest": True, "n_gpus": int(t[1].item()), "rank_sum": t[0].item(), "steps": args.steps,
                                   "master": os.environ.get("MASTER_ADDR") + ":" + os.environ.get("MASTER_PORT")}) + "\n")
        line_out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c2", choices=["c2", "c5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-device-allreduce", action="store_true",
                    help="with --gpus > 1: skip the last leg (ea_solve_sharded_device with its all-reduce enqueued on the stream over RCCL)")
    ap.add_argument("--device-allreduce", action="store_true", help=argparse.SUPPRESS)  # (round-2 spelling; now the default)
    ap.add_argument("--extras-timeout", type=float, default=240.0,
                    help="seconds the secondary measurements may take before the line is printed without the unfinished ones")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the timed steps launch by launch instead of replaying a hipGraph")
    ap.add_argument("--collective-barrier", action="store_true", help="N > 1: bracket the timed region with torch.distributed.barrier() instead of the shared-memory barrier")
    ap.add_argument("--serial-steps", action="store_true",
                    help="timed region as evaluation -> fold -> evaluation ... (two dependent launches per step) instead of the fold of step k-1 riding in the launch of evaluation k")
    # rehearsal knobs (the driver never passes them): run the N>1 control flow on a one-GPU box
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--force-device", type=int, default=None, help="use this HIP device on every rank")
    ap.add_argument("--force-dist", action="store_true",
                    help="world size 1: create the process group anyway and take the N>1 branches (RCCL on a one-rank group)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torch.distributed.run around us: start the N ranks ourselves (nothing below runs in this process)
        sys.exit(launch_ranks(args.gpus))
    # stdout carries ONE JSON line.  Libraries under this process talk on stdout too (RCCL's version banner at communicator
    # creation, gloo's peer report): file descriptor 1 points at stderr from here on, the line goes to the saved descriptor.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    line_out = os.fdopen(line_fd, "w")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if os.environ.get("EA_BENCH_LAUNCHER_SELFTEST"):
        return launcher_selftest(rank, world, args, line_out)

    import torch  # plumbing only: process group (RCCL), barriers, device sync
    dist = None
    if args.force_device is not None:
        local_rank = args.force_device
    coll_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    multi = world > 1 or args.force_dist   # the N>1 control flow (collectives on coll_dev)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29511"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    from edge_alignment_amd import capi
    from edge_alignment_amd import dist as ead
    if capi.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (gfx950): " + capi.load().ea_last_error().decode())
    # The collectives of the measured modes are the library's own (include/ea_hip.h ea_comm_*: ncclAllGather /
    # ncclAllReduce from librccl on the library's stream); torch.distributed carries the rendezvous (the 128-byte id), the
    # max-over-ranks of the timed region and -- in a gloo rehearsal, where RCCL cannot hold two ranks on one GPU -- the
    # exchanges themselves.  One rank: a one-rank communicator.  Created below, with the extras and under their watchdog: a
    # communicator meeting a real node for the first time must not stand between the process and its headline.
    comm, comm_error = None, None   # (created with the extras, under their watchdog: after the headline is complete)
    # The barrier of the timed bracket.  Ranks of one node: an epoch barrier through shared memory (a few microseconds;
    # edge_alignment_amd/dist.py NodeBarrier) -- the closing barrier sits INSIDE the timed region, and a collective-based one
    # costs as much as the K = 20 steps it brackets.  Anything unexpected: torch.distributed.barrier().
    node_barrier = None
    if dist is not None and not args.collective_barrier:
        try:
            node_barrier = ead.NodeBarrier(rank, max(world, 1))
        except Exception as e:
            sys.stderr.write("bench.py: shared-memory barrier unavailable (%r); using torch.distributed.barrier()\n" % (e,))
            node_barrier = None

    def barrier_sync():
        if node_barrier is not None:
            node_barrier.wait(timeout_s=3600.0)   # (a missing rank: the watchdog below prints the line, as with a collective barrier)
        elif dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    cfg, dtype, tag, loss, desc = build_workload(args.workload, rank)
    P = capi.Problem(*cfg["K"], dtype=dtype, device=local_rank)
    P.set_points(cfg["xyz"])
    P.set_dt_grid(cfg["grid"])
    P.set_loss(*loss)
    B = capi.Batch([P])
    n_pts = P.num_points
    H, W = cfg["image"].shape
    esize = 8 if dtype == capi.EA_F64 else 4
    q0, t0 = np.array([1.0, 0, 0, 0]), np.zeros(3)

    # Untimed: descriptors + pose upload + W warm-up steps (ea_batch_bench_eval builds the batch, uploads the poses and
    # creates its events), then the K poses of the timed region go up (inputs resident in HBM when the clock starts) and
    # their sequence is run once (captures the graph) plus EA_BENCH_WARM_REPLAYS times more.
    # Timed: EXACTLY K steps between barrier+sync brackets = ONE call of the product API ea_batch_eval_resident_poses --
    # K evaluations at K DIFFERENT poses (no two steps read the same stencils), every (point, pose) pair through the whole
    # per-point arithmetic and every pose's partial rows through the fold; the pose is a batch dimension of the launch (G
    # poses per evaluation launch + one fold launch, ceil(K / G) such pairs), one synchronisation, the K results unpacked
    # into the caller's arrays.  --serial-steps: the LM loop's dependency instead (evaluation -> fold, two dependent
    # launches per step, at one pose; a measurement hook).
    B.bench_eval(q0, t0, 0, max(args.warmup, 1), kernel_pass=False)
    Qk, Tk = step_poses(args.steps, 1000 + rank)
    mode = "serial" if (args.serial_steps or args.no_graph) else "poses"
    graph, pipelined, out_k, warm_replays, G = None, False, None, 0, 1
    if mode == "poses":
        # This caller ends its timed bracket on a device synchronisation (the contract): it waits for the stream's completion
        # signal inside the call rather than returning on the library's done flag ~6 us earlier and paying a cold device
        # wait of ~11 us right behind it (profiles/r03_ab_poll.txt: 33.6 against 39.0 us per K = 20 call + torch.cuda.synchronize;
        # a caller that consumes the results at once sees 25.5 against 31.2 us with the flag, the library's default).
        B.set_tuning("poll_results", 0)
        B.set_poses(Qk, Tk)
        out_k = B.eval_resident_poses()
        pipelined = True
        G = int(B.info("poses_per_launch"))
        graph = ("ea_batch_eval_resident_poses: %d evaluations at %d different poses (rotations <= 0.2 deg, translations <= 5 mm "
                 "around the identity), one call; %d evaluation launch(es) of up to %d poses each (grid = chunks x poses: every "
                 "(point, pose) pair is evaluated in full) + as many fold launches, one synchronisation, results unpacked"
                 % (args.steps, args.steps, -(-args.steps // G), G))
        warm_replays = int(os.environ.get("EA_BENCH_WARM_REPLAYS", "2"))
    elif not args.no_graph:
        try:
            B.bench_capture(args.steps)
            graph = "hipGraph of %d dependent steps at one pose (evaluation -> fold, 2 kernel nodes per step), one replay" % args.steps
            warm_replays = int(os.environ.get("EA_BENCH_WARM_REPLAYS", "2"))
        except capi.EAError as e:
            graph = "eager launches (graph capture failed: %s)" % e
    # The untimed runs of the region are rehearsals of the WHOLE bracket (barrier + sync, clock, the call, barrier + sync, clock),
    # so that the one that counts -- always the last, never the best -- does not also pay for the first execution of the
    # bracket's own host code (a one-shot bracket reads ~45 us where the same bracket in a loop reads ~37:
    # scripts/archive/probe_oneshot.py).
    for _ in range(warm_replays + 1):
        barrier_sync()
        t_start = time.perf_counter()
        if mode == "poses":
            B.eval_resident_poses(out=out_k)
        else:
            B.bench_steps(args.steps)
        barrier_sync()
        elapsed = time.perf_counter() - t_start
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = world * n_pts * args.steps / elapsed
    # what the bracket itself costs on idle GPUs (N > 1: the closing barrier is a collective inside the timed region; at
    # K = 20 steps of ~3 us it is comparable to the work): reported beside the value, not subtracted from it
    bracket_ms = None
    if dist is not None:
        costs = []
        for _ in range(5):
            tb = time.perf_counter()
            barrier_sync()
            costs.append(time.perf_counter() - tb)
        bracket_ms = min(costs) * 1e3
    # the timed steps computed what ea_batch_eval computes at their poses (first, middle and last step checked; the riding
    # folds sum in another order: equal to rounding)
    if mode == "poses":
        # (the pose-batched launch cuts the points into other chunks than a single evaluation: fp64 sums agree to rounding of
        # the fold, fp32 ones to the rounding of the per-lane / per-wavefront fp32 partial sums)
        rt = 1.0 if esize == 8 else 1e6
        for k in sorted({0, args.steps // 2, args.steps - 1}):
            want = B.eval(Qk[k], Tk[k])
            sc = np.abs(want["JtJ"]).max()
            if not (np.allclose(out_k["cost"][k], want["cost"], rtol=1e-12 * rt, atol=0) and np.allclose(out_k["JtJ"][k], want["JtJ"], rtol=1e-11 * rt, atol=1e-12 * rt * sc)
                    and np.allclose(out_k["Jtr"][k], want["Jtr"], rtol=1e-10 * rt, atol=1e-12 * rt * np.sqrt(sc * max(float(np.max(want["cost"])), 1e-300))) and np.array_equal(out_k["n_invalid"][k], want["n_invalid"])):
                raise SystemExit("bench.py: timed step %d differs from ea_batch_eval at its pose" % k)
        if args.steps > 1 and np.array_equal(out_k["cost"][0], out_k["cost"][-1]):
            raise SystemExit("bench.py: the timed steps were not evaluated at different poses")
    else:
        got, want = B.bench_result(), B.eval(q0, t0)
        if not (np.allclose(got["cost"], want["cost"], rtol=1e-12, atol=0) and np.allclose(got["JtJ"], want["JtJ"], rtol=1e-12, atol=1e-300)
                and np.allclose(got["Jtr"], want["Jtr"], rtol=1e-11, atol=1e-300) and np.array_equal(got["n_invalid"], want["n_invalid"])):
            raise SystemExit("bench.py: the timed steps' result differs from ea_batch_eval's")
    # the product calls by the wall clock, beside the headline: one ea_batch_eval (launch pair + synchronisation), and
    # ea_batch_eval_poses all in one (pose upload and device-side pose constants inside)
    B.set_tuning("poll_results", 1)   # (the library's default: return on the done flag)
    B.eval(q0, t0)
    tl = time.perf_counter()
    for _ in range(50):
        B.eval(q0, t0)
    single_eval_ms = (time.perf_counter() - tl) / 50 * 1e3
    eval_poses_call_ms = None
    if mode == "poses":
        B.eval_poses(Qk, Tk)
        tl = time.perf_counter()
        for _ in range(5):
            B.eval_poses(Qk, Tk)
        eval_poses_call_ms = (time.perf_counter() - tl) / 5 * 1e3
    B.bench_eval(q0, t0, 0, 1, kernel_pass=False)  # (poses resident again for the measurements below)

    # Duration of the dominant kernel, HIP events on the library's stream.
    #   kernel_ms               (event pair around the timed call's evaluation launches, executing back to back from the
    #                           queue) / their number: the average duration of the dominant kernel -- the evaluation launch
    #                           of G poses -- the figure rocprofv3 --kernel-trace reports per launch for this command
    #                           (profiles/).  `achieved` / `frac` = G x (one evaluation's algorithmic bytes) / kernel_ms;
    #   launch_floor_ms         an EMPTY kernel of the same grid in a replayed graph: what the launch mechanism costs per
    #                           node whatever the kernel does; frac_ceiling_at_floor = algorithmic bytes / floor / peak;
    #   one_pose_per_launch     the same kernel at one pose per launch (an LM iteration's launch, rounds 1-2's headline
    #                           kernel): back-to-back duration, its floor and ceiling, the dependent two-launch step;
    #   serial region (--serial-steps): the evaluation's share of a dependent step.
    nk = min(max(args.steps, 100), 1000)
    ms_steps, ms_kernel_isolated = B.bench_eval(q0, t0, 2, min(args.steps, 64))
    ms_steps, _ = B.bench_eval(q0, t0, 20, nk, kernel_pass=False)       # launch by launch (host-bound: ~3.1 us per launch)
    step_ms_eager = ms_steps / nk
    try:
        B.bench_capture(nk)
        step_ms_serial = min(B.bench_steps(nk, host_times=True)[2] for _ in range(3)) / nk
    except capi.EAError:
        step_ms_serial = step_ms_eager
    ms_fold = B.bench_fold(10, nk)
    ms_kernel_b2b = B.bench_kernel(q0, t0, 10, nk)          # one pose per launch (the LM loop's launch), back to back
    bytes_eval = algorithmic_bytes(n_pts, H, W, esize)      # ONE evaluation of every point: 3 s per point + the image once
    poses_launch = launches = None
    if pipelined:
        # The timed call's own launches (the K poses are resident again), between one event pair on the library's stream:
        # with and without the fold launches.  The dominant kernel is the evaluation launch of G poses; its duration is the
        # evaluations-only time / number of launches, its algorithmic bytes G x (one evaluation's bytes).
        B.set_poses(Qk, Tk)
        ms_run, launches = min(B.bench_resident_poses(5) for _ in range(3))
        ms_evals, _ = min(B.bench_resident_poses(5, evaluations_only=True) for _ in range(3))
        B.bench_eval(q0, t0, 0, 1, kernel_pass=False)
        poses_launch = args.steps / launches
        ms_kernel = ms_evals / launches
        bytes_launch = bytes_eval * poses_launch
        step_ms = ms_run / args.steps
    else:
        step_ms = step_ms_serial
        ms_kernel = max(step_ms - ms_fold, ms_kernel_b2b)
        bytes_launch = bytes_eval
    achieved = bytes_launch / (ms_kernel * 1e-3) / 1e9
    try:   # the launch mechanism's floor for THIS grid: an empty kernel of as many workgroups in a replayed graph
        floor_ms = capi.graph_floor_ms(local_rank, nodes=max(8, min(nk, 20000 // max(1, int(poses_launch or 1)))),
                                       grid=max(1, int((B.info("poses_tiles") if pipelined else B.info("num_tiles")) * (poses_launch or 1))),
                                       block=int(B.info("poses_threads") if pipelined else B.info("threads")))
        floor_one_ms = capi.graph_floor_ms(local_rank, nodes=nk, grid=max(1, int(B.info("num_tiles"))), block=int(B.info("threads")))
    except capi.EAError:
        floor_ms = floor_one_ms = None
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            tj = json.load(open(pmc_path))
            traffic = (tj.get(args.workload + "_poses_%d" % args.steps) if pipelined else None) or {}
            traffic = traffic.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": ("ea_eval_poses_kernel<%s>" if pipelined else "ea_eval_fused_kernel<%s>") % ("double" if esize == 8 else "float"),
                "kernel_ms": ms_kernel,
                "counter_busy": counter_busy("%s_poses_%d" % (args.workload, args.steps), ms_kernel) if pipelined else None,
                "evaluation_launches_in_timed_region": launches, "poses_per_launch": poses_launch,
                "launch_shape": {"threads": int(B.info("poses_threads")), "points_per_thread": int(B.info("poses_points_per_thread")),
                                 "workgroups_per_pose": int(B.info("poses_tiles"))} if pipelined else None,
                "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes_per_evaluation": bytes_eval,
                "launch_floor_ms": floor_ms,
                # (what the launch mechanism alone would allow this grid; above 1 it has stopped being the bound)
                # (what the launch mechanism alone allows: 1.0 = the launch is long enough for the floor not to bind)
                "frac_ceiling_at_floor": min(1.0, bytes_launch / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if floor_ms else None,
                # the same kernel with ONE pose per launch (what an LM iteration launches; rounds 1-2's headline kernel):
                # a launch of 196 workgroups is bounded by the launch mechanism, not by the chip
                "one_pose_per_launch": {"kernel_ms_back_to_back": ms_kernel_b2b,
                                        "frac": bytes_eval / (ms_kernel_b2b * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "launch_floor_ms": floor_one_ms,
                                        "frac_ceiling_at_floor": min(1.0, bytes_eval / (floor_one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if floor_one_ms else None,
                                        "kernel_ms_isolated": ms_kernel_isolated, "fold_kernel_ms": ms_fold,
                                        "step_ms_events_serial_dependent": step_ms_serial, "step_ms_events_eager_launches": step_ms_eager,
                                        "counter_busy": counter_busy(args.workload, ms_kernel_b2b)},
                "step_ms_events": step_ms, "step_ms_events_serial_dependent": step_ms_serial,
                # (fp64 over a float32-stored image, when every texel is float-representable: same doubles, 4-byte texels read;
                # achieved / frac stay on the algorithmic 8-byte definition)
                "image_texel_bytes_read": 4 if B.info("dt_f32") else esize,
                "secondary": None}
    if pipelined:
        cb = roofline["counter_busy"] or {}
        if cb.get("sq_insts_valu_per_launch") and cb.get("grid_size"):
            # instructions counted on a full launch of the committed PMC pass -> per pose -> this run's average launch.
            # (one problem per batch here: the grid is 8 * ceil(workgroups per pose / 8) columns x poses x threads)
            cols = -(-int(B.info("poses_tiles")) // 8) * 8
            poses_pmc = cb["grid_size"] / float(cols * int(B.info("poses_threads")))
            roofline["secondary"] = valu_issue(args.workload, ms_kernel, insts=cb["sq_insts_valu_per_launch"] / poses_pmc * poses_launch,
                                               source="profiles/pmc_busy.json (rocprofv3 --pmc SQ_INSTS_VALU on ea_eval_poses_kernel) / live kernel time")
        if roofline["secondary"] is None:
            roofline["secondary"] = valu_issue(args.workload, ms_kernel / max(1.0, float(poses_launch or 1)))
    else:
        roofline["secondary"] = valu_issue(args.workload, ms_kernel)

    # Materialised mode of the same workload ("report both numbers"): r and the 1x6 row of every point written
    # out in the batch's dtype (ea_batch_eval_rows_device), the bandwidth-bound form of the path: 3 s in + 7 s out per
    # point + one pass over the DT image.  Back-to-back launches between one event pair on the library's stream.
    def materialised(Bx, Q, T, n_points, images, es, launches=100):
        msr = min(Bx.bench_rows(Q, T, 5, launches, corrected=True, layout=0, mode=1) for _ in range(2))
        by = 10 * es * n_points + sum(h * w * es for h, w in images)
        # An fp64 batch whose images are exactly float-representable reads a float32 mirror of them (same doubles after the
        # widening, bit-identical results): `frac` stays on the ALGORITHMIC bytes -- 8-byte texels, the definition the rounds
        # compare on, which this form can push past 1 -- and `frac_bytes_moved` is what the kernel really moves.
        tex = 4 if Bx.info("dt_f32") else es
        moved = 10 * es * n_points + sum(h * w * tex for h, w in images)
        return {"kernel": "ea_eval_rows_kernel<%s>" % ("double" if es == 8 else "float"), "kernel_ms": msr,
                "evals_per_s": n_points / (msr * 1e-3), "algorithmic_bytes_per_launch": by, "achieved": by / (msr * 1e-3) / 1e9,
                "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": by / (msr * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "image_texel_bytes_read": tex, "bytes_moved_per_launch": moved, "frac_bytes_moved": moved / (msr * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "layout": "r [rows], J row-major [rows][6] through a wavefront-local LDS transpose, corrected rows"}
    try:
        mat = materialised(B, q0, t0, n_pts, [(H, W)], esize)
    except capi.EAError as e:
        mat = {"error": str(e)}
    B.bench_eval(q0, t0, 0, 1, kernel_pass=False)

    # The line's mandatory part is complete here.  Everything below is secondary; a watchdog prints the line without the
    # unfinished measurements if they take longer than --extras-timeout (a collective meeting a real node for the first
    # time must not cost the scaling record) and ends the process; every rank runs one.
    base = None
    if rank == 0:
        base = {"metric": "edge-point residual+Jacobian evals/sec", "value": value, "unit": "evals/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": tag, "data": "synthetic",
                "config": {"workload": desc, "points_per_gpu": int(n_pts), "dt_image": "%dx%d" % (W, H),
                           "parallelism": "independent frame pairs, one per GPU; single RCCL pose all-gather",
                           "tiles": B.info("num_tiles"), "points_per_thread": B.info("points_per_thread"),
                           "lds_bytes": B.info("lds_bytes"), "point_order_tile_px": P.point_order,
                           "timed_region": graph or "eager launches",
                           "untimed_warmup": "%d launch-by-launch steps at one pose, the upload of the K poses, one run of the timed call, then %d rehearsals of the whole bracket (the last bracket is the one reported)" % (max(args.warmup, 1), warm_replays)
                                             if mode == "poses" else "%d launch-by-launch steps, then %d rehearsals of the whole bracket" % (max(args.warmup, 1), warm_replays),
                           "closing_barrier_and_sync_ms_on_idle_gpus": bracket_ms,
                           "bracket_barrier": None if dist is None else ("shared-memory epoch barrier (one node)" if node_barrier is not None else "torch.distributed.barrier")},
                # the same workload with the trust-region loop's dependency (evaluation -> fold, two dependent launches per
                # step, replayed from a graph): what ONE evaluation costs when the next one needs its result
                "value_serial_dependent_steps": world * n_pts / (step_ms_serial * 1e-3) if step_ms_serial else None,
                "single_eval_call_ms": single_eval_ms, "eval_poses_call_ms": eval_poses_call_ms,
                "calls": {"timed": "ea_batch_eval_resident_poses (K poses resident)" if mode == "poses" else "ea_batch_bench_steps (hook)",
                          "single_eval_call_ms": "ea_batch_eval: pose upload, evaluation, fold, synchronisation -- wall clock",
                          "eval_poses_call_ms": "ea_batch_eval_poses(K = steps): pose upload + device-side pose constants + the K evaluations -- wall clock"},
                "roofline": roofline, "materialised_mode": mat}
    extras, others, leg = {}, {}, ["start"]

    def compose(note=None):
        out = dict(base)
        out.update(extras)
        if others:
            out["other_workloads"] = dict(others)
        out.setdefault("cpu_baseline", None)
        if note:
            out["extras_incomplete"] = note
        return out

    def on_timeout():
        if rank == 0:
            line_out.write(json.dumps(compose("timed out after %.0f s in: %s" % (args.extras_timeout, leg[0]))) + "\n")
            line_out.flush()
        # a leg that hangs is a finding, not an "ok": the line above keeps the mandatory part of the record, the exit
        # code and stderr say what stalled so that it can be fixed once
        try:
            last = capi.load().ea_last_error().decode()
        except Exception:
            last = "?"
        sys.stderr.write("bench.py: rank %d: extras watchdog fired after %.0f s in leg %r; last library error: %r\n"
                         % (rank, args.extras_timeout, leg[0], last))
        sys.stderr.flush()
        try:
            if node_barrier is not None:
                node_barrier.close()
        except Exception:
            pass
        os._exit(3)
    watchdog = threading.Timer(args.extras_timeout, on_timeout)
    watchdog.daemon = True
    if not args.no_extras:
        watchdog.start()
        leg[0] = "RCCL communicator (ea_comm_create)"
        if multi and args.dist_backend == "nccl":
            try:
                comm = capi.Comm.from_process_group(device=local_rank)
            except Exception as e:   # (the torch.distributed forms of the same exchanges take over; the line says so)
                comm_error = repr(e)
            ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:   # every rank or none
                comm.close(); comm = None
        elif not multi:
            try:
                comm = capi.Comm(capi.comm_unique_id(), 1, 0, device=local_rank)   # (N = 1: the same calls on a one-rank communicator)
            except Exception as e:
                comm_error = repr(e)
        if comm_error:
            extras["comm_error"] = comm_error

    if not args.no_extras:
        leg[0] = "LM iterations/s at 1e5 points"
        # LM iterations/s at 1e5 points (second headline of BASELINE.json), device-resident loop
        cfg2, dt2, _, loss2, _ = build_workload("lm1e5", rank)
        P2 = capi.Problem(*cfg2["K"], dtype=dt2, device=local_rank)
        P2.set_points(cfg2["xyz"]); P2.set_dt_grid(cfg2["grid"]); P2.set_loss(*loss2)
        P2.solve(q0, t0)
        barrier_sync()
        ts = time.perf_counter()
        reps, its, lib_ms = 40, 0, 0.0
        for _ in range(reps):
            q, t, s = P2.solve(q0, t0)
            its += s["num_iterations"]
            lib_ms += s["total_time_ms"]  # the library's own clock around ea_solve: what a C++ caller sees
        torch.cuda.synchronize()
        el = time.perf_counter() - ts
        # the library's own clock around every ea_solve (call entry to return): what a C / C++ caller -- the reference's
        # language -- sees; the Python wrapper adds ~8 us per solve of ctypes marshalling, reported beside it
        lm_local = its / (lib_ms * 1e-3)
        # the one collective: all-gather of the solved poses (7 doubles + status per problem)
        pg1 = ead.PoseGather(1, world, device=coll_dev if multi else "cpu", force_collective=multi, comm=comm)  # buffers allocated once, outside the clock
        pg1.gather([q], [t], [s["termination"]])
        tg = time.perf_counter()
        qa, ta, st = pg1.gather([q], [t], [s["termination"]])
        gather_ms = (time.perf_counter() - tg) * 1e3
        if dist is not None:
            tt = torch.tensor([lm_local], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            lm_total = float(tt.item())
        else:
            lm_total = lm_local
        from edge_alignment_amd import synth
        err_rot = synth.rotation_angle_between(q, cfg2["q_true"])
        err_t = float(np.linalg.norm(t - cfg2["t_true"]))
        extras.update({"lm_iters_per_s_at_1e5_pts": lm_total, "lm_iters_per_s_at_1e5_pts_clock": "inside ea_solve (entry to return), summed over ranks",
                  "lm_iters_per_s_at_1e5_pts_through_ctypes": its / el, "lm_iters_per_s_at_1e5_pts_library_clock": its / (lib_ms * 1e-3),
                  "lm_iterations_per_solve": its / reps,
                  "lm_solve_ms": lib_ms / reps, "lm_solve_ms_through_ctypes": el / reps * 1e3, "pose_gather_ms": gather_ms,
                  "lm_pose_err_vs_planted": {"rad": err_rot, "m": err_t}})
        extras["lm_form"] = ("one launch per iteration: every workgroup of the evaluation folds the previous rows, takes the trust-region "
                             "step itself and evaluates at the candidate pose (ea_lm_iter_kernel); iterates bit-identical to the "
                             "(evaluate, step) pairs")
        if rank == 0 and world == 1:
            # the same solves as (evaluate, step) pairs -- rounds 1-2's form of the loop, still what batches and large problems run
            try:
                Bp = capi.Batch([P2])
                Bp.set_tuning("fused_iterations", 0)
                Bp.solve(q0, t0)
                itsp, libp = 0, 0.0
                for _ in range(reps):
                    qp, tp, sp = Bp.solve(q0, t0)
                    itsp += sp[0]["num_iterations"]
                    libp += sp[0]["total_time_ms"]
                extras["lm_iters_per_s_at_1e5_pts_pair_form"] = {"iters_per_s": itsp / (libp * 1e-3), "solve_ms": libp / reps,
                                                                 "same_pose_bits": bool(np.array_equal(qp[0], q) and np.array_equal(tp[0], t))}
                Bp.close()
            except Exception as e:
                extras["lm_iters_per_s_at_1e5_pts_pair_form"] = {"error": repr(e)}
            # the same problem in fp32 (the arithmetic of BASELINE configs C3 / C5; pose tolerance 1e-4 rad / 1e-3 m)
            try:
                P2f = capi.Problem(*cfg2["K"], dtype=capi.EA_F32, device=local_rank)
                P2f.set_points(cfg2["xyz"]); P2f.set_dt_grid(cfg2["grid"]); P2f.set_loss(*loss2)
                P2f.solve(q0, t0)
                tsf = time.perf_counter()
                itsf = 0
                for _ in range(reps):
                    qf, tf, sf = P2f.solve(q0, t0)
                    itsf += sf["num_iterations"]
                elf = time.perf_counter() - tsf
                extras["lm_fp32_at_1e5_pts"] = {"iters_per_s": itsf / elf, "solve_ms": elf / reps * 1e3,
                                                "iterations_per_solve": itsf / reps,
                                                "pose_err_vs_planted": {"rad": synth.rotation_angle_between(qf, cfg2["q_true"]),
                                                                        "m": float(np.linalg.norm(tf - cfg2["t_true"]))}}
                P2f.close()
            except Exception as e:
                extras["lm_fp32_at_1e5_pts"] = {"error": repr(e)}
        P2.close()

    # BASELINE config C4 as a run shape (every rank, collective): 32 frame pairs per GPU built on the device from the
    # bundled grabs (edge points of frame A, distance transform of frame B), ONE ea_batch_solve, ONE all-gather of the
    # 32 x 8 doubles from preallocated device tensors.
    if not args.no_extras:
        leg[0] = "C4 run shape (32 pairs per GPU, one pose all-gather)"
        c4 = None
        solve_fn = None
        try:
            from edge_alignment_amd import synth
            frames = synth.load_bundled_frames(os.path.join(ROOT, "tests", "golden", "rgbd"))
            keep = []

            def build_and_solve(specs):
                Ps = []
                for a, b, _, _ in specs:
                    Px = capi.Problem(*synth.TUM_K, dtype=capi.EA_F64, device=local_rank)
                    Px.set_ref_frame(frames[a][0], frames[a][1], z_scaling=5000.0)
                    Px.set_now_frame(frames[b][0])
                    Px.set_loss(capi.LOSS_CAUCHY, 1.0)
                    Ps.append(Px)
                Bx = capi.Batch(Ps)
                keep.extend([Bx] + Ps)
                Q = np.stack([sp[2] for sp in specs]); T = np.stack([sp[3] for sp in specs])
                npts = [Px.num_points for Px in Ps]
                return (lambda: Bx.solve(Q, T)), {"dtype": "f64", "points_per_pair_mean": float(np.mean(npts)),
                                                  "source": "20 ordered pairs of the 5 bundled TUM grabs x seeded start poses (seed 4)"}
            ok = 1
        except Exception as e:  # the fixtures are part of the repository; never lose the headline over them
            ok, c4 = 0, {"error": repr(e)}
        if dist is not None:
            tt = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            ok = int(tt.item())
        if ok:
            c4, _ = ead.run_c4(rank, world, build_and_solve, per_gpu=32, device=coll_dev if multi else "cpu", repeats=3, force_collective=multi, comm=comm)
            if dist is not None:
                tt = torch.tensor([c4["lm_iters_per_s_per_gpu"], c4["evals_per_s_per_gpu"]], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.SUM)
                c4["lm_iters_per_s"], c4["evals_per_s"] = float(tt[0].item()), float(tt[1].item())
                tm = torch.tensor([c4["solve_ms"], c4["pose_gather_ms"]], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                c4["solve_ms"], c4["pose_gather_ms"] = float(tm[0].item()), float(tm[1].item())
            else:
                c4["lm_iters_per_s"], c4["evals_per_s"] = c4["lm_iters_per_s_per_gpu"], c4["evals_per_s_per_gpu"]
            for x in keep:
                x.close()
        extras["c4_batch_32_pairs_per_gpu"] = c4

    # Throughput-regime context for the same kernel (not the headline value): the C5 roofline-stress
    # cloud and a C4-style batch of 32 C2-shaped frame pairs evaluated by one launch.
    if not args.no_extras and rank == 0 and world == 1:
        leg[0] = "other workloads"
        from edge_alignment_amd import synth
        def measure(problems, dtype, esz, loss, tile=None, valu_key=None, traffic_key=None):
            Ps = []
            for cfgx in problems:
                Px = capi.Problem(*cfgx["K"], dtype=dtype, device=local_rank)
                if tile is not None:
                    Px.set_point_order(tile)
                Px.set_points(cfgx["xyz"]); Px.set_dt_grid(cfgx["grid"]); Px.set_loss(*loss)
                Ps.append(Px)
            Bx = capi.Batch(Ps)
            m = len(Ps)
            Q = np.tile(q0, (m, 1)); T = np.zeros((m, 3))
            ms, _ = Bx.bench_eval(Q, T, 10, 100, kernel_pass=False)
            msk = Bx.bench_kernel(Q, T, 5, 100)
            npts = sum(Px.num_points for Px in Ps)
            by = sum(algorithmic_bytes(Px.num_points, cfgx["image"].shape[0], cfgx["image"].shape[1], esz)
                     for Px, cfgx in zip(Ps, problems))
            us_serial = ms / 100 * 1e3
            us_step = us_serial
            try:  # independent evaluations of the whole batch at one pose: fold of step k-1 riding in evaluation k, 100 steps replayed from a graph
                Bx.bench_capture_pipelined(100)
                us_step = min(Bx.bench_steps(100, host_times=True)[2] for _ in range(3)) / 100 * 1e3
            except capi.EAError:
                pass
            res = {"evals_per_s": npts / (us_step * 1e-6), "us_per_step": us_step, "us_per_step_serial_launches": us_serial, "kernel_us": msk * 1e3,
                   "roofline_frac": by / (msk * 1e-3) / 1e9 / HBM_PEAK_GBS, "image_texel_bytes_read": 4 if Bx.info("dt_f32") else esz, "points": int(npts),
                   "point_order_tile_px": Ps[0].point_order}
            try:  # the product call for independent evaluations: K different poses per problem, the pose a batch dimension of the launch
                Kp = 8
                rngp = np.random.default_rng(9)
                Qp = np.stack([Q] * Kp); Tp = np.stack([T + rngp.uniform(-0.005, 0.005, size=T.shape) for _ in range(Kp)])
                Bx.set_poses(Qp, Tp)
                ms_run, nl = min(Bx.bench_resident_poses(3) for _ in range(2))
                ms_ev, _ = min(Bx.bench_resident_poses(3, evaluations_only=True) for _ in range(2))
                res["k_poses"] = {"K": Kp, "evaluation_launches": nl, "us_per_evaluation": ms_run / Kp * 1e3, "evals_per_s": npts * Kp / (ms_run * 1e-3),
                                  "kernel_us_per_launch": ms_ev / nl * 1e3, "roofline_frac": by * Kp / (ms_ev * 1e-3) / 1e9 / HBM_PEAK_GBS}
                Bx.bench_eval(Q, T, 0, 1, kernel_pass=False)
            except capi.EAError as e:
                res["k_poses"] = {"error": str(e)}
            if valu_key:
                res["valu_issue"] = valu_issue(valu_key, msk)
            if traffic_key:   # HBM-side bytes per launch from the PMC passes (profiles/pmc_traffic.json) over the live kernel time
                try:
                    tb = json.load(open(pmc_path)).get(traffic_key, {}).get("hbm_bytes_per_launch")
                    if tb:
                        res["traffic"] = tb
                        res["traffic_GBps"] = tb / (msk * 1e-3) / 1e9
                        res["traffic_frac_of_measured_stream_copy_6290GBps"] = tb / (msk * 1e-3) / 1e9 / 6290.0
                except Exception:
                    pass
            try:
                mm = materialised(Bx, Q, T, npts, [cfgx["image"].shape for cfgx in problems], esz, launches=50)
                res["materialised_mode"] = {k: mm[k] for k in ("kernel_ms", "evals_per_s", "achieved", "frac", "algorithmic_bytes_per_launch", "image_texel_bytes_read", "frac_bytes_moved")}
            except capi.EAError as e:
                res["materialised_mode"] = {"error": str(e)}
            if m > 1 and tile is None:  # the production shape of BASELINE config C4: all frame pairs of a GPU solved by one launch sequence
                Bx.solve(Q, T)
                tsv = time.perf_counter()
                for _ in range(5):
                    qs, tsol, ss = Bx.solve(Q, T)
                elv = (time.perf_counter() - tsv) / 5
                res["solve_ms"] = elv * 1e3
                res["lm_iters_per_s"] = sum(x["num_iterations"] for x in ss) / elv
            Bx.close()
            for Px in Ps:
                Px.close()
            return res
        if args.workload != "c5":
            others["c5_fp32_1e6pts_2048x1536"] = measure([synth.config_c5()], capi.EA_F32, 4, (capi.LOSS_TRIVIAL, 1.0), valu_key="c5")
        # C3: 3-level pyramid (1280x960 / 640x480 / 320x240, 2.0e5 points in total), fp32, coarse-to-fine solve
        lv = synth.config_c3_levels()
        Pl = []
        for cfgx in lv:
            Px = capi.Problem(*cfgx["K"], dtype=capi.EA_F32, device=local_rank)
            Px.set_points(cfgx["xyz"]); Px.set_dt_grid(cfgx["grid"])
            Pl.append(Px)
        capi.solve_pyramid(Pl, q0, t0)
        t3 = time.perf_counter()
        for _ in range(10):
            q3, tt3, ss3 = capi.solve_pyramid(Pl, q0, t0)
        el3 = (time.perf_counter() - t3) / 10
        its3 = [x["num_iterations"] for x in ss3]
        evals3 = sum(x["num_point_evals"] for x in ss3)
        others["c3_pyramid_fp32_2e5pts"] = {"solve_ms": el3 * 1e3, "iterations_per_level_fine_to_coarse": its3,
                                            "evals_per_s": evals3 / el3, "lm_iters_per_s": sum(its3) / el3,
                                            "pose_err_vs_planted": {"rad": synth.rotation_angle_between(q3, lv[0]["q_true"]),
                                                                    "m": float(np.linalg.norm(tt3 - lv[0]["t_true"]))}}
        for Px in Pl:
            Px.close()
        batch = [synth.config_c2_twin(seed=100 + i) for i in range(32)]
        others["batch32_c2_fp64"] = measure(batch, capi.EA_F64, 8, (capi.LOSS_CAUCHY, 1.0), valu_key="batch32_c2_fp64")
        others["batch32_c2_fp32"] = measure(batch, capi.EA_F32, 4, (capi.LOSS_CAUCHY, 1.0), valu_key="batch32_c2_fp32")
        # the same batch with the points stored tile by tile (ea_problem_set_point_order(16): what the storage order of
        # large clouds buys, applied to frame-sized clouds the caller intends to evaluate many times)
        others["batch32_c2_fp64_tile16"] = measure(batch, capi.EA_F64, 8, (capi.LOSS_CAUCHY, 1.0), tile=16)
        others["batch32_c2_fp32_tile16"] = measure(batch, capi.EA_F32, 4, (capi.LOSS_CAUCHY, 1.0), tile=16)
        # 64 pairs per launch: where the launch has grown out of its ramp and tail (profiles/r02_batch_size_sweep.txt);
        # 256 pairs (fp32: 468 MB): beyond the 256 MB Infinity Cache that serves repeated launches over a smaller
        # batch, i.e. the rate with every byte coming from HBM
        others["batch64_c2_fp32_tile16"] = measure(batch * 2, capi.EA_F32, 4, (capi.LOSS_CAUCHY, 1.0), tile=16, traffic_key="batch64_c2_fp32_tile16")
        others["batch64_c2_fp64_tile16"] = measure(batch * 2, capi.EA_F64, 8, (capi.LOSS_CAUCHY, 1.0), tile=16)
        others["batch256_c2_fp32_tile16_beyond_infinity_cache"] = measure(batch * 8, capi.EA_F32, 4, (capi.LOSS_CAUCHY, 1.0), tile=16,
                                                                          traffic_key="batch256_c2_fp32_tile16")

    # Last, because it is the one measurement whose collective pattern (one ncclAllReduce per iteration enqueued on the
    # solve's stream) has only met a one-rank communicator so far: the point-sharded solve with the exchange issued by the
    # library (ea_solve_sharded_comm; look-ahead rule of ea_solve_sharded_device).  gloo rehearsal: the same protocol with
    # the exchange staged through torch.distributed.
    if not args.no_extras and not (world > 1 and args.no_device_allreduce):
        leg[0] = "point-sharded LM, all-reduce enqueued on the stream"
        try:
            from edge_alignment_amd import synth
            cfg2, dt2, _, loss2, _ = build_workload("lm1e5", rank)
            sl = ead.shard_slice(cfg2["xyz"].shape[0], rank, world)
            P3 = capi.Problem(*cfg2["K"], dtype=dt2, device=local_rank)
            P3.set_points(cfg2["xyz"][sl]); P3.set_dt_grid(cfg2["grid"]); P3.set_loss(*loss2)
            if comm is not None:
                solve4 = lambda: P3.solve_sharded_comm(q0, t0, comm)
                exchange = ("ea_solve_sharded_comm on a %d-rank communicator: per iteration ONE kernel launch (fold of the summed rows, "
                            "trust-region step and evaluation in every workgroup) + ONE in-place ncclAllReduce of the shard's partial rows "
                            "when every shard fits one workgroup per CU, else evaluation -> fold -> ncclAllReduce(32 doubles) -> step" % world)
            else:
                sums, enqueue = ead.make_device_allreduce(world, torch.device("cuda", local_rank), force_collective=multi)
                solve4 = lambda: P3.solve_sharded_device(q0, t0, enqueue, sums.data_ptr())
                exchange = "torch.distributed (%s) under the library's stream, through a Python callback" % args.dist_backend
            solve4()
            barrier_sync()
            ts = time.perf_counter()
            reps4, its4, lib4 = 10, 0, 0.0
            for _ in range(reps4):
                q4, t4, s4 = solve4()
                its4 += s4["num_iterations"]
                lib4 += s4["total_time_ms"]
            barrier_sync()
            el4 = time.perf_counter() - ts
            rows_form = None
            if comm is not None:
                try:
                    rows_form = bool(comm.info("row_solves") > 0)
                except Exception:
                    pass
            extras["lm_point_sharded_device_1e5_pts"] = {"iters_per_s": its4 / (lib4 * 1e-3), "iters_per_s_through_ctypes": its4 / el4,
                                                         "solve_ms": lib4 / reps4, "exchange": exchange, "rows_exchanged": rows_form,
                                                         "points_per_gpu": int(sl.stop - sl.start),
                                                         "pose_err_vs_planted": {"rad": synth.rotation_angle_between(q4, cfg2["q_true"]),
                                                                                 "m": float(np.linalg.norm(t4 - cfg2["t_true"]))}}
            P3.close()
        except Exception as e:
            extras["lm_point_sharded_device_1e5_pts"] = {"error": repr(e)}

    out = None
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            leg[0] = "cpu baseline"
            scale = float(os.environ.get("EA_BENCH_CPU_SAMPLE_SCALE", "1"))   # (tests/test_gpu_bench_line.py shortens the sample)
            extras["cpu_baseline"] = cpu_baseline(cfg, loss, budget_s=12.0 * scale)
            try:
                extras["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cfg, loss, budget_s=6.0 * scale)
            except Exception as e:  # a reported extra, never a reason to lose the line
                extras["cpu_baseline_all_cores"] = {"error": repr(e)}
        out = compose()
    leg[0] = "teardown"
    B.close()
    P.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        if node_barrier is not None:
            node_barrier.close()
        dist.destroy_process_group()
    watchdog.cancel()
    if rank == 0:
        line_out.write(json.dumps(out) + "\n")
        line_out.flush()


if __name__ == "__main__":
    main()
