"""bench.py's contract, checked on the line the driver's command prints: `python bench.py --gpus 1 --steps 20 --warmup 5`
(here with the CPU sample shortened to a tenth; everything else as the driver runs it).  One JSON line on stdout; metric, unit and config of
BASELINE.json's C2; value = points x steps / time of the timed region; the roofline object's fraction follows from its
achieved and peak figures and its achieved figure from the algorithmic bytes and the kernel's measured duration; the CPU
baseline is the oracle's (kind "port") on a stated sample."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_command_prints_one_valid_line(hip):
    env = dict(os.environ, EA_BENCH_CPU_SAMPLE_SCALE="0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == "edge-point residual+Jacobian evals/sec" and d["unit"] == "evals/s"
    assert d["metric"].split()[0] in json.dumps(base)            # (the metric BASELINE.json names)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    cfg = d["config"]
    assert cfg["workload"].startswith("c2") and "model" not in cfg and cfg["points_per_gpu"] == 50000
    # value is whole-job throughput of the timed region
    assert d["value"] > 1e9
    assert abs(d["value"] - cfg["points_per_gpu"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    # the dominant kernel is the evaluation launch of G poses: its bytes are G x one evaluation's
    assert "ea_eval_poses_kernel" in r["kernel"] and r["evaluation_launches_in_timed_region"] == 1 and r["poses_per_launch"] == 20
    assert abs(r["algorithmic_bytes_per_launch"] - 20 * r["algorithmic_bytes_per_evaluation"]) < 1 and r["algorithmic_bytes_per_evaluation"] == 3657600
    assert 2e-3 < r["kernel_ms"] < 0.2
    assert r["traffic"] is None or r["traffic"] / r["algorithmic_bytes_per_launch"] < 2.0
    assert r["secondary"]["bound"].startswith("valu")
    assert 5e-4 < r["launch_floor_ms"] < r["kernel_ms"]
    assert abs(r["frac_ceiling_at_floor"] - min(1.0, r["algorithmic_bytes_per_launch"] / (r["launch_floor_ms"] * 1e-3) / 1e9 / r["peak"])) <= 1e-9
    assert r["frac"] < r["frac_ceiling_at_floor"] <= 1.0
    # the same kernel at one pose per launch (an LM iteration's launch) and the dependent two-launch step, beside the headline
    o = r["one_pose_per_launch"]
    assert 1e-3 < o["kernel_ms_back_to_back"] < 1e-2 and 5e-4 < o["launch_floor_ms"] < o["kernel_ms_back_to_back"]
    assert o["frac"] < o["frac_ceiling_at_floor"] < 1.0
    assert r["step_ms_events_serial_dependent"] > r["step_ms_events"]
    assert 0 < d["value_serial_dependent_steps"] < d["value"]
    assert "ea_batch_eval_resident_poses" in cfg["timed_region"] and "different poses" in cfg["timed_region"]
    assert d["single_eval_call_ms"] > d["ms_per_step"] and d["eval_poses_call_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "evals/s" and c["cores"] == 1 and c["value"] > 1e5 and "passes over" in c["sample"]
    m = d["materialised_mode"]
    assert "ea_eval_rows_kernel" in m["kernel"] and 0.0 < m["frac"] < 1.0
    assert d["lm_iters_per_s_at_1e5_pts"] > 1e4 and d.get("incomplete") is None
    assert set(d["other_workloads"]) >= {"c5_fp32_1e6pts_2048x1536", "batch32_c2_fp64", "batch32_c2_fp32"}
