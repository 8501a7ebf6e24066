"""`python3 bench.py --gpus N` with no launcher in the environment starts its own N ranks (VERDICT r02 "what's missing" 4):
the parent makes no GPU call, sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* for N fresh processes, relays rank 0's one
line and exits non-zero when a rank fails.  Checked here on the CPU through the launcher's self-test mode (the ranks
rendezvous over gloo instead of touching a GPU); the GPU rehearsal of the real thing is scripts/rehearse_launcher.sh."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EA_BENCH_LAUNCHER_SELFTEST="1", **kw)
    return env


def test_plain_command_starts_n_ranks_and_relays_one_line():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "7"], capture_output=True, text=True, env=_env(),
                         cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["rank_sum"] == 6.0 and d["steps"] == 7
    assert d["master"].startswith("127.0.0.1:")


def test_a_failing_rank_fails_the_command():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True,
                         env=_env(EA_BENCH_LAUNCHER_SELFTEST_FAIL="1"), cwd=ROOT, timeout=300)
    assert out.returncode != 0
    assert "rank 1 exited with code 7" in out.stderr


def test_the_parent_does_not_import_torch_or_load_the_library():
    """the launcher branch is taken before anything that could initialise a device"""
    src = open(BENCH).read()
    head = src[:src.index("def main():")]
    main = src[src.index("def main():"):]
    branch = main.index("sys.exit(launch_ranks(args.gpus))")
    assert "import torch" not in main[:branch] and "capi" not in main[:branch]
    body = head[head.index("def launch_ranks(n):"):head.index("def launcher_selftest")]
    assert "import torch" not in body and "capi" not in body


def test_world_size_must_match_gpus():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True,
                         env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), cwd=ROOT, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE (4) != --gpus (2)" in out.stderr
