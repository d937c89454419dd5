"""bench.py called without a launcher and with --gpus N > 1 (or --spawn) starts the ranks itself, as child
processes, before it touches a GPU -- the way the scaling run calls it.  Driven here on the CPU with a stub worker in
place of the per-rank part (which needs GPUs): world size 2, gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_BENCH_WORKER"] = os.path.join(ROOT, "tests", "bench_stub_worker.py")
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=300)


def test_launcher_starts_two_ranks_and_relays_one_line():
    p = run({}, "--gpus", "2", "--steps", "7", "--warmup", "1")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().splitlines()
    assert len(lines) == 1, lines  # stdout carries the result line and nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] == 3.0  # both ranks took part in the reduction
    assert out["argv"] == ["--gpus", "2", "--steps", "7", "--warmup", "1"]
    assert b"says hello" in p.stderr


def test_spawn_flag_runs_one_rank_as_a_child_and_is_not_passed_on():
    p = run({}, "--gpus", "1", "--spawn")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    out = json.loads(p.stdout.decode().strip())
    assert out["n_gpus"] == 1 and out["argv"] == ["--gpus", "1"]


def test_a_failing_rank_fails_the_launcher():
    p = run({"RT_STUB_FAIL_RANK": "1"}, "--gpus", "2")
    assert p.returncode != 0
    assert p.stdout.decode().strip() == ""
