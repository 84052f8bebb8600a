"""CPU: `python bench.py --gpus N` starts N ranks itself (no torchrun wrapper), the group really spans N ranks, and the
JSON line says so.  The `noop` workload walks the launcher / rendezvous / timing / JSON code of the real workloads
without touching the GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=e)


def test_gpus_2_spawns_two_gloo_ranks_and_reports_them():
    r = _run("--gpus", "2", "--dist-backend", "gloo", "--workload", "noop", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert j["ranks"]["world"] == 2 and j["ranks"]["collective_ranks"] == 2 and j["ranks"]["backend"] == "gloo"
    assert len(j["ranks"]["per_rank_clips_per_s"]) == 2


def test_single_rank_needs_no_process_group():
    r = _run("--workload", "noop")
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["ranks"]["backend"] == "none"


def test_more_gpus_than_visible_is_an_error_not_a_silent_single_gpu_run():
    r = _run("--gpus", "64", "--workload", "encode")
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr and not r.stdout.strip()


def test_world_size_mismatch_is_refused():
    r = _run("--gpus", "2", "--workload", "noop", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
