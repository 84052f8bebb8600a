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


def test_launcher_parent_imports_no_torch_and_counts_gpus_from_sysfs():
    """the process that starts the ranks must never initialise the GPU.  `bench.visible_gpus()` reads sysfs only;
    spawning N ranks does not pull torch (or anything HIP) into the parent."""
    code = ("import sys, bench\n"
            "n = bench.visible_gpus()\n"
            "assert isinstance(n, int) and n >= 0\n"
            "assert 'torch' not in sys.modules and 'ctypes' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)\n"
            "import inspect\n"
            "assert 'import torch' not in inspect.getsource(bench.spawn_ranks)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]


def test_a_failing_rank_ends_the_others_quickly(tmp_path):
    """the launcher polls every rank; one rank's failure kills the rest instead of waiting for rank 0's collective to time out."""
    import time
    script = tmp_path / "fake_bench.py"
    script.write_text("import os, sys, time\n"
                      "sys.path.insert(0, %r)\n"
                      "import bench\n"
                      "if 'WORLD_SIZE' in os.environ:\n"
                      "    if os.environ['RANK'] == '1':\n"
                      "        sys.exit(7)\n"
                      "    time.sleep(600)\n"
                      "class A: gpus = 2; dist_backend = 'gloo'; workload = 'noop'\n"
                      "bench.__file__ = __file__\n"
                      "sys.exit(bench.spawn_ranks(A()))\n" % ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 7 and time.time() - t0 < 60, (r.returncode, r.stderr[-500:])


def test_power_is_read_from_sysfs_without_any_child_process(tmp_path):
    """package power / shader clock come from hwmon + pp_dpm_sclk files, read by a thread: no rocm-smi, no exec; and no
    sampler at all under a profiler preload."""
    sys.path.insert(0, ROOT)
    import bench
    dev = tmp_path / "card1" / "device"
    (dev / "hwmon" / "hwmon3").mkdir(parents=True)
    (dev / "hwmon" / "hwmon3" / "power1_average").write_text("1364000000\n")
    (dev / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 1917Mhz *\n2: 2400Mhz\n")
    idle = tmp_path / "card0" / "device"
    (idle / "hwmon" / "hwmon1").mkdir(parents=True)
    (idle / "hwmon" / "hwmon1" / "power1_average").write_text("95000000\n")
    (idle / "pp_dpm_sclk").write_text("0: 132Mhz *\n")
    assert bench.read_power_sysfs(str(tmp_path))[:2] == (1364.0, 1917) and "busiest" in bench.read_power_sysfs(str(tmp_path))[2]
    assert bench.read_power_sysfs(str(tmp_path / "nothing")) is None
    # with the rank's PCI address the sample comes from THAT card (sysfs names a card's device directory by its address), not from the busiest one
    pci_root = tmp_path / "pci"
    for card, bdf, uw, clk in (("card0", "0000:05:00.0", "95000000", "0: 132Mhz *\n"), ("card1", "0000:85:00.0", "1364000000", "0: 132Mhz\n1: 1917Mhz *\n")):
        real = pci_root / "devices" / bdf
        (real / "hwmon" / "hwmon0").mkdir(parents=True)
        (real / "hwmon" / "hwmon0" / "power1_average").write_text(uw + "\n")
        (real / "pp_dpm_sclk").write_text(clk)
        (pci_root / "drm" / card).mkdir(parents=True)
        os.symlink(real, pci_root / "drm" / card / "device")
    assert bench.read_power_sysfs(str(pci_root / "drm"), pci="0000:05:00.0") == (95.0, 132, "device 0000:05:00.0")
    assert bench.read_power_sysfs(str(pci_root / "drm"), pci="0000:85:00.0")[:2] == (1364.0, 1917)
    assert bench.read_power_sysfs(str(pci_root / "drm"), pci="0000:99:00.0")[:2] == (1364.0, 1917)        # unknown address: the busiest card, and `which` says so
    import inspect
    src = inspect.getsource(bench.PowerSampler) + inspect.getsource(bench.read_power_sysfs)
    assert "subprocess" not in src and "Popen" not in src and "os.exec" not in src and "os.system" not in src
    s = bench.PowerSampler(root=str(tmp_path), period=0.01)
    import time
    time.sleep(0.1)
    s.close()
    assert s.thread is not None and len(s.rows) >= 2 and s.rows[0][1:3] == (1364.0, 1917)
    old = os.environ.get("ROCP_TOOL_LIBRARIES")
    os.environ["ROCP_TOOL_LIBRARIES"] = "/opt/rocm/lib/librocprofiler-sdk-tool.so"
    try:
        assert bench.profiler_attached() and bench.PowerSampler(root=str(tmp_path)).thread is None
    finally:
        if old is None:
            os.environ.pop("ROCP_TOOL_LIBRARIES")
        else:
            os.environ["ROCP_TOOL_LIBRARIES"] = old
