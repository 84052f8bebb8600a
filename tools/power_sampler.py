#!/usr/bin/env python3
"""rocm-smi sampler used by bench.py: appends {"t", "package_w", "sclk_mhz"} JSON lines to <out> until <stop> exists or the parent is gone.
Started by bench.py BEFORE it initialises the GPU (a GPU-initialised process must not exec other programs on this pool)."""
import json, os, re, subprocess, sys, time

out, stop, ppid = sys.argv[1], sys.argv[2], int(sys.argv[3])
t_end = time.time() + 900
with open(out, "a") as f:
    while time.time() < t_end and not os.path.exists(stop) and os.path.exists("/proc/%d" % ppid):
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            w = [float(x) for x in re.findall(r"Power \(W\):\s*([0-9.]+)", txt)]
            c = [int(x) for x in re.findall(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", txt)]
            if w and c:
                f.write(json.dumps({"t": time.time(), "package_w": max(w), "sclk_mhz": min(c)}) + "\n")
                f.flush()
        except Exception:
            pass
        time.sleep(0.2)
