#!/usr/bin/env python3
"""Copies the summaries of tools/profile_round.sh runs (gpurun_out/prof_<tag>/) into profiles/ under round-tagged names and merges their
traffic summaries into profiles/traffic.json ({"entries": [...]}: one entry per (build, precision, weights); bench.py attaches the entry that
matches the running library).    python tools/collect_profiles.py r03 prof_r03_fp32 prof_r03_fp16"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tags = sys.argv[1], sys.argv[2:]
entries = []
for tag in tags:
    src = os.path.join(ROOT, "gpurun_out", tag)
    suffix = tag.split(rnd + "_", 1)[-1] if rnd + "_" in tag else tag
    for name in ("kernel_stats.csv", "mfma.json", "traffic.json", "fabric.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            base, ext = os.path.splitext(name)
            shutil.copy(p, os.path.join(ROOT, "profiles", "%s_%s_%s%s" % (rnd, base, suffix, ext)))
    t = os.path.join(src, "traffic.json")
    if os.path.exists(t):
        e = json.load(open(t))
        if e.get("backward_precision"):          # a fine-tune-step profile: its own file, read by bench.py --workload finetune
            json.dump({"note": "HBM / fabric traffic per launch of the fine-tune step (tools/profile_round.sh <tag> <prec> <wts> finetune); FETCH_SIZE x2 (gfx950)",
                       "entries": [e]}, open(os.path.join(ROOT, "profiles", "traffic_finetune.json"), "w"), indent=1)
            print("profiles/traffic_finetune.json:", e["build"], e["backward_precision"])
        else:
            entries.append(e)
if entries:
    json.dump({"note": "HBM / fabric traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh); FETCH_SIZE x2 per the gfx950 "
                       "correction of MI355X_MICROARCH.md; bench.py attaches the entry whose build / precision / weights match the running library",
               "entries": entries}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print("profiles/traffic.json:", [(e["build"], e["precision"], e.get("weights")) for e in entries])
