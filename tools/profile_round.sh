#!/bin/bash
# Round profile of the default bench workload: kernel-trace stats + HBM traffic counters (separate PMC passes).
# Usage (GPU box): bash tools/profile_round.sh <tag> <precision> <weights: fp32 | fp16> [encode | finetune]
# finetune: the fine-tune step (BASELINE configs[2], library-default backward): kernel-trace stats + the FETCH_SIZE / WRITE_SIZE passes only -> traffic.json
# with "backward_precision", which tools/collect_profiles.py merges into profiles/traffic_finetune.json (bench.py --workload finetune attaches it)
# (bench.py starts no helper process of any kind under a profiler: its power sampler is a thread and stays off when rocprofv3's preload is present)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=${1:-r03}
prec=${2:-f16f8}
wts=${3:-fp32}
wl=${4:-encode}
export AWT_PROFILE_WORKLOAD=$wl
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
if [ "$wl" = finetune ]; then
  common="--workload finetune --no-cpu-baseline --precision $prec --backward-precision bf16x3"
else
  common="--no-cpu-baseline --no-fast-mode --no-power --no-configs --precision $prec --weights $wts"
fi
args="bench.py --steps 3 --warmup 1 $common"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $args > $out/trace.log 2>&1 || { echo "kernel-trace pass failed"; exit 1; }
echo "pass trace done"
pmc_pass() {   # <dir> <counters...>: one --pmc pass of a 1-step bench; a failed pass ends the script (no further GPU step after a failure)
  local d=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$d -- python3 bench.py --steps 1 --warmup 1 $common > $out/$d.log 2>&1 || { echo "pass $d failed"; tail -3 $out/$d.log | cut -c1-300; exit 1; }
  echo "pass $d done"
}
pmc_pass pmc_FETCH_SIZE FETCH_SIZE
pmc_pass pmc_WRITE_SIZE WRITE_SIZE
if [ "$wl" != finetune ]; then
# MFMA utilisation and effective clock (own pass; SQ + GRBM slots)
pmc_pass pmc_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
# Fabric side of the L2 (own passes): average L2 -> fabric read latency, cycles stalled for DRAM credits, L2 hit rate.  rocprofv3 on gfx950 exposes no
# Infinity-Cache (MALL) or UMC counter, so "fetched from HBM" vs "served by the Infinity Cache" cannot be split from here: TCC_EA0_RDREQ_DRAM counts both.
# (at most two TCC counters per pass: five in one pass fail with "exceeds the capabilities of the hardware to collect")
pmc_pass pmc_FABRIC_lat TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
pmc_pass pmc_FABRIC_dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pmc_pass pmc_L2 TCC_HIT_sum TCC_MISS_sum
pmc_pass pmc_L2cyc TCC_CYCLE_sum
fi
python3 - $out $tag $prec $wts <<'PY3'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, "."); import bench
def short(n):
    for k, pats in (("gemm_kernel", ("gemm_kernel", "gemm_f8_kernel", "gemm_f8s_kernel", "gemm_pp_kernel")), ("attention_kernel", ("attention_kernel", "attention_f16f8")), ("layernorm_kernel", ("layernorm_kernel",))):
        if any(p in n for p in pats): return k
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ("pmc_FABRIC_lat", "pmc_FABRIC_dram", "pmc_L2", "pmc_L2cyc"):
    cc = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not cc: continue
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        k = short(r["Kernel_Name"])
        if not k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if d == "pmc_FABRIC_lat" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n[k] += 1
res = {}
for k, c in acc.items():
    rd, lvl, cyc = c.get("TCC_EA0_RDREQ_sum", 0.0), c.get("TCC_EA0_RDREQ_LEVEL_sum", 0.0), c.get("TCC_CYCLE_sum", 0.0)
    res[k] = {"launches_counted": n[k], "ea_read_requests_per_launch": rd / max(n[k], 1), "ea_read_requests_to_dram_frac": c.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / rd if rd else None,
              "avg_ea_read_latency_l2_cycles": lvl / rd if rd else None,
              "dram_credit_stall_frac_of_tcc_cycles": c.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", 0.0) / cyc if cyc else None,
              "l2_hit_rate": c.get("TCC_HIT_sum", 0.0) / (c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)) if (c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)) else None}
json.dump({"tag": tag, "build": bench.source_hash(), "precision": sys.argv[3], "weights": sys.argv[4],
           "note": "L2 (TCC) fabric interface; no MALL / UMC counters are exposed on gfx950, so HBM vs Infinity-Cache reads cannot be separated: RDREQ_DRAM counts requests to the memory side (both)",
           "per_kernel": res}, open(f"{out}/fabric.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY3
python3 - $out $tag $prec $wts <<'PY2'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, "."); import bench
def short(n):
    for k, pats in (("gemm_kernel", ("gemm_kernel", "gemm_f8_kernel", "gemm_f8s_kernel", "gemm_pp_kernel")), ("attention_kernel", ("attention_kernel", "attention_f16f8")), ("layernorm_kernel", ("layernorm_kernel",)),
                    ("logmel_stage1", ("logmel_stage1",)), ("logmel_finalize", ("logmel_finalize",)), ("im2col", ("im2col",))):
        if any(p in n for p in pats): return k
cc = glob.glob(f"{out}/pmc_MFMA/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"{out}/pmc_MFMA/**/*kernel_trace.csv", recursive=True)
if cc and kt:
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); t = collections.Counter(); seen = set()
    for r in csv.DictReader(open(cc[0])):
        k = short(r["Kernel_Name"])
        if not k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n[k] += 1; t[k] += dur.get(r["Dispatch_Id"], 0)
    res = {}
    for k, d in acc.items():
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
        res[k] = {"launches_counted": n[k], "avg_ns": t[k] / max(n[k], 1),
                  "effective_clock_ghz": cyc / max(t[k], 1),
                  "mfma_busy_frac": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc) if cyc else None,   # 1024 SIMDs
                  "wave_cycles_wait_any_frac": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"] if d["SQ_WAVE_CYCLES"] else None,
                  "wave_cycles_wait_inst_frac": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"] if d["SQ_WAVE_CYCLES"] else None,
                  "wave_cycles_active_frac": d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"] if d["SQ_WAVE_CYCLES"] else None}
    json.dump({"tag": tag, "build": bench.source_hash(), "precision": sys.argv[3], "weights": sys.argv[4], "note": "profiled pass (clocks read ~2-3 % lower than un-profiled); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
               "per_kernel": res}, open(f"{out}/mfma.json", "w"), indent=1)
    print(json.dumps(res, indent=1))
PY2
python3 - $out $tag $prec $wts <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, "."); import bench
def short(n):
    for k, pats in (("gemm_kernel", ("gemm_kernel", "gemm_f8_kernel", "gemm_f8s_kernel", "gemm_pp_kernel")), ("attention_kernel", ("attention_kernel", "attention_f16f8")), ("layernorm_kernel", ("layernorm_kernel",)),
                    ("logmel_stage1", ("logmel_stage1",)), ("logmel_finalize", ("logmel_finalize",)), ("im2col", ("im2col",))):
        if any(p in n for p in pats): return k
    return None
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    half = ids[len(ids) // 2] if ids else 0          # second half = the timed step (first half is the warm-up step)
    for r in rows:
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == c and int(r["Dispatch_Id"]) >= half:
            res[k][c].append(float(r["Counter_Value"]))
summary = {}
for k, d in res.items():
    n = max(len(v) for v in d.values())
    fetch_kb = sum(d.get("FETCH_SIZE", [])); write_kb = sum(d.get("WRITE_SIZE", []))
    # gfx950: FETCH_SIZE counts 64 B per 128 B request of a wide coalesced stream -> x2 (MI355X_MICROARCH.md, HBM section)
    summary[k] = {"launches_counted": n, "fetch_bytes_per_launch_corrected": fetch_kb * 1024 * 2 / max(n, 1),
                  "write_bytes_per_launch": write_kb * 1024 / max(n, 1),
                  "hbm_bytes_per_launch": (fetch_kb * 2 + write_kb) * 1024 / max(n, 1)}
import os
wl = os.environ.get("AWT_PROFILE_WORKLOAD", "encode")
json.dump({"tag": tag, "build": bench.source_hash(), "precision": sys.argv[3], "weights": sys.argv[4], "backward_precision": sys.argv[3] if wl == "finetune" else None,
           "workload": ("bench.py --workload finetune (Whisper-small + LoRA r=8, B=64, %s, library-default backward)" % sys.argv[3]) if wl == "finetune" else "bench.py default (Whisper-small, parity, B=64, %s, %s weights)" % (sys.argv[3], sys.argv[4]), "per_kernel": summary,
           "note": "FETCH_SIZE x2 per the gfx950 correction; counters from separate --pmc passes"}, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
head -12 $out/kernel_stats.csv | cut -c1-160
