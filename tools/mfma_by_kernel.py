#!/usr/bin/env python3
"""Per kernel NAME (not class) summary of the SQ / GRBM counter pass of tools/profile_round.sh: MFMA-busy fraction, effective clock, wave wait shares.
    python tools/mfma_by_kernel.py r04 prof_r04_fp32 prof_r04_fp16   ->  profiles/r04_mfma_by_kernel.json
Reads gpurun_out/<tag>/pmc_MFMA/**/counter_collection.csv (merged back by gpurun); CPU only."""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tags = sys.argv[1], sys.argv[2:]
PAT = re.compile(r"(gemm_pp_kernel<[^>]*>|gemm_f8s_kernel<[^>]*>|gemm_f8_kernel<[^>]*>|gemm_kernel<[^>]*>|attention_f16f8_pipe_kernel<[^>]*>|attention_kernel<[^>]*>|layernorm_kernel<[^>]*>)")
out = {}
for tag in tags:
    files = glob.glob(os.path.join(ROOT, "gpurun_out", tag, "pmc_MFMA", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):      # gpurun merges: older passes of the same tag may still lie beside the newest
        m = PAT.search(r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        if not m:
            continue
        k = m.group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    res = {}
    for k, c in acc.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        if not gui or not c.get("SQ_WAVE_CYCLES"):
            continue
        res[k] = {"launches": n[k], "avg_us": round(dur[k] / n[k] / 1e3, 1), "effective_clock_ghz": round(gui / 8 / dur[k], 3),
                  "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui / 8), 3),
                  "wave_wait_inst_frac": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3), "wave_wait_any_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)}
    out[tag.split(rnd + "_", 1)[-1]] = dict(sorted(res.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]))
json.dump({"note": "per kernel NAME (not class) from the SQ / GRBM pass of tools/profile_round.sh (gpurun_out/<tag>/pmc_MFMA); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                   "GRBM_GUI_ACTIVE / 8), effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; profiled passes run 2-3 % slower than un-profiled ones.  fp32 weights: "
                   "gemm_pp_kernel<3, true> = fc1 (GELU, split-line output), <1, false> = fc2 (fp32 + residual); gemm_f8s_kernel<4, false> = QKV, <1, false> = out_proj, <3, false> = conv1; "
                   "gemm_f8_kernel<5, CfgF8W4, true, false> = conv2 (three K segments, 32 x 32 form).  fp16 = fp16-exact weights: the one-cross-term 32 x 32 kernels (gemm_f8_kernel<EPI, CfgF8W4, false, true>)",
           "per_kernel": out}, open(os.path.join(ROOT, "profiles", "%s_mfma_by_kernel.json" % rnd), "w"), indent=1)
for tag, res in out.items():
    for k, v in res.items():
        print(tag, k, v)
