#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cat > /tmp/one_gemm.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mlx8_ws_audio_transformer_amd import ops
m, n, k = 96000, 3072, 768
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * k ** -0.5
for _ in range(20): ops.linear(x, w, None, "bf16x3")
torch.cuda.synchronize()
PY
for v in $(seq 0 $(( $(ls mlx8-ws-audio-transformer_amd/variants/libawt_v*.so | wc -l) - 1 ))); do
  export AWT_LIB=$PWD/mlx8-ws-audio-transformer_amd/variants/libawt_v$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/clk_$v -- python3 /tmp/one_gemm.py > gpurun_out/clk_$v.log 2>&1
  python3 - gpurun_out/clk_$v $v <<'PY'
import csv, sys, glob, collections
d, v = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if "gemm_kernel" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "gemm_kernel" in r["Kernel_Name"]: acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(acc, key=int)[5:]
clk = [acc[i]["GRBM_GUI_ACTIVE"] / 8 / dur[i] for i in ids if i in dur]
busy = [acc[i]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * acc[i]["GRBM_GUI_ACTIVE"] / 8) for i in ids if i in dur]
print("variant", v, "avg ns", sum(dur[i] for i in ids) / len(ids), "clock GHz %.3f" % (sum(clk) / len(clk)), "mfma busy %.3f" % (sum(busy) / len(busy)))
PY
done
