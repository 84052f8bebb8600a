#!/bin/bash
# FETCH_SIZE (L2 -> fabric reads, KB) and time of each encoder GEMM shape for tile-order settings (AWT_GEMM_GROUP_N)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cat > /tmp/one_gemm.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mlx8_ws_audio_transformer_amd import ops
m, n, k = 96000, int(os.environ["GEMM_N"]), int(os.environ["GEMM_K"])
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * k ** -0.5
for _ in range(4): ops.linear(x, w, None, "bf16x3")
torch.cuda.synchronize()
PY
for shape in "3072 768" "2304 768" "768 768" "768 3072"; do
  set -- $shape; export GEMM_N=$1 GEMM_K=$2
  for grp in ${GROUPS_LIST:-auto 0}; do
    if [ $grp = auto ]; then unset AWT_GEMM_GROUP_N; else export AWT_GEMM_GROUP_N=$grp; fi
    tag=N${GEMM_N}_K${GEMM_K}_g$grp
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ts_$tag -- python3 /tmp/one_gemm.py > gpurun_out/ts_$tag.log 2>&1
    python3 - gpurun_out/ts_$tag $tag <<'PY'
import csv, sys, glob
d, tag = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True); kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not cc: print(tag, "no output"); sys.exit()
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
v = [(float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 0)) for r in csv.DictReader(open(cc[0])) if "gemm_kernel" in r["Kernel_Name"] and "CfgW4" in r["Kernel_Name"]][1:]
print(tag, "fetch GB (x2 corrected) %.3f" % (sum(a for a, _ in v) / len(v) * 2 * 1024 / 1e9), "avg us %.1f" % (sum(b for _, b in v) / len(v) / 1e3))
PY
  done
done
