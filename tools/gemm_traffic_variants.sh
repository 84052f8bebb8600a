#!/bin/bash
# L2 hit / fabric traffic counters of one fc1-shaped GEMM for every variant library (separate PMC passes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cat > /tmp/one_gemm.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mlx8_ws_audio_transformer_amd import ops
m, n, k = 96000, int(os.environ.get("GEMM_N", 3072)), int(os.environ.get("GEMM_K", 768))
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * k ** -0.5
for _ in range(3): ops.linear(x, w, None, "bf16x3")
torch.cuda.synchronize()
PY
for v in mlx8-ws-audio-transformer_amd/variants/libawt_v*.so; do
  export AWT_LIB=$PWD/$v
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(basename $v .so)_$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/tv_$tag -- python3 /tmp/one_gemm.py > gpurun_out/tv_$tag.log 2>&1
    f=$(find gpurun_out/tv_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
if not f: print(tag, "no output"); sys.exit()
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemm_kernel" in r["Kernel_Name"] and "CfgW4" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, {k: "%.4g" % (sum(v[1:]) / max(len(v) - 1, 1)) for k, v in acc.items()})
PY
  done
done
