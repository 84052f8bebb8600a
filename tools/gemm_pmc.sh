#!/bin/bash
# PMC counters for the GEMM kernel (separate passes; never combined with trace domains other than kernel-trace)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cat > /tmp/one_gemm.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mlx8_ws_audio_transformer_amd import ops
m, n, k = 96000, 3072, 768
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * k ** -0.5
for _ in range(2): ops.linear(x, w, None, "bf16x3")
torch.cuda.synchronize()
PY
rocprofv3 -L > gpurun_out/pmc_list.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$tag -- python3 /tmp/one_gemm.py > gpurun_out/pmc_$tag.log 2>&1
  f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
  echo "== $set -> $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f: sys.exit()
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemm_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("  ", k, "per-dispatch avg", sum(v)/len(v), "n", len(v))
PY
done
