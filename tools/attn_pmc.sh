#!/bin/bash
# PMC counters of the attention kernels at the bench shape (GPU box): where the wave cycles go, per precision / workgroup shape.
# Usage: bash tools/attn_pmc.sh <outdir-tag>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=${1:-attn}
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
cat > /tmp/one_attn.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mlx8_ws_audio_transformer_amd import _lib, ops
prec, shape = sys.argv[1], int(sys.argv[2])
B, H, S = 64, 12, 1500
q, k, v = (torch.randn(B, H, S, 64, device="cuda") for _ in range(3)); q *= 0.35
_lib.tuning_set("attn_shape", shape)
for _ in range(6): ops.attention(q, k, v, prec)
torch.cuda.synchronize()
PY
IFS=";" read -ra CFGS <<< "${AWT_PMC_CFGS:-bf16x3 0;f16f8 1;f16f8 2;f16f8 3}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  for pass in A B; do
    if [ $pass = A ]; then ctr="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
    else ctr="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$1_$2_$pass -- python3 /tmp/one_attn.py $1 $2 > $out/$1_$2_$pass.log 2>&1
  done
done
python3 - $out <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*_A")) :
    base = d[:-2]
    row = {}
    for p in ("A", "B"):
        cc = glob.glob(base + "_" + p + "/**/*counter_collection.csv", recursive=True)
        kt = glob.glob(base + "_" + p + "/**/*kernel_trace.csv", recursive=True)
        if not cc or not kt: continue
        dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "attention" in r["Kernel_Name"]}
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(cc[0])):
            if "attention" in r["Kernel_Name"]: acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        ids = sorted(acc, key=int)[2:]
        for i in ids:
            for k, v in acc[i].items(): row[k] = row.get(k, 0.0) + v / len(ids)
        row["ns_" + p] = sum(dur[i] for i in ids) / len(ids)
    if not row: continue
    wc = row.get("SQ_WAVE_CYCLES", 1)
    cyc = row.get("GRBM_GUI_ACTIVE", 0) / 8
    print(os.path.basename(base), "ns %.0f" % row.get("ns_A", 0), "clock %.2f GHz" % (cyc / max(row.get("ns_A", 1), 1)),
          "| wave-cycle shares: wait_any %.2f wait_inst %.2f active %.2f (valu %.2f)" % (row.get("SQ_WAIT_ANY", 0) / wc, row.get("SQ_WAIT_INST_ANY", 0) / wc, row.get("SQ_ACTIVE_INST_ANY", 0) / wc, row.get("SQ_ACTIVE_INST_VALU", 0) / wc),
          "| mfma busy %.2f" % (row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0),
          "| insts valu %.3g mfma %.3g lds %.3g salu %.3g | lds conflict/active %.2f wait_inst_lds/wave %.2f" % (
              row.get("SQ_INSTS_VALU", 0), row.get("SQ_INSTS_MFMA", 0), row.get("SQ_INSTS_LDS", 0), row.get("SQ_INSTS_SALU", 0),
              row.get("SQ_LDS_BANK_CONFLICT", 0) / max(row.get("SQ_LDS_IDX_ACTIVE", 1), 1), row.get("SQ_WAIT_INST_LDS", 0) / wc))
PY
