#!/bin/bash
# samples sclk / power with rocm-smi while one GEMM shape runs in a loop, for every variant library
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cat > /tmp/loop_gemm.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mlx8_ws_audio_transformer_amd import ops
m, n, k = 96000, 3072, 768
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * k ** -0.5
t0 = time.time(); it = 0
while time.time() - t0 < 8:
    for _ in range(200): ops.linear(x, w, None, "bf16x3")
    torch.cuda.synchronize(); it += 200
print("iters/s", it / (time.time() - t0))
PY
for v in mlx8-ws-audio-transformer_amd/variants/libawt_v*.so; do
  echo "== $v"
  AWT_LIB=$PWD/$v timeout -k 10 100 python3 /tmp/loop_gemm.py &
  pid=$!
  sleep 5
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '; echo; sleep 0.7; done
  wait $pid
done
