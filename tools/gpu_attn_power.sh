#!/bin/bash
# samples sclk / power with rocm-smi while the attention kernel runs in a loop, for every variant library
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cat > /tmp/loop_attn.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mlx8_ws_audio_transformer_amd import ops
B, H, S = 64, 12, 1500
q, k, v = (torch.randn(B, H, S, 64, device="cuda") for _ in range(3))
q *= 0.125 * 1.4427
t0 = time.time(); it = 0
while time.time() - t0 < 8:
    for _ in range(50): ops.attention(q, k, v, "bf16x3")
    torch.cuda.synchronize(); it += 50
print("iters/s", it / (time.time() - t0))
PY
for v in mlx8-ws-audio-transformer_amd/variants/libawt_v*.so; do
  echo "== $v"
  AWT_LIB=$PWD/$v timeout -k 10 100 python3 /tmp/loop_attn.py &
  pid=$!
  sleep 5
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.7; done
  wait $pid
done
