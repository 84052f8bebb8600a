#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in mlx8-ws-audio-transformer_amd/variants/libawt_v*.so; do
  echo "== $v"
  AWT_LIB=$PWD/$v timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep "attention B="
done
