#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for m in 5120 20480 96000; do echo "M=$m"; timeout -k 10 200 python tools/gemm_bench.py $m 2>&1 | grep -E "bf16x3 +(fc1|fc2)|bf16 +(fc1|fc2)"; done
