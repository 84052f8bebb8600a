#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for shape in ${SHAPES:-16 32}; do
  timeout -k 10 60 tools/_bin/mfma_power $shape 8 &
  pid=$!
  sleep 4
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.7; done
  wait $pid
done
