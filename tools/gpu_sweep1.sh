#!/bin/bash
# chunk-size sweep + first rocprof kernel trace (round 1)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for c in 8 32 64; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --chunk $c --no-cpu-baseline --no-fast-mode > gpurun_out/bench_chunk$c.log 2>&1
  tail -1 gpurun_out/bench_chunk$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk', $c, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['time_share_ms_per_step'])"
done
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode > gpurun_out/prof_r1.log 2>&1
find gpurun_out/prof_r1 -name "*stats*" | head
