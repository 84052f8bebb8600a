#!/bin/bash
# Kernel-trace statistics of the fine-tune step (GPU box): which kernels the step runs and for how long.  bash tools/profile_finetune.sh <tag>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=${1:-r02}
out=gpurun_out/prof_ft_$tag; rm -rf $out; mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload finetune --steps 3 --warmup 1 > $out/trace.log 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
head -40 $out/kernel_stats.csv | cut -c1-170
tail -2 $out/trace.log | cut -c1-600
