#!/bin/bash
# kernel-trace profile of the batched bench (no side measurements); summary csv under gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --no-extra > $R/gpurun_out/prof_bench.log 2>&1
f=$(ls $R/gpurun_out/prof/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/kernel_stats.csv
cut -c1-200 $f | head -30
