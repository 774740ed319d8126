#!/bin/bash
# kernel-trace profile of one large-map filter; per-kernel summary csv under gpurun_out/prof_lm
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_lm
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lm -- python3 $R/scripts/large_map_time.py > $R/gpurun_out/prof_lm.log 2>&1
f=$(ls $R/gpurun_out/prof_lm/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/kernel_stats_large_map.csv
cut -d, -f1-4 $f | cut -c1-220 | head -24
