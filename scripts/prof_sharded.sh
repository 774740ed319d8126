#!/bin/bash
# kernel-trace profile of rank 0's share of the sharded large-map update in an 8-rank world (scripts/project_sharded.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_sh
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sh -- python3 $R/scripts/project_sharded.py --worlds ${1:-8} > $R/gpurun_out/prof_sh.log 2>&1
f=$(ls $R/gpurun_out/prof_sh/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/kernel_stats_sharded.csv
cut -d, -f1-4 $f | cut -c1-200 | head -30
