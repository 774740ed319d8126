#!/bin/bash
# HBM traffic of the batched bench per kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# (each pass: counters only + kernel trace), outputs under gpurun_out/pmc_{fetch,write}.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extra > $R/gpurun_out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extra > $R/gpurun_out/pmc_write.log 2>&1 || exit 1
python3 $R/scripts/pmc_summarize.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_traffic.json
