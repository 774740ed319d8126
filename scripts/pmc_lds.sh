#!/bin/bash
# LDS bank conflicts and MFMA busy cycles of the two per-filter kernels (standalone benches)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for k in joseph solver; do
  rm -rf $R/gpurun_out/pmc_$k
  timeout -s KILL 60 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$k -- $R/scripts/${k}_bench_FULL > $R/gpurun_out/pmc_$k.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$R/gpurun_out/pmc_$k/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for kn,v in acc.items():
    print(kn, dict(v))
PY
done
