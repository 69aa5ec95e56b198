#!/bin/bash
# Kernel times of a fit at the reference's own size: tools/profile_small.sh  (run from the repository root on the GPU box)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/small_kt; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/small_kt -- python $ROOT/tools/small_fit_once.py > /dev/null 2>&1
f=$(find $ROOT/gpurun_out/small_kt -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"].split("(")[0][-60:], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2))
PY
