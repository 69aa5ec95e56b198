#!/bin/bash
# Kernel times of the front-end: tools/profile_mfcc.sh [tag]  (run from the repository root on the GPU box)
set -e
ROOT=$(pwd)
TAG=${1:-mfcc}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/$TAG; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG -- python $ROOT/tools/mfcc_bench.py $2 $3 $4 > /dev/null 2>&1
f=$(find $ROOT/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "cepstra" in n or "finish" in n:
        print(n.split("(")[0][-40:], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
PY
