#!/bin/bash
# Per-kernel times of one program (run from the repository root on the GPU box):
#   bash tools/profile_kt.sh <tag> <program.py> [args]  -> gpurun_out/kt_<tag>.csv (rocprofv3 kernel stats)
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
TAG=$1; shift
OUT="$ROOT/gpurun_out"
cd /tmp; export TMPDIR=/tmp
rm -rf "$OUT/kt_${TAG}"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_${TAG}" -o p -- python3 "$ROOT/$1" "${@:2}" > "$OUT/kt_${TAG}.log" 2>&1 || exit 1
f=$(find "$OUT/kt_${TAG}" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/kt_${TAG}.csv"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f'{r["Name"].split("(")[0][-48:]:48s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.2f} min {float(r["MinNs"])/1e3:9.2f} max {float(r["MaxNs"])/1e3:9.2f}')
PY
