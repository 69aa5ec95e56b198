#!/bin/bash
# Profile the scoring bench on the GPU box: kernel trace statistics, then HBM
# read and write counters in passes of their own.  Run from the repo root:
#   bash tools/profile_score.sh
# Outputs land in gpurun_out/prof_*; tools/make_traffic.py turns them into
# profiles/ summaries.
set -e
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
# kernel times: the default bench command (without its CPU leg); counters: the scoring part alone
CMD="python3 $ROOT/bench.py --no-extra --no-train --no-cpu"
rm -rf "$OUT/prof_kt" "$OUT/prof_fetch" "$OUT/prof_write"
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_kt" -o score -- python3 $ROOT/bench.py --no-cpu > "$OUT/prof_kt.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -o score -- $CMD > "$OUT/prof_fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -o score -- $CMD > "$OUT/prof_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
