#!/bin/bash
# HBM read / write counters of a secondary path, in passes of their own (run from the
# repository root on the GPU box):  bash tools/profile_traffic.sh <name>   where
# tools/<name>_bench.py is the program.  Outputs: gpurun_out/pmc_<name>_{fetch,write}/
set -e
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
NAME=$1
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rm -rf "$OUT/pmc_${NAME}_fetch" "$OUT/pmc_${NAME}_write"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_${NAME}_fetch" -o $NAME -- python3 "$ROOT/tools/${NAME}_bench.py" > "$OUT/pmc_${NAME}_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_${NAME}_write" -o $NAME -- python3 "$ROOT/tools/${NAME}_bench.py" > "$OUT/pmc_${NAME}_write.log" 2>&1
python3 - "$OUT" "$NAME" <<'PY'
import csv, sys, collections
out, name = sys.argv[1], sys.argv[2]
for cname, sub, mult in (("FETCH_SIZE", "fetch", 2048.0), ("WRITE_SIZE", "write", 1024.0)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/pmc_{name}_{sub}/{name}_counter_collection.csv")):
        if r["Counter_Name"] == cname:
            acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:4]:
        print(cname, "bytes/launch (corrected) %.4e" % (sum(v) / len(v) * mult), "launches", len(v), k)
PY
