#!/bin/bash
# Shader-core counters of one program's kernels, passes of at most 8 counters each (run from the
# repository root on the GPU box):  bash tools/profile_sq.sh <tag> <program.py> [args]
# Output: gpurun_out/sq_<tag>.txt (per kernel, averages per launch)
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
TAG=$1; shift
OUT="$ROOT/gpurun_out"
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAVES"
rm -rf "$OUT/sq_${TAG}_1" "$OUT/sq_${TAG}_2"
timeout -s KILL 150 rocprofv3 --pmc $P1 --output-format csv -d "$OUT/sq_${TAG}_1" -o p -- python3 "$ROOT/$1" "${@:2}" > "$OUT/sq_${TAG}_1.log" 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc $P2 --output-format csv -d "$OUT/sq_${TAG}_2" -o p -- python3 "$ROOT/$1" "${@:2}" > "$OUT/sq_${TAG}_2.log" 2>&1 || exit 1
# (a pass of its own: cycles spent issuing vector-memory instructions)
rm -rf "$OUT/sq_${TAG}_3"
timeout -s KILL 150 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d "$OUT/sq_${TAG}_3" -o p -- python3 "$ROOT/$1" "${@:2}" > "$OUT/sq_${TAG}_3.log" 2>&1 || echo "third pass not collected"
python3 - "$OUT" "$TAG" <<'PY' > "$OUT/sq_${TAG}.txt"
import csv, sys, collections, glob
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("1", "2", "3"):
    for f in glob.glob(f"{out}/sq_{tag}_{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    if "SQ_WAVE_CYCLES" not in d: continue
    print(k, "launches", len(d["SQ_WAVE_CYCLES"]))
    for c, v in sorted(d.items()):
        print(f"    {c:26s} {sum(v)/len(v):.4e}")
PY
cat "$OUT/sq_${TAG}.txt" | head -60
