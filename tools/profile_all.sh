#!/bin/bash
# Kernel-trace statistics of the secondary paths on the GPU box (run from the repo
# root): Baum-Welch E-step, MFCC front-end, Viterbi, emission, large model, posteriors.
# Outputs: gpurun_out/prof_<name>/<name>_kernel_stats.csv
set -e
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
for name in estep mfcc viterbi emission large posteriors endpoint; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -o "$name" -- python3 "$ROOT/tools/${name}_bench.py" > "$OUT/prof_$name.log" 2>&1
done
# the librosa-convention front-end (centred 2048-point frames): tools/mfcc_bench.py 2 1 librosa
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_mfcc_librosa" -o mfcc_librosa -- python3 "$ROOT/tools/mfcc_bench.py" 2 1 librosa > "$OUT/prof_mfcc_librosa.log" 2>&1
ls "$OUT"/prof_*/*_kernel_stats.csv
