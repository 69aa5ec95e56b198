#!/bin/bash
# A variant of libsrx.so with ONE source rebuilt with extra flags (measurements, diagnostics):
#   tools/build_variant.sh <name> <source.hip> [flags...]   ->  build_variants/libsrx_<name>.so
# Load it with SRX_LIB=build_variants/libsrx_<name>.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
CS=$ROOT/speech-recognition_amd/csrc
mkdir -p $ROOT/build_variants
OBJ=$ROOT/build_variants/${NAME}_$(basename ${SRC%.hip}).o
EXTRA=""
case $(basename $SRC) in score_*|viterbi_fused.hip|mfcc_fast.hip) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -fvisibility=default \
    -I $ROOT/include -I $CS $EXTRA "$@" -c $CS/$(basename $SRC) -o $OBJ
OTHERS=$(ls $CS/*.o | grep -v "/$(basename ${SRC%.hip}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/build_variants/libsrx_${NAME}.so $OBJ $OTHERS
echo $ROOT/build_variants/libsrx_${NAME}.so
