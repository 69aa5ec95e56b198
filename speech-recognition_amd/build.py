"""Build libsrx.so (HIP, gfx950 only) in-tree with hipcc.

Usage: ``python speech-recognition_amd/build.py [--force]``.  hipcc
cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrx.so")
SOURCES = ["api.hip", "hmm_generic.hip", "score_mfma.hip", "score_split8.hip", "score_split4.hip", "score_split16.hip", "score_f64.hip", "estep_fused.hip", "viterbi_fused.hip", "mfcc.hip", "mfcc_fast.hip", "mfcc_wide.hip", "large_model.hip", "endpoint.hip", "pipeline.hip", "estep_wave.hip"]
# Packed f32 vector instructions (what SLP vectorisation makes of adjacent adds and
# multiplies) issue slower than the two plain ones beside matrix instructions.
FILE_FLAGS = {"score_mfma.hip": ["-fno-slp-vectorize"], "score_split8.hip": ["-fno-slp-vectorize"],
              "score_split4.hip": ["-fno-slp-vectorize"], "score_split16.hip": ["-fno-slp-vectorize"],
              "viterbi_fused.hip": ["-fno-slp-vectorize"],
              # (the 512-point transform is written on plain registers: re-packed by the
              # vectoriser it needs a move per pair and runs 7 % slower)
              "mfcc_fast.hip": ["-fno-slp-vectorize"],
              "mfcc_wide.hip": ["-fno-slp-vectorize"]}
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall",
         "-Wno-unused-function", "-fvisibility=default",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    extra = os.environ.get("SRX_CXXFLAGS", "").split()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "srx.h"))
    objs = []
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    procs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [_hipcc()] + FLAGS + FILE_FLAGS.get(s, []) + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
