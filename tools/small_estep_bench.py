"""E-step time of small jobs (utterances of 60-240 frames, one of 6 models each) on the float32
paths (auto = one lane per utterance up to 8 states and 8192 utterances, fused matrix-core
kernel otherwise), the shape-generic float32 kernels and float64.
    python3 tools/small_estep_bench.py [f32]     (f32: the float32 default path only, for profiling)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_recognition_amd as sra
ctx = sra.Context.default(0)
rng = np.random.default_rng(1)
only = len(sys.argv) > 1 and sys.argv[1] == "f32"
shapes = ((6, 3, 26, 150), (6, 8, 39, 150), (6, 16, 39, 150), (6, 3, 26, 4096), (6, 8, 39, 16384))
if only and len(sys.argv) > 2:
    shapes = [shapes[int(sys.argv[2])]]
if os.environ.get("SHAPES"):                     # e.g. SHAPES="6,8,39,2048;6,3,26,16384"
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
paths = os.environ.get("PATHS")                  # e.g. PATHS="fused,lane": float32 on these paths only
for (M, S, D, U) in shapes:
    lengths = rng.integers(60, 240, size=U)
    X = rng.standard_normal((int(lengths.sum()), D))
    um = ctx.tensor((np.arange(U) % M).astype(np.int32))
    bank = sra.HMMBank(ctx, rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
                       rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))
    variants = (("float32", "auto", False), ("float32", "fused", False), ("float32", "lane", False),
                ("float32", "auto", True), ("float64", "auto", False), ("float64", "fused", False),
                ("float64", "lane", False))
    if paths:
        variants = tuple(("float32", q, False) for q in paths.split(","))
    for dtype, path, generic in variants[:1] if only else variants:
        data = sra.Ragged(ctx, X, lengths, dtype=dtype)
        ctx.set_estep_path(path)
        out = None
        for _ in range(20):
            out = bank.estep(data, um, out=out, generic=generic)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 100
        for _ in range(n):
            out = bank.estep(data, um, out=out, generic=generic)
        torch.cuda.synchronize()
        ctx.set_estep_path("auto")
        print(f"M{M} S{S} D{D} U{U} {dtype} path={path} generic={generic}: {(time.perf_counter()-t0)/n*1e6:.0f} us per E-step")
