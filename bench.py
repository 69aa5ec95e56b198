#!/usr/bin/env python
"""Headline benchmark: batched HMM scoring, 65,536 synthetic utterances x 200
frames x 39-dim features against 6 word models of 8 states, on N GPUs of one node
(utterances sharded, no collective), and Baum-Welch iterations/s on the config's
1,048,576 utterances (6 models x 16 states, one all-reduce of the statistics per
iteration).

    python bench.py --gpus N --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no RANK in the environment this process only launches N
children (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), before any
GPU call of its own, waits for them and exits with their status; under
torch.distributed.run the ranks are already there and nothing is launched.

A step is one pass of srx_forward_score over the rank's batch, inputs resident
in HBM.  Rank 0 prints ONE JSON line.  Beside the headline it carries
  roofline      algorithmic HBM bytes (156 B/frame in + 48 B/utterance out)
                / event-timed duration of the scoring launch, against 8 TB/s
  cpu_baseline  the loop-faithful fp64 CPU oracle on a bounded sample (N = 1)
  baum_welch_iter_per_s, allreduce_ms_per_iter, rccl_ranks
                the training metric (fixed total work: strong scaling), the time of
                the statistics all-reduce alone, and the rank count an actual
                all-reduce over the process group returned
  extra         (N = 1) front-end frames/s (16 kHz, 512-pt FFT, 40 mel, 39-dim),
                emission, Viterbi, posteriors, end-pointing and the large-model
                lines, each with its own size, measured outside the timed region
                of the headline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec
N_UTTS, T_FRAMES, N_FEAT = 65536, 200, 39
N_MODELS, N_STATES = 6, 8


def synth_models(rng, M, S, D):
    return (rng.dirichlet(np.ones(S), size=M), rng.dirichlet(np.ones(S), size=(M, S)),
            rng.standard_normal((M, S, D)), rng.uniform(0.5, 1.5, (M, S, D)))


def cpu_baseline(sample_utts, seed=1234):
    """Loop-faithful CPU oracle (same per-frame / per-state Python loops as
    my_hmmlearn/utils.py:132-150) on `sample_utts` utterances, one core."""
    from oracle import hmm_oracle as ho
    rng = np.random.default_rng(seed)
    sp, A, mu, cv = synth_models(np.random.default_rng(7), N_MODELS, N_STATES, N_FEAT)
    X = rng.standard_normal((sample_utts * T_FRAMES, N_FEAT))
    lengths = [T_FRAMES] * sample_utts
    t0 = time.perf_counter()
    for m in range(N_MODELS):
        ho.score(ho.Params(sp[m], A[m], mu[m], cv[m]), X, lengths, loops=True)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for m in range(N_MODELS):
        ho.score(ho.Params(sp[m], A[m], mu[m], cv[m]), X, lengths, loops=False)
    dtv = time.perf_counter() - t1
    res = {"value": sample_utts * T_FRAMES / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{sample_utts} utterances x {T_FRAMES} frames x {N_FEAT} dims, 6 models x 8 "
                     f"states, fp64 Python loops as in the reference ({dt:.1f} s)",
           "vectorised_numpy_frames_per_s": sample_utts * T_FRAMES / dtv}
    from oracle import c_oracle as co
    if co.available():
        big = min(16 * sample_utts, 4096)
        Xb = rng.standard_normal((big * T_FRAMES, N_FEAT))
        t2 = time.perf_counter()
        for m in range(N_MODELS):
            co.score(ho.Params(sp[m], A[m], mu[m], cv[m]), Xb, [T_FRAMES] * big)
        res["compiled_c_frames_per_s"] = big * T_FRAMES / (time.perf_counter() - t2)
        # the same compiled oracle on the box's CPU share: utterances are independent, one
        # slice per worker thread (ctypes drops the interpreter lock inside the C call)
        from concurrent.futures import ThreadPoolExecutor
        workers = min(16, os.cpu_count() or 1)
        per = big // workers
        if per > 0:
            prm = [ho.Params(sp[m], A[m], mu[m], cv[m]) for m in range(N_MODELS)]

            def work(w):
                Xs = Xb[w * per * T_FRAMES:(w + 1) * per * T_FRAMES]
                for q in prm:
                    co.score(q, Xs, [T_FRAMES] * per)

            for rep in range(2):                       # second pass is the timed one
                t3 = time.perf_counter()
                with ThreadPoolExecutor(workers) as ex:
                    list(ex.map(work, range(workers)))
                dtp = time.perf_counter() - t3
            res["compiled_c_pool_frames_per_s"] = workers * per * T_FRAMES / dtp
            res["compiled_c_pool_workers"] = workers
    # the loop-faithful Python code itself on the box's CPU share: one process per core (child
    # processes that never touch the GPU), each scoring its own slice of utterances
    import subprocess
    workers = min(16, os.cpu_count() or 1)
    per = max(sample_utts // 8, 8)
    code = ("import sys, time, numpy as np; sys.path.insert(0, %r); from oracle import hmm_oracle as ho\n"
            "r = np.random.default_rng(7); M, S, D, T, n = %d, %d, %d, %d, %d\n"
            "sp, A = r.dirichlet(np.ones(S), size=M), r.dirichlet(np.ones(S), size=(M, S))\n"
            "mu, cv = r.standard_normal((M, S, D)), r.uniform(0.5, 1.5, (M, S, D))\n"
            "X = np.random.default_rng(int(sys.argv[1])).standard_normal((n * T, D))\n"
            "t0 = time.perf_counter()\n"
            "for m in range(M): ho.score(ho.Params(sp[m], A[m], mu[m], cv[m]), X, [T] * n, loops=True)\n"
            "print(time.perf_counter() - t0)\n") % (ROOT, N_MODELS, N_STATES, N_FEAT, T_FRAMES, per)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t4 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(w)], stdout=subprocess.PIPE, env=env)
             for w in range(workers)]
    ok = all(p.wait() == 0 for p in procs)
    dtq = time.perf_counter() - t4
    if ok:
        inner = max(float(p.stdout.read().decode().strip() or 0.0) for p in procs)
        res["python_loops_pool_frames_per_s"] = workers * per * T_FRAMES / max(inner, 1e-9)
        res["python_loops_pool_workers"] = workers
        res["python_loops_pool_wall_s"] = dtq
    return res


def launch_ranks(n, argv, dry_run=False):
    """Start one child per GPU and wait.  The parent makes no GPU call (a process
    that has initialised the GPU must not be replaced or forked).  Returns the exit
    status: 0 only if every rank succeeded."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            try:
                code = p.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            del pending[r]
            if code != 0:
                rc = rc or code or 1
                print(f"bench.py: rank {r} exited with status {code}", file=sys.stderr, flush=True)
                for q in pending.values():       # a failed rank leaves the others in a collective
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=N_UTTS, help="utterances per GPU (default = config)")
    ap.add_argument("--cpu-sample", type=int, default=768,
                    help="utterances of the workload the CPU baseline is timed on (about 15 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the audio -> label chain line")
    ap.add_argument("--pipeline-utts", type=int, default=65536,
                    help="recordings (3 s each) of the audio -> label chain line")
    ap.add_argument("--no-large", action="store_true", help="skip the large-model (1024-state GMM) lines")
    ap.add_argument("--large-utts", type=int, default=4096)
    ap.add_argument("--train-utts", type=int, default=0,
                    help="utterances per GPU for the Baum-Welch extra (default: the config's "
                         "1,048,576 utterances in total, split over the GPUs)")
    ap.add_argument("--prewarm", type=int, default=150,
                    help="untimed passes before the warm-up steps that bring the clock to its steady state")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-start timing (1 s idle, 20 passes)")
    ap.add_argument("--no-train", action="store_true", help="skip the Baum-Welch measurement")
    ap.add_argument("--no-verify", action="store_true",
                    help="N > 1: skip the comparison of the all-reduced statistics with rank 0's "
                         "E-step over all ranks' utterances")
    ap.add_argument("--all-extra", action="store_true",
                    help="run the secondary extra lines on more than one GPU too")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses GPU 0 (needs --backend gloo, "
                         "RCCL refuses two ranks on one device)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="launcher check: every rank prints its rank environment and exits "
                         "without touching a GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run_launch:
        print(json.dumps({k: os.environ.get(k) for k in
                          ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}),
              flush=True)
        return
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}; "
              "the launched rank count is what runs", file=sys.stderr)

    if os.environ.get("SRX_BENCH_WATCHDOG"):          # debugging aid: where does a rank hang?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SRX_BENCH_WATCHDOG"]), exit=True)

    import torch
    import torch.distributed as dist
    import speech_recognition_amd as sra

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local)
    ctx = sra.Context.default(local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def warm_gpu(fn, ms=50.0):
        """Run fn for about `ms` of GPU time: a section that follows host work starts on a clock
        that has dropped again, and the steady state is what the lines below quote."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            fn()
            if i % 2 == 1:
                e1.record()
                e1.synchronize()
                if e0.elapsed_time(e1) >= ms:
                    break

    # ---------------------------------------------------------------- workload
    U = args.utts
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    X = torch.randn((U * T_FRAMES, N_FEAT), device=dev, dtype=torch.float32, generator=gen)
    offsets = torch.arange(0, (U + 1) * T_FRAMES, T_FRAMES, device=dev, dtype=torch.int64)
    data = sra.Ragged(ctx, X, offsets=offsets, dtype="float32")
    sp, A, mu, cv = synth_models(np.random.default_rng(7), N_MODELS, N_STATES, N_FEAT)
    bank = sra.HMMBank(ctx, sp, A, mu, cv)
    scores = torch.empty((U, N_MODELS), dtype=torch.float64, device=dev)

    # The chip raises its clock over the first tens of milliseconds of load (a scoring pass
    # takes 0.67 ms right after start-up and 0.54 ms from about 40 ms on): the steady state is
    # what is measured, so load comes first, then the W warm-up steps, then the K timed ones.
    for _ in range(args.prewarm):
        bank.score(data, out=scores)
    for _ in range(args.warmup):
        bank.score(data, out=scores)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        bank.score(data, out=scores)
        ev[k][1].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    checksum = float(scores.sum().item())
    # What a caller that scores one batch out of the blue gets: the chip has dropped its clock
    # after a second without work, and raises it again over the first tens of milliseconds.
    cold_ms = None
    if not args.no_cold:
        time.sleep(1.0)
        evc = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evc:
            a.record()
            bank.score(data, out=scores)
            b.record()
        torch.cuda.synchronize()
        cold_ms = float(np.mean([a.elapsed_time(b) for a, b in evc]))
    assert np.isfinite(checksum) or os.environ.get("SRX_EXP_NOCHECK"), "scores are not finite"

    frames_per_step = U * T_FRAMES * world
    value = frames_per_step * args.steps / dt
    alg_bytes = U * T_FRAMES * N_FEAT * 4 + U * N_MODELS * 8      # per launch, per GPU
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("score_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "hmm_scored_frames_per_sec_6_models", "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (emission contraction as 3 split-bf16 partial products on the bf16 matrix "
                 "instruction, f32 accumulate; recursion f32; log-probability summed in f64)",
        "data": "synthetic",
        "config": {"workload": "batched scoring: 65,536 utterances x 200 frames x 39-dim vs 6 HMMs "
                               "x 8 states (diag Gaussians)",
                   "utterances_per_gpu": U, "frames_per_utterance": T_FRAMES,
                   "n_features": N_FEAT, "n_models": N_MODELS, "n_states": N_STATES,
                   "parallelism": f"utterances sharded over {world} GPU(s), no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes},
        "checksum": checksum,
        "prewarm_passes": args.prewarm,
        "ms_per_step_cold": cold_ms,
        "cold_note": "mean of the first 20 passes after 1 s without GPU work (clock ramp); "
                     "ms_per_step is the steady state after prewarm_passes + warmup",
    }

    # what the arithmetic named in "dtype" costs: the same scores in float64 (the validation
    # dtype, every product on the f64 matrix instruction) on the first utterances of the batch
    n_sub = min(U, 8192)
    d64 = sra.Ragged(ctx, X[:n_sub * T_FRAMES].double(), offsets=offsets[:n_sub + 1], dtype="float64")
    sc64 = bank.score(d64)
    out["max_rel_err_vs_f64"] = float(((sc64 - scores[:n_sub]).abs() / sc64.abs()).max().item())
    out["max_rel_err_vs_f64_utterances"] = n_sub
    del d64, sc64

    # ------------------------------------------------------------------ extras
    extra = {}
    if not args.no_extra and (world == 1 or args.all_extra):
        # stand-alone emission log-likelihoods and Viterbi decoding on the scoring batch
        ll = bank.emission_loglik(data)
        warm_gpu(lambda: bank.emission_loglik(data))
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            ll = bank.emission_loglik(data)
        barrier()
        dte = (time.perf_counter() - t0) / 3
        n_fr = U * T_FRAMES
        extra["emission"] = {"frames_per_s": n_fr * world / dte, "ms": dte * 1e3,
                             "hbm_frac": n_fr * (N_FEAT + N_MODELS * N_STATES) * 4 / dte / 1e9 / HBM_PEAK_GBS,
                             "config": "log-densities of 48 state columns, float32 in and out "
                                       "(156 B in + 192 B out per frame)"}
        del ll
        um = ctx.tensor((np.arange(U) % N_MODELS).astype(np.int32))
        vlp, vpath = bank.viterbi(data, um)
        warm_gpu(lambda: bank.viterbi(data, um))
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            vlp, vpath = bank.viterbi(data, um)
        barrier()
        dtv = (time.perf_counter() - t0) / 3
        extra["viterbi"] = {"frames_per_s": n_fr * world / dtv, "ms": dtv * 1e3,
                            "hbm_frac": n_fr * (N_FEAT * 4 + 4) / dtv / 1e9 / HBM_PEAK_GBS,
                            "config": "one 8-state model per utterance (u mod 6), int32 path out "
                                      "(156 B in + 4 B out per frame)"}
        del vlp, vpath
        plp, pgam = bank.posteriors(data, um)
        warm_gpu(lambda: bank.posteriors(data, um))
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            plp, pgam = bank.posteriors(data, um)
        barrier()
        dtp = (time.perf_counter() - t0) / 3
        extra["posteriors"] = {"frames_per_s": n_fr * world / dtp, "ms": dtp * 1e3,
                               "hbm_frac": n_fr * (N_FEAT + N_STATES) * 4 / dtp / 1e9 / HBM_PEAK_GBS,
                               "config": "state posteriors, one 8-state model per utterance (u mod 6), "
                                         "float32 (156 B in + 32 B out per frame)"}
        del plp, pgam, um
        # the headline workload with ragged lengths (100 to 300 frames, mean 200), scored with the
        # length order Ragged makes for such batches (utterances of similar length share a wave)
        lens_r = np.random.default_rng(4).integers(100, 301, size=U)
        n_r = int(lens_r.sum())
        if n_r <= X.shape[0]:
            data_r = sra.Ragged(ctx, X[:n_r], lens_r, dtype="float32")
            sc_r = bank.score(data_r)
            warm_gpu(lambda: bank.score(data_r, out=sc_r))
            barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                bank.score(data_r, out=sc_r)
            barrier()
            dtr = (time.perf_counter() - t0) / 10
            extra["score_ragged"] = {"frames_per_s": n_r * world / dtr, "ms": dtr * 1e3,
                                     "hbm_frac": n_r * N_FEAT * 4 / dtr / 1e9 / HBM_PEAK_GBS,
                                     "config": f"{U} utterances of 100 to 300 frames ({n_r} frames), 6 x 8 states, "
                                               "scored in descending length buckets (srx_forward_score_ordered)"}
            del data_r, sc_r
        # the headline workload in the validation dtype: float64 features, every product of
        # the emission contraction on the f64 matrix instruction
        X64 = X.double()
        data64 = sra.Ragged(ctx, X64, offsets=data.offsets, dtype="float64")
        sc64 = bank.score(data64)
        warm_gpu(lambda: bank.score(data64, out=sc64))
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            bank.score(data64, out=sc64)
        barrier()
        dt64 = (time.perf_counter() - t0) / 3
        extra["score_f64"] = {"frames_per_s": n_fr * world / dt64, "ms": dt64 * 1e3,
                              "hbm_frac": n_fr * N_FEAT * 8 / dt64 / 1e9 / HBM_PEAK_GBS,
                              "max_rel_diff_vs_f32": float(((sc64 - scores).abs() / sc64.abs()).max().item()),
                              "config": "the same 65,536 x 200 x 39 frames and 6 x 8-state models in float64 "
                                        "(312 B in per frame, one pass over X per tile of two models)"}
        del X64, data64, sc64
        del X, data
        torch.cuda.empty_cache()
        # front-end: 16 kHz, 25/10 ms, 512-pt FFT, 40 mel, 13 ceps + delta + delta-delta
        n_utt_fe, n_samp = 8192, 48000                      # 3 s pseudo-utterances
        g2 = torch.Generator(device=dev)
        g2.manual_seed(99 + rank)
        audio = 0.2 * torch.randn((n_utt_fe * n_samp,), device=dev, dtype=torch.float32, generator=g2)
        so = np.arange(0, (n_utt_fe + 1) * n_samp, n_samp, dtype=np.int64)
        plan = sra.MfccPlan(ctx, 16000, nfft=512, n_mel=40, delta_order=2, zscore=True)
        feats, fo = plan.run(audio, so)
        for _ in range(30):                      # (the clock settles over the first tens of ms of load)
            plan.run(audio, so, out=feats)
        barrier()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            plan.run(audio, so, out=feats)
        barrier()
        dtf = (time.perf_counter() - t0) / reps
        reps = 3
        raw_frames = n_utt_fe * ((n_samp - 400) // 160 + 1)
        # The transform kernel executes about 16.6 kflop of float32 vector / 4x4 matrix work per
        # frame (butterfly adds 5.1 k, multiplies and fused multiply-adds 4.8 k, window and
        # power 1.1 k, mel and DCT blocks 5.6 k): the 796 B/frame HBM line (1.0e10 frames/s) would
        # need 167 TFLOP/s of it, more than the 157.3 TFLOP/s vector peak -- this formulation is
        # bound by vector-instruction issue, which is what "vector_peak_frac" measures (shader
        # counters: profiles/r3_mfcc_sq_counters.txt).
        flop_per_frame = 16.6e3
        extra["mfcc"] = {"frames_per_s": raw_frames * world / dtf, "frames_per_gpu": raw_frames,
                         "ms": dtf * 1e3,
                         "hbm_frac": raw_frames * 796 / dtf / 1e9 / HBM_PEAK_GBS,
                         "bound": "vector ALU issue (749 vector + 108 LDS + 44 matrix instructions per wave-step of "
                                  "4 frames; the three waves of a SIMD keep its vector pipe busy 0.87-0.90 of the kernel's "
                                  "time by the shader counters); HBM line shown for reference",
                         "flop_per_frame_executed": flop_per_frame,
                         "vector_peak_frac": raw_frames * flop_per_frame / dtf / 157.3e12,
                         "config": "16 kHz, 400/160 samples, 512-pt FFT, 40 mel, 13 MFCC+d+dd, z-score"}
        # The whole chain the reference's recognize runs per file (HMM_final.py:49-52, :18-23), for
        # 65,536 recordings at once and with nothing leaving the device: audio -> 39-dim features
        # (the front-end above) -> scores against the 6 x 8-state bank -> label (arg max).
        if not args.no_pipeline:
            n_utt_p = args.pipeline_utts
            audio_p = torch.empty((n_utt_p * n_samp,), device=dev, dtype=torch.float32)
            for i0 in range(0, audio_p.numel(), 1 << 28):
                audio_p[i0:i0 + (1 << 28)].normal_(generator=g2).mul_(0.2)
            so_p = np.arange(0, (n_utt_p + 1) * n_samp, n_samp, dtype=np.int64)
            plan_p = sra.MfccPlan(ctx, 16000, nfft=512, n_mel=40, delta_order=2, zscore=True)
            data_p = plan_p.run_ragged(audio_p, so_p)
            feats_p = data_p.X
            lab_p, sc_p = bank.classify(data_p)

            def chain():
                d = plan_p.run_ragged(audio_p, so_p, out=feats_p)
                return bank.classify(d, scores=sc_p, labels=lab_p)

            def timed(fn, n):
                barrier()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                barrier()
                return (time.perf_counter() - t0) / n

            warm_gpu(chain, ms=100.0)
            dt_chain = timed(chain, 5)
            dt_fe = timed(lambda: plan_p.run_ragged(audio_p, so_p, out=feats_p), 5)
            dt_sc = timed(lambda: bank.score(data_p, out=sc_p), 5)
            dt_am = timed(lambda: sra.argmax_scores(ctx, sc_p, out=lab_p), 5)
            n_fr_p = int(data_p.n_frames)
            hist = torch.bincount(lab_p.long(), minlength=N_MODELS).cpu().tolist()
            extra["pipeline"] = {
                "utterances_per_s": n_utt_p * world / dt_chain, "frames_per_s": n_fr_p * world / dt_chain,
                "audio_seconds_per_s": n_utt_p * 3.0 * world / dt_chain, "ms": dt_chain * 1e3,
                "stage_ms": {"front_end": dt_fe * 1e3, "scoring": dt_sc * 1e3, "argmax": dt_am * 1e3},
                "stage_share": {"front_end": dt_fe / dt_chain, "scoring": dt_sc / dt_chain,
                                "argmax": dt_am / dt_chain},
                "hbm_frac": (n_utt_p * n_samp * 4 + n_utt_p * 4) / dt_chain / 1e9 / HBM_PEAK_GBS,
                "label_histogram": hist,
                "config": f"{n_utt_p} recordings x 3 s at 16 kHz (float32, resident) -> 13 MFCC+d+dd, z-score "
                          f"({n_fr_p // n_utt_p} frames each) -> 6 x 8-state scores -> int32 label; "
                          "algorithmic bytes: 4 B per sample in + 4 B per recording out"}
            del audio_p, data_p, feats_p, lab_p, sc_p, plan_p
            torch.cuda.empty_cache()
        # the reference's other front-end (extract_mfcc, HMM_lib.py:11-15): librosa's conventions,
        # 2048-point transform every 512 samples, 128 Slaney filters, top_db clamp, 13 cepstra,
        # z-score; the same buffer read as 22.05 kHz audio
        plan_l = sra.MfccPlan.librosa(ctx, 22050)
        feats_l, fo_l = plan_l.run(audio, so)
        warm_gpu(lambda: plan_l.run(audio, so, out=feats_l))
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            plan_l.run(audio, so, out=feats_l)
        barrier()
        dtl = (time.perf_counter() - t0) / reps
        lib_frames = n_utt_fe * (n_samp // 512 + 1)
        extra["mfcc_librosa"] = {"frames_per_s": lib_frames * world / dtl, "frames_per_gpu": lib_frames,
                                 "ms": dtl * 1e3,
                                 "hbm_frac": lib_frames * (512 * 4 + 13 * 4) / dtl / 1e9 / HBM_PEAK_GBS,
                                 "bound": "vector ALU issue (1879 vector + 190 LDS instructions per wave-step of 2 "
                                          "frames; the two waves of a SIMD keep its vector pipe busy 0.81 of the "
                                          "kernel's time by the shader counters); HBM line shown for reference",
                                 "config": "librosa.feature.mfcc conventions: 22.05 kHz, centred 2048-pt frames "
                                           "every 512 samples, Hann, 128 Slaney mel, 10 log10, top_db 80, 13 MFCC, "
                                           "z-score (2048-point transform in registers, 32 lanes per frame)"}
        # the chain recognize_lib runs per file (HMM_lib.py:11-15, :33, :85-97), for all recordings
        # at once and with nothing leaving the device: audio -> 13 librosa-convention cepstra,
        # z-scored -> scores against 6 word models of 3 states (GaussianHMM(n_components=3)) -> label
        rng_l = np.random.default_rng(5)
        bank_l = sra.HMMBank(ctx, rng_l.dirichlet(np.ones(3), size=N_MODELS),
                             rng_l.dirichlet(np.ones(3), size=(N_MODELS, 3)),
                             rng_l.standard_normal((N_MODELS, 3, 13)), rng_l.uniform(0.5, 1.5, (N_MODELS, 3, 13)))
        data_l = plan_l.run_ragged(audio, so)
        lab_l, sc_l = bank_l.classify(data_l)

        def chain_l():
            d = plan_l.run_ragged(audio, so, out=data_l.X)
            return bank_l.classify(d, scores=sc_l, labels=lab_l)

        warm_gpu(chain_l)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            chain_l()
        barrier()
        dtc = (time.perf_counter() - t0) / reps
        extra["pipeline_librosa"] = {
            "utterances_per_s": n_utt_fe * world / dtc, "frames_per_s": lib_frames * world / dtc, "ms": dtc * 1e3,
            "config": f"{n_utt_fe} recordings x {n_samp} samples read as 22.05 kHz (float32, resident) -> 13 "
                      "librosa-convention MFCC, z-score -> 6 x 3-state scores -> int32 label"}
        del feats_l, data_l, lab_l, sc_l, bank_l
        # end-pointing pre-pass over the same recordings (10 ms block energies and
        # zero crossings, thresholds, run search)
        # (the offsets of the batch layout go to the device once, like the front-end's)
        lay = sra.endpoint.BatchLayout(so, 16000, ctx=ctx)
        ep = sra.endpoint.detect_batch(audio, lay, 16000, ctx=ctx)
        warm_gpu(lambda: sra.endpoint.detect_batch(audio, lay, 16000, ctx=ctx))
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            ep = sra.endpoint.detect_batch(audio, lay, 16000, ctx=ctx)
        barrier()
        dte = (time.perf_counter() - t0) / reps
        extra["endpoint"] = {"samples_per_s": n_utt_fe * n_samp * world / dte, "ms": dte * 1e3,
                             "hbm_frac": n_utt_fe * n_samp * 4 / dte / 1e9 / HBM_PEAK_GBS,
                             "config": "8192 recordings x 3 s at 16 kHz, float32 (4 B in per sample)"}
        del audio, feats, ep
        torch.cuda.empty_cache()
        if not args.no_large:
            # large model: 1024 states x 16-component diagonal GMMs x 39 dims, 4096 x 200 frames
            Sl, Ml, Ul = 1024, 16, args.large_utts
            rl = np.random.default_rng(5)
            em = sra.GmmEmission(ctx, rl.dirichlet(np.ones(Ml), size=Sl),
                                 rl.standard_normal((Sl, Ml, N_FEAT)), rl.uniform(0.5, 1.5, (Sl, Ml, N_FEAT)))
            g4 = torch.Generator(device=dev)
            g4.manual_seed(777 + rank)
            Xl = torch.randn((Ul * T_FRAMES, N_FEAT), device=dev, dtype=torch.float32, generator=g4)
            offl = torch.arange(0, (Ul + 1) * T_FRAMES, T_FRAMES, device=dev, dtype=torch.int64)
            dl = sra.Ragged(ctx, Xl, offsets=offl, dtype="float32")
            lll = em.loglik(dl)
            warm_gpu(lambda: em.loglik(dl, out=lll))
            barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                em.loglik(dl, out=lll)
            barrier()
            dle = (time.perf_counter() - t0) / 3
            fl = 2.0 * 2 * N_FEAT * Sl * Ml * Ul * T_FRAMES
            extra["large_emission"] = {
                "frames_per_s": Ul * T_FRAMES * world / dle, "ms": dle * 1e3,
                "tflops_algorithmic": fl / dle / 1e12,
                "mfma_frac": 3.0 * fl / dle / 1e12 / 2500.0,
                "config": "1024 states x 16 diag components x 39 dims, log-sum-exp per state; split-bf16 "
                          "(3 bf16 products per f32 product) against the 2.5 PFLOP/s dense bf16 peak"}
            Al = ctx.tensor(rl.dirichlet(np.ones(Sl), size=Sl))
            pil = ctx.tensor(rl.dirichlet(np.ones(Sl)))
            vl, vp = sra.viterbi_dense(ctx, lll, offl, pil, Al)
            warm_gpu(lambda: sra.viterbi_dense(ctx, lll, offl, pil, Al))
            barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                vl, vp = sra.viterbi_dense(ctx, lll, offl, pil, Al)
            barrier()
            dlv = (time.perf_counter() - t0) / 3
            extra["large_viterbi"] = {
                "frames_per_s": Ul * T_FRAMES * world / dlv, "ms": dlv * 1e3,
                "tera_maxplus_per_s": Ul * T_FRAMES * float(Sl) * Sl / dlv / 1e12,
                "config": "dense 1024 x 1024 log A, (max,+) S^2 per frame on the vector ALUs, "
                          "backtrace by recomputation"}

    # ---------------------------------------------------------------- training
    # Baum-Welch on the config's 1,048,576 utterances in total (fixed total work: strong
    # scaling), 6 models x 16 states, utterance u trains model u mod 6; one EM iteration =
    # E-step over the rank's shard + ONE all-reduce of all models' statistics + M-step
    if not args.no_train:
        for name in ("X", "data", "scores"):
            locals().pop(name, None)
        X = data = scores = None
        torch.cuda.empty_cache()
        Ut = args.train_utts if args.train_utts > 0 else (1 << 20) // world
        g3 = torch.Generator(device=dev)
        g3.manual_seed(4321 + rank)
        Xt = torch.empty((Ut * T_FRAMES, N_FEAT), device=dev, dtype=torch.float32)
        for i0 in range(0, Ut * T_FRAMES, 1 << 24):          # 31 GB at one GPU: fill in pieces
            Xt[i0:i0 + (1 << 24)].normal_(generator=g3)
        offt = torch.arange(0, (Ut + 1) * T_FRAMES, T_FRAMES, device=dev, dtype=torch.int64)
        dt_ = sra.Ragged(ctx, Xt, offsets=offt, dtype="float32")
        sp2, A2, mu2, cv2 = synth_models(np.random.default_rng(8), N_MODELS, 16, N_FEAT)
        bank2 = sra.HMMBank(ctx, sp2, A2, mu2, cv2)
        tr = sra.parallel.BaumWelchTrainer(bank2, dt_, np.arange(Ut) % N_MODELS)
        for _ in range(3):
            tr.step()
        barrier()
        t0 = time.perf_counter()
        iters = 5
        for _ in range(iters):
            lp = tr.step()
        barrier()
        dtt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dtt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtt = float(t.item())
        dtt /= iters
        # the exchange step alone: the all-reduce of the packed float64 records, back to back
        ar_ms, ranks_seen = 0.0, 1
        if world > 1:
            one = torch.ones((1,), device=dev, dtype=torch.float64)
            dist.all_reduce(one, op=dist.ReduceOp.SUM)
            ranks_seen = int(round(float(one.item())))
            buf = tr.stats.clone()
            for _ in range(5):
                sra.parallel.allreduce_stats(buf)
            barrier()
            t0 = time.perf_counter()
            for _ in range(50):
                sra.parallel.allreduce_stats(buf)
            torch.cuda.synchronize()
            ar_ms = (time.perf_counter() - t0) / 50 * 1e3
        # the same iteration as the reference's own fit() runs it: start and transition
        # probabilities only (my_hmmlearn/hmm.py:213), reported beside the full one
        bank3 = sra.HMMBank(ctx, sp2, A2, mu2, cv2)
        tr3 = sra.parallel.BaumWelchTrainer(bank3, dt_, np.arange(Ut) % N_MODELS, mstep="transitions_only")
        for _ in range(2):
            tr3.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            tr3.step()
        barrier()
        dt3 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt3], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt3 = float(t.item())
        dt3 /= iters
        del tr3, bank3
        # The sharded E-step against the unsharded one: every rank's statistics of one E-step
        # from the initial parameters, summed by the all-reduce, beside rank 0's E-step over
        # all ranks' utterances in one launch (their features regenerated from the ranks' seeds).
        stats_err = None
        if world > 1 and not args.no_verify:
            bank4 = sra.HMMBank(ctx, sp2, A2, mu2, cv2)
            um_t = ctx.tensor((np.arange(Ut) % N_MODELS).astype(np.int32))
            st_sum = sra.parallel.allreduce_stats(bank4.estep(dt_, um_t).clone())
            if rank == 0:
                Xall = torch.empty((world * Ut * T_FRAMES, N_FEAT), device=dev, dtype=torch.float32)
                for r in range(world):
                    gr = torch.Generator(device=dev)
                    gr.manual_seed(4321 + r)
                    part = Xall[r * Ut * T_FRAMES:(r + 1) * Ut * T_FRAMES]
                    for i0 in range(0, Ut * T_FRAMES, 1 << 24):
                        part[i0:i0 + (1 << 24)].normal_(generator=gr)
                offa = torch.arange(0, (world * Ut + 1) * T_FRAMES, T_FRAMES, device=dev, dtype=torch.int64)
                um_a = ctx.tensor(np.tile(np.arange(Ut) % N_MODELS, world).astype(np.int32))
                st_one = bank4.estep(sra.Ragged(ctx, Xall, offsets=offa, dtype="float32"), um_a)
                # largest difference relative to the largest entry of its own array (count,
                # log-probability, start, trans, post, obs, obs2 of each model)
                S2 = 16
                cuts = np.cumsum([0, 1, 1, S2, S2 * S2, S2, S2 * N_FEAT, S2 * N_FEAT])
                stats_err = 0.0
                for a, b in zip(cuts[:-1], cuts[1:]):
                    seg_one, seg_sum = st_one[:, a:b], st_sum[:, a:b]
                    scale = seg_one.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
                    stats_err = max(stats_err, float(((seg_sum - seg_one).abs() / scale).max().item()))
                del Xall, st_one
            barrier()
        out["sharded_vs_single_rank_stats_max_rel_err"] = stats_err
        out["baum_welch_iter_per_s"] = 1.0 / dtt
        out["allreduce_ms_per_iter"] = ar_ms
        out["rccl_ranks"] = ranks_seen
        out["collective_backend"] = dist.get_backend() if world > 1 else None
        extra["baum_welch"] = {"iter_per_s": 1.0 / dtt, "ms_per_iter": dtt * 1e3,
                               "utterances_total": Ut * world, "frames_per_s": Ut * world * T_FRAMES / dtt,
                               "scaling": "strong (fixed total)" if args.train_utts <= 0 else "weak",
                               "config": "6 models x 16 states x 39 dims, full M-step, one all-reduce of "
                                         f"{tr.stats.numel()} float64 per iteration",
                               "allreduce_ms": ar_ms, "rccl_ranks": ranks_seen,
                               # the E-step kernel reads X twice and writes and reads its float32
                               # lattice: 446 B/frame by the counters (profiles/traffic_secondary.json),
                               # against 156 B/frame of features; per GPU
                               "hbm_frac_algorithmic": Ut * T_FRAMES * 156.0 / dtt / 1e9 / HBM_PEAK_GBS,
                               "hbm_frac_measured_traffic": Ut * T_FRAMES * 446.0 / dtt / 1e9 / HBM_PEAK_GBS,
                               "logprob_model0": float(lp[0].item()),
                               "transitions_only_iter_per_s": 1.0 / dt3,
                               "transitions_only_ms_per_iter": dt3 * 1e3}
        del Xt, dt_, tr
    if extra:
        out["extra"] = extra

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
