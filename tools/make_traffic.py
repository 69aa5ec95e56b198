"""Turn the rocprofv3 outputs of tools/profile_score.sh (gpurun_out/prof_*) into
the committed summaries under profiles/: kernel statistics, the HBM counter
rows of the scoring kernel, and traffic.json (bytes per launch, corrected as
the gfx950 counters need: FETCH_SIZE counts 128-byte requests as 64 bytes, so
x2; WRITE_SIZE is exact; both in KiB).

    python tools/make_traffic.py [round-tag, default r2]

Also, when present: the kernel statistics of the secondary benches (tools/profile_all.sh) and
the counter passes of the E-step and the front-end (tools/profile_traffic.sh) ->
profiles/<tag>_<name>_kernel_stats.csv, profiles/traffic_secondary.json.
"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
KERNEL = "score_split_kernel"


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def short(name):
    return name if len(name) < 200 else name[:120] + " ...<torch kernel>"


# kernel statistics, torch's template names shortened
stats = rows(os.path.join(G, "prof_kt", "score_kernel_stats.csv"))
with open(os.path.join(P, f"{tag}_score_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
    w.writeheader()
    for r in stats:
        r["Name"] = short(r["Name"])
        w.writerow(r)
krow = [r for r in stats if KERNEL in r["Name"]][0]
avg_ns = float(krow["AverageNs"])
import re
kname = re.search(r"(score_split_kernel<[^>]*>)", krow["Name"]).group(1)
# The average above mixes the passes that bring the clock up (bench.py: 150 by default) with the
# steady state.  The timed region of the bench is the K launches after the pre-warm and warm-up
# ones: their durations come from the kernel trace, in dispatch order.
PREWARM, WARMUP, STEPS = 150, 2, 10
trace = [r for r in rows(os.path.join(G, "prof_kt", "score_kernel_trace.csv")) if kname in r["Kernel_Name"]]
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace]
timed = dur[PREWARM + WARMUP:PREWARM + WARMUP + STEPS]
with open(os.path.join(P, f"{tag}_score_kernel_trace_timed_region.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["launch_index", "kernel", "duration_ns", "part"])
    for i, d in enumerate(dur[:PREWARM + WARMUP + STEPS]):
        part = "prewarm" if i < PREWARM else ("warmup" if i < PREWARM + WARMUP else "timed")
        w.writerow([i, kname, d, part])

out = {}
for cname, sub in (("FETCH_SIZE", "prof_fetch"), ("WRITE_SIZE", "prof_write")):
    rs = [r for r in rows(os.path.join(G, sub, "score_counter_collection.csv"))
          if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == cname]
    with open(os.path.join(P, f"{tag}_score_pmc_{cname.split('_')[0].lower()}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for r in rs:
            w.writerow([r["Dispatch_Id"], short(r["Kernel_Name"]), cname, r["Counter_Value"]])
    vals = [float(r["Counter_Value"]) for r in rs]
    out[cname] = sum(vals) / len(vals)

fetch_b = out["FETCH_SIZE"] * 1024 * 2
write_b = out["WRITE_SIZE"] * 1024
U, T, D, M = 65536, 200, 39, 6
res = {
    "score_bytes_per_launch": fetch_b + write_b,
    "fetch_size_kb_raw": out["FETCH_SIZE"],
    "write_size_kb_raw": out["WRITE_SIZE"],
    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE exact; KB = 1024 B",
    "algorithmic_bytes_per_launch": U * T * D * 4 + U * M * 8,
    "kernel": kname,
    "kernel_avg_ns_rocprof_all_launches": avg_ns,
    "kernel_avg_ns_rocprof_timed_region": sum(timed) / max(len(timed), 1),
    "kernel_min_median_max_ns_timed_region": [min(timed), sorted(timed)[len(timed) // 2], max(timed)] if timed else None,
    "kernel_avg_ns_first_20_launches": sum(dur[:20]) / max(len(dur[:20]), 1),
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu (kernel times); --pmc FETCH_SIZE / "
               "--pmc WRITE_SIZE in passes of their own -- python3 bench.py --no-extra --no-train --no-cpu",
}
json.dump(res, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))

# ---- secondary benches: kernel statistics and HBM counters
import glob
for d in sorted(glob.glob(os.path.join(G, "prof_*"))):
    name = os.path.basename(d)[5:]
    src = os.path.join(d, f"{name}_kernel_stats.csv")
    if name in ("kt", "fetch", "write") or not os.path.exists(src):
        continue
    st = rows(src)
    with open(os.path.join(P, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(st[0].keys()))
        w.writeheader()
        for r in st:
            r["Name"] = short(r["Name"])
            w.writerow(r)


def counter(name, sub, cname, kernels):
    """Average corrected bytes per launch of each kernel; the rows go to profiles/ too."""
    path = os.path.join(G, f"pmc_{name}_{sub}", f"{name}_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    res = {}
    with open(os.path.join(P, f"{tag}_{name}_pmc_{sub}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for k in kernels:
            rs = [r for r in rows(path) if k in r["Kernel_Name"] and r["Counter_Name"] == cname]
            for r in rs:
                w.writerow([r["Dispatch_Id"], short(r["Kernel_Name"])[:100], cname, r["Counter_Value"]])
            vals = [float(r["Counter_Value"]) for r in rs]
            if vals:
                res[k] = sum(vals) / len(vals) * (2048.0 if cname == "FETCH_SIZE" else 1024.0)
    return res


sec = {}
for name, kernels in (("estep", ["estep_fused_kernel"]), ("mfcc", ["cepstra_fast_kernel", "finish_stream_kernel"])):
    fb, wb = counter(name, "fetch", "FETCH_SIZE", kernels), counter(name, "write", "WRITE_SIZE", kernels)
    for k in kernels:
        if k in fb and k in wb:
            sec[k] = {"FETCH_SIZE_bytes_per_launch": fb[k], "WRITE_SIZE_bytes_per_launch": wb[k]}
if "estep_fused_kernel" in sec:
    fr = 131072 * 200
    e = sec["estep_fused_kernel"]
    e.update(workload="131,072 utterances x 200 frames x 39 dims, 6 models x 16 states", frames=fr,
             bytes_per_frame_measured=(e["FETCH_SIZE_bytes_per_launch"] + e["WRITE_SIZE_bytes_per_launch"]) / fr,
             bytes_per_frame_design=440)
if "cepstra_fast_kernel" in sec:
    c = sec["cepstra_fast_kernel"]
    c.update(workload="8192 recordings x 48,000 samples, 16 kHz, 512-point FFT", sample_bytes=8192 * 48000 * 4,
             fetch_over_sample_bytes=c["FETCH_SIZE_bytes_per_launch"] / (8192 * 48000 * 4))
if sec:
    sec["correction"] = ("FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE exact; KB = 1024 B; "
                         "separate --pmc passes (tools/profile_traffic.sh)")
    json.dump(sec, open(os.path.join(P, "traffic_secondary.json"), "w"), indent=1)
    print(json.dumps(sec, indent=1))
