"""Turn the rocprofv3 outputs of tools/profile_score.sh (gpurun_out/prof_*) into
the committed summaries under profiles/: kernel statistics, the HBM counter
rows of the scoring kernel, and traffic.json (bytes per launch, corrected as
the gfx950 counters need: FETCH_SIZE counts 128-byte requests as 64 bytes, so
x2; WRITE_SIZE is exact; both in KiB).

    python tools/make_traffic.py [round-tag, default r1]
"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
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
    "kernel_avg_ns_rocprof": avg_ns,
    "command": "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) "
               "-- python3 bench.py --steps 10 --warmup 2 --no-extra --no-cpu",
}
json.dump(res, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
