"""Summarise the rocprofv3 passes of tools/prof_k1.sh into profiles/ (tracked):
   profiles/<tag>_k1_kernel_stats.csv, profiles/<tag>_k1_pmc.json and profiles/k1_traffic.json (read by bench.py).
HBM traffic follows MI355X_MICROARCH.md: FETCH_SIZE (KB) is doubled on gfx950 for 16-byte-per-lane coalesced
streaming reads; WRITE_SIZE (KB) is taken as is."""
import collections, csv, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(out, f"{tag}_k1_kernel_stats.csv"))
pmc = {}
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    path = os.path.join(src, d, "p_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "sp_moments" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
kern = [r for r in csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))) if "sp_moments" in r["Name"]][0]
fetch = pmc["FETCH_SIZE"]["per_launch_mean"] * 1024.0 * 2.0     # KB -> B, gfx950 x2 correction
write = pmc["WRITE_SIZE"]["per_launch_mean"] * 1024.0
summary = {"kernel": kern["Name"], "avg_ns": float(kern["AverageNs"]), "calls": int(kern["Calls"]),
           "workload": "256 frames x 640x480 (C5 batch), tools/k1_prof.py", "counters": pmc,
           "hbm_fetch_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
           "hbm_bytes_per_launch": fetch + write,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950, 16 B/lane coalesced reads); separate --pmc passes"}
json.dump(summary, open(os.path.join(out, f"{tag}_k1_pmc.json"), "w"), indent=1)
json.dump({"hbm_bytes_per_launch": fetch + write, "source": f"profiles/{tag}_k1_pmc.json"},
          open(os.path.join(out, "k1_traffic.json"), "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("avg_ns", "hbm_bytes_per_launch")}))
