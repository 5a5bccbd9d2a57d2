"""Condense rocprofv3 csv output (kernel stats + per-dispatch counters) into small files for profiles/."""
import csv, glob, json, os, re, sys
out = sys.argv[1]
def short(n): return re.sub(r"\(.*", "", n).replace("void ", "")
st = glob.glob(os.path.join(out, "bench", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
pmc = {}
for d in glob.glob(os.path.join(out, "pmc_*")):
    if not os.path.isdir(d): continue
    for cc in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(cc)):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]; v = float(r["Counter_Value"])
            e = pmc.setdefault(k, {}).setdefault(c, {"max": 0.0, "sum": 0.0, "dispatches": 0})
            e["max"] = max(e["max"], v); e["sum"] += v; e["dispatches"] += 1
json.dump(pmc, open(os.path.join(out, "pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)
print("kernels with counters:", len(pmc))
