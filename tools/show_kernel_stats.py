"""Print a rocprofv3 kernel_stats.csv compactly: name, calls, average and maximum duration in microseconds."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = ("k_points_expand", "k_points_progression", "k_scalars_synthetic")
other = 0.0
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")[:64]
    avg = float(r["AverageNs"]) / 1e3
    print(f"{name:66s} {r['Calls']:>4s} avg {avg:10.1f} us  max {float(r['MaxNs'])/1e3:10.1f}")
    if "k_accumulate" not in name and not any(k in name for k in skip):
        other += avg
print(f"sum of the averages outside k_accumulate and input generation: {other:.1f} us")
