"""Print the headline fields of a bench.py JSON line (file argument), one per row."""
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.lstrip().startswith("{")][-1]
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"])
r = d["roofline"]
print("kernel_ms", r["kernel_ms"], "frac", r["frac"], "int frac", r.get("int_issue_context", {}).get("frac"))
cb = d.get("cpu_baseline") or {}
print("cpu", cb.get("value"), cb.get("unit"), cb.get("cores"))
for k, v in d["config"].items():
    print("config", k, v)
for k, v in (d.get("extras") or {}).items():
    print("extra", k, json.dumps(v)[:260])
