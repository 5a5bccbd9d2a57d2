"""Reads a rocprofv3 kernel trace of tools/gpu_groth16.py (one proof size) and prints the last proof's timeline: per kernel start / duration /
queue, and how much of the proof's span has at least one k_accumulate (the VALU-bound work) running."""
import csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0])))
ev = []
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("dvn::", "")
    n = re.sub(r"<.*Fe2P.*", "<G2>", n); n = re.sub(r"<.*", "", n) if "<G2>" not in n else n
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Queue_Id"]))
ev.sort()
# a proof starts with the wire filter (two k_gather_fr launches on the main stream); the last proof = everything from the last such
# pair on, minus a small lead for the witness copy
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_gather_fr")]
i0 = starts[-2] if len(starts) >= 2 else max(0, len(ev) - 150)
t0 = ev[i0][0]; t1 = max(e[1] for e in ev[i0:])
print(f"proof span {(t1 - t0) / 1e6:.3f} ms, {len(ev) - i0} kernels")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
acc = [(s, e) for s, e, n, q in ev[i0:] if n.startswith("k_accumulate")]
ntt = [(s, e) for s, e, n, q in ev[i0:] if n.startswith("k_ntt") or n.startswith("k_h_")]
anyk = [(s, e) for s, e, n, q in ev[i0:]]
print(f"accumulate running {union(acc) / 1e6:.3f} ms, ntt running {union(ntt) / 1e6:.3f} ms, any kernel {union(anyk) / 1e6:.3f} ms")
for s, e, n, q in ev[i0:]:
    if e - s > 30000: print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f} q{q} {n}")
