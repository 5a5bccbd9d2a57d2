"""Generates the committed golden vectors from the pure-Python big-int oracle (oracle/bn254_py.py).

The reference holds no MSM/NTT/proof vectors, so these pin the C oracle and
the HIP library against the independent Python restatement.  Deterministic: rerunning reproduces the files.
"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import bn254_py as o

def hx(v): return "%064x" % v

def msm_case(name, pts, ks, g2=False):
    res = (o.g2_msm if g2 else o.g1_msm)(pts, ks)
    enc = (lambda p: None if p is None else ([hx(p[0][0]), hx(p[0][1]), hx(p[1][0]), hx(p[1][1])] if g2 else [hx(p[0]), hx(p[1])]))
    return {"name": name, "points": [enc(p) for p in pts], "scalars": [hx(k) for k in ks], "result": enc(res)}

out = {"msm_g1": [], "msm_g2": [], "ntt": [], "compute_h": [], "groth16": []}
P = o.synthetic_g1_points(0xDA71C1, 64)
uni = [o.fr_at(1, i) for i in range(64)]
out["msm_g1"].append(msm_case("uniform64", P, uni))
out["msm_g1"].append(msm_case("witness_like64", P, [o.witness_like_scalar(1, i) for i in range(64)]))
out["msm_g1"].append(msm_case("n1", P[:1], uni[:1]))
out["msm_g1"].append(msm_case("n37_ragged", P[:37], uni[:37]))
out["msm_g1"].append(msm_case("all_zero_scalars", P[:16], [0] * 16))
out["msm_g1"].append(msm_case("r_minus_1", P[:16], [o.R - 1] * 16))
out["msm_g1"].append(msm_case("all_ones", P[:33], [1] * 33))
out["msm_g1"].append(msm_case("infinity_points", [None, P[0], None, P[1], None], uni[:5]))
out["msm_g1"].append(msm_case("repeated_point_same_bucket", [P[3]] * 20, [5] * 20))
out["msm_g1"].append(msm_case("p_and_minus_p", [P[4], o.g1_neg(P[4]), P[5], o.g1_neg(P[5])], [7, 7, 9, 9]))
out["msm_g1"].append(msm_case("p_and_minus_p_residual", [P[4], o.g1_neg(P[4]), P[6]], [uni[0], uni[0], 3]))
out["msm_g1"].append(msm_case("window_boundary_scalars", P[:8], [(1 << 16) - 1, 1 << 15, (1 << 15) + 1, (1 << 32) - 1, (1 << 253), o.R - 2, 1 << 128, (1 << 64) + (1 << 15)]))
Q = o.synthetic_g2_points(5, 24)
out["msm_g2"].append(msm_case("uniform24", Q, uni[:24], g2=True))
out["msm_g2"].append(msm_case("edge", [Q[0], o.g2_neg(Q[0]), None, Q[1], Q[1]], [3, 3, 9, 0, o.R - 1], g2=True))
for log_n in (1, 3, 6):
    n = 1 << log_n
    x = [o.fr_at(2, i) for i in range(n)]
    d = o.Domain(n)
    for inv in (0, 1):
        for dec in (0, 1):
            for cos in (0, 1):
                y = list(x)
                (d.fft_inverse if inv else d.fft)(y, dec, bool(cos))
                out["ntt"].append({"log_n": log_n, "inverse": inv, "decimation": dec, "coset": cos, "input": [hx(v) for v in x], "output": [hx(v) for v in y]})
for log_n, m in ((3, 8), (5, 27), (6, 50)):
    a = [o.fr_at(3, i) for i in range(m)]; b = [o.fr_at(4, i) for i in range(m)]; c = [o.fr_at(5, i) for i in range(m)]
    out["compute_h"].append({"log_n": log_n, "a": [hx(v) for v in a], "b": [hx(v) for v in b], "c": [hx(v) for v in c],
                             "h": [hx(v) for v in o.compute_h(a, b, c, o.Domain(1 << log_n))]})
json.dump(out, open(os.path.join(HERE, "oracle_vectors.json"), "w"))
print("wrote oracle_vectors.json", os.path.getsize(os.path.join(HERE, "oracle_vectors.json")), "bytes")
