"""Extracts the BN254 constants and points the reference holds into tests/golden/reference_constants.json.

Run in the authoring container only (reads /root/reference, which does not exist on the GPU box).  The output
is DATA (numbers the reference's own files state), cited per entry; no reference source text is copied.
"""
import json, re, os
REF = "/root/reference"
sol = open(os.path.join(REF, "config/statetransition_vkey.sol")).read()
def const(name):
    m = re.search(r"uint256 constant %s = (0x[0-9a-fA-F]+|\d+);" % name, sol)
    return int(m.group(1), 0)
out = {"_source": {"sol": "config/statetransition_vkey.sol:42-114", "dummy": "circuits/voteverifier/dummy.go:15-47",
                   "gen": "crypto/ecc/bn254/bn254.go:24-29"}}
for k in ("P", "R", "FRACTION_1_2_FP", "FRACTION_27_82_FP", "FRACTION_3_82_FP", "EXP_INVERSE_FP", "EXP_SQRT_FP"):
    out[k] = str(const(k))
g1 = {}
for name in ["ALPHA", "CONSTANT"] + ["PUB_%d" % i for i in range(9)]:
    g1[name] = [str(const(name + "_X")), str(const(name + "_Y"))]
out["g1_points"] = g1
g2 = {}
for name in ("BETA_NEG", "GAMMA_NEG", "DELTA_NEG", "PEDERSEN_G", "PEDERSEN_GSIGMANEG"):
    g2[name] = [[str(const(name + "_X_0")), str(const(name + "_X_1"))], [str(const(name + "_Y_0")), str(const(name + "_Y_1"))]]
out["g2_points"] = g2
dummy = open(os.path.join(REF, "circuits/voteverifier/dummy.go")).read()
proof = json.loads(re.search(r"dummyBallotProof = `(.*?)`", dummy, re.S).group(1))
out["snarkjs_proof"] = {"pi_a": proof["pi_a"][:2], "pi_b": proof["pi_b"][:2], "pi_c": proof["pi_c"][:2]}
json.dump(out, open(os.path.join(os.path.dirname(__file__), "reference_constants.json"), "w"), indent=1)
print("wrote reference_constants.json")
