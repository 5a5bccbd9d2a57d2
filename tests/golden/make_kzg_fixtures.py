"""Extracts the EIP-4844 KZG data the reference holds into tests/golden/ (run in the authoring container only; the
reference does not exist on the GPU box).  Everything written is DATA, cited per file; no reference source text is copied.

  kzg_trusted_setup.txt      <- config/kzg_trusted_setup.txt (the public Ethereum ceremony output the reference embeds,
                                config/kzg_setup.go:7-8): 4096 G1 Lagrange + 65 G2 monomial + 4096 G1 monomial points
  kzg_blobdata1.txt          <- crypto/blobs/testdata/blobdata1.txt (test blob of crypto/blobs/barycentric_test.go:41-63)
  kzg_blobdata2.txt          <- crypto/blobs/testdata/blobdata2.txt (barycentric_test.go:65-87)
  bls12381_omega_brp_4096.json <- the 4096 numbers of crypto/blobs/omega.go:4-4101 (bit-reversed 4096th roots of unity of
                                the BLS12-381 scalar field) + the root-of-unity literal of crypto/blobs/barycentric.go:52
"""
import json, os, re, shutil
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
shutil.copyfile(os.path.join(REF, "config/kzg_trusted_setup.txt"), os.path.join(HERE, "kzg_trusted_setup.txt"))
for k in (1, 2):
    shutil.copyfile(os.path.join(REF, "crypto/blobs/testdata/blobdata%d.txt" % k), os.path.join(HERE, "kzg_blobdata%d.txt" % k))
omega = re.findall(r'"(0x[0-9a-fA-F]+)"', open(os.path.join(REF, "crypto/blobs/omega.go")).read())
assert len(omega) == 4096
root = re.search(r'rootOfUnity\.SetString\("(\d+)"\)', open(os.path.join(REF, "crypto/blobs/barycentric.go")).read()).group(1)
json.dump({"_source": {"omega_brp": "crypto/blobs/omega.go:4-4101", "root_of_unity_2^32": "crypto/blobs/barycentric.go:52",
                       "domain_generator_exponent": "crypto/blobs/barycentric.go:58 (2^20)"},
           "root_of_unity_2^32": root, "omega_brp": omega},
          open(os.path.join(HERE, "bls12381_omega_brp_4096.json"), "w"))
print("wrote kzg fixtures")
