#!/bin/sh
# Applies the MI355X (`hip` build tag) integration to a checkout of vocdoni/davinci-node.
#   usage: apply_hip_patch.sh /path/to/davinci-node
# What it does, and why (file:line are the upstream ones):
#  1. prover/prover_cpu.go:1 and prover/prover_cpu_test.go: the CPU-only build excludes the hip build as it excludes icicle.
#  2. prover/setup.go:24-27: the gpugroth16.Setup branch moves behind the icicle tag (setup_icicle.go / setup_noicicle.go);
#     the file stops importing gnark's icicle packages in every build.
#  3. circuits/artifacts.go:21,624-646: newConstraintSystem / newProvingKey / newVerifyingKey (the file's last three
#     functions) move to artifacts_icicle.go / artifacts_hip.go; the un-gated file drops the icicle import.
#  4. types/blobs.go:88-133: the four geth-backed blob KZG methods (ComputeCommitment, ComputeCellProofs, ComputeBlobProof,
#     ComputeProof) move to types/blobs_geth.go (`!hip`); types/blobs_hip.go defines them over the dvn_kzg_* entry points.
#  5. circuits/artifacts.go:330-352 (LoadOrDownload): one optional call, preloadDeviceKey, between the proving key and the
#     verifying key -- a no-op unless DVN_LOAD_KEY_FILES=1 under the hip tag.
#  6. the new files are copied in.
set -eu
dst=${1:?path to a davinci-node checkout}
src=$(cd "$(dirname "$0")" && pwd)

sed -i '1s|^//go:build !icicle$|//go:build !icicle \&\& !hip|' "$dst/prover/prover_cpu.go"
grep -q '^//go:build' "$dst/prover/prover_cpu_test.go" || sed -i '1i //go:build !icicle \&\& !hip\n' "$dst/prover/prover_cpu_test.go"

sed -i '/gpugroth16 "github.com\/consensys\/gnark\/backend\/accelerated\/icicle\/groth16"/d' "$dst/prover/setup.go" "$dst/circuits/artifacts.go"
# setup.go: everything from the `if UseGPUProver {` branch to the end of Setup becomes one call
sed -i '/^\tif UseGPUProver {$/,/^\treturn groth16.Setup(ccs)$/c\	return backendSetup(ccs)' "$dst/prover/setup.go"
# artifacts.go: the three constructors are the tail of the file
sed -i '/^func newConstraintSystem(curve ecc.ID)/,$d' "$dst/circuits/artifacts.go"

# blobs.go: drop everything from the first of the four methods up to (not including) ComputeCommitmentAndProof
sed -i '/^\/\/ ComputeCommitment creates a small commitment out of a data blob\.$/,/^\/\/ ComputeCommitmentAndProof calculates/{/^\/\/ ComputeCommitmentAndProof calculates/!d}' "$dst/types/blobs.go"

# artifacts.go LoadOrDownload (:330-352): once ccs and pk are both in memory, optionally let the library read the key FILE itself
# (DVN_LOAD_KEY_FILES=1: device copy straight from the sha256-named artifact, preloadDeviceKey in artifacts_hip.go)
sed -i '/^\tvk, err := ca.LoadOrDownloadVerifyingKey(ctx)$/i\
\tif path, perr := ca.provingKey.cachePath(); perr == nil {\
\t\tif perr = preloadDeviceKey(ca.curve, ccs, pk, path, hex.EncodeToString(ca.provingKey.Hash)); perr != nil {\
\t\t\tlog.Warnw("device key preload failed, the first proof will import the key instead", "circuit", ca.name, "error", perr)\
\t\t}\
\t}' "$dst/circuits/artifacts.go"

cp "$src"/prover/*.go "$dst/prover/"
cp "$src"/types/*.go "$dst/types/"
cp "$src"/circuits/*.go "$dst/circuits/"
echo "applied: build with  go build -tags=hip ./...  (libdvn_prover.so + dvn_prover.h under third_party/dvn/{lib,include})"
