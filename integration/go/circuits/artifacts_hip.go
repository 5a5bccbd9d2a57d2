//go:build !icicle

package circuits

// The three constructors of /root/reference/circuits/artifacts.go:624-646 for the CPU build and the MI355X build (`hip`):
// plain gnark objects whatever UseGPUProver says.  The MI355X prover works on gnark's own *groth16_<curve>.ProvingKey (read
// with UnsafeReadFrom as today, artifacts.go:391-406) and keeps its device-resident copy in a side table, so a
// GPU_PROVER=true run must never see an ICICLE-typed key here.

import (
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/constraint"
)

func newConstraintSystem(curve ecc.ID) constraint.ConstraintSystem { return groth16.NewCS(curve) }

func newProvingKey(curve ecc.ID) groth16.ProvingKey { return groth16.NewProvingKey(curve) }

func newVerifyingKey(curve ecc.ID) groth16.VerifyingKey { return groth16.NewVerifyingKey(curve) }

// preloadDeviceKey is the optional second half of LoadOrDownloadProvingKey under the hip tag (apply_hip_patch.sh adds the one
// call after pk.UnsafeReadFrom, /root/reference/circuits/artifacts.go:401): with DVN_LOAD_KEY_FILES=1 the library reads the
// same sha256-named file itself and keeps the device copy resident (prover.PreloadProvingKeyFile); a failure is logged by the
// caller and ignored -- the first proof then imports the key from the decoded slices, as without the variable.
func preloadDeviceKey(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, path, sha256hex string) error {
	return preloadDeviceKeyImpl(curve, ccs, pk, path, sha256hex)
}
