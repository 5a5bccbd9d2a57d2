//go:build !icicle

package prover

import (
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/constraint"
)

// backendSetup for the CPU build and for the MI355X build (`hip`): both work on gnark's plain keys.  The device-resident
// copy of a key is made on first use by the prover (prover_hip_<curve>.go), not at Setup.
func backendSetup(ccs constraint.ConstraintSystem) (groth16.ProvingKey, groth16.VerifyingKey, error) {
	return groth16.Setup(ccs)
}
