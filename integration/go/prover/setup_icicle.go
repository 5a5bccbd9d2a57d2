//go:build icicle

package prover

import (
	gpugroth16 "github.com/consensys/gnark/backend/accelerated/icicle/groth16"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/constraint"
)

// backendSetup is what /root/reference/prover/setup.go:24-27 does inline today.
func backendSetup(ccs constraint.ConstraintSystem) (groth16.ProvingKey, groth16.VerifyingKey, error) {
	if UseGPUProver {
		return gpugroth16.Setup(ccs)
	}
	return groth16.Setup(ccs)
}
