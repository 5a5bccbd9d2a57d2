//go:build icicle

package circuits

// Backend-typed constructors for the icicle build: with the GPU prover enabled, keys and constraint systems are read into
// gnark's ICICLE types; the un-gated circuits/artifacts.go no longer imports the icicle packages (apply_hip_patch.sh).

import (
	"github.com/consensys/gnark-crypto/ecc"
	gpugroth16 "github.com/consensys/gnark/backend/accelerated/icicle/groth16"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/constraint"
	"github.com/vocdoni/davinci-node/prover"
)

// pick returns the ICICLE-typed object when the GPU prover is on, the plain gnark one otherwise.
func pick[T any](curve ecc.ID, gpu, plain func(ecc.ID) T) T {
	if prover.UseGPUProver {
		return gpu(curve)
	}
	return plain(curve)
}

func newConstraintSystem(curve ecc.ID) constraint.ConstraintSystem {
	return pick(curve, gpugroth16.NewCS, groth16.NewCS)
}

func newProvingKey(curve ecc.ID) groth16.ProvingKey {
	return pick(curve, gpugroth16.NewProvingKey, groth16.NewProvingKey)
}

func newVerifyingKey(curve ecc.ID) groth16.VerifyingKey {
	return pick(curve, gpugroth16.NewVerifyingKey, groth16.NewVerifyingKey)
}

// preloadDeviceKey: the hip build's optional file preload (artifacts_hip.go) has no counterpart under the icicle tag.
func preloadDeviceKey(ecc.ID, constraint.ConstraintSystem, groth16.ProvingKey, string, string) error { return nil }
