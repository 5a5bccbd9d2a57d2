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
