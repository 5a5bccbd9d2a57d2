//go:build !hip && !icicle

package circuits

import (
	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/constraint"
)

func preloadDeviceKeyImpl(ecc.ID, constraint.ConstraintSystem, groth16.ProvingKey, string, string) error { return nil }
