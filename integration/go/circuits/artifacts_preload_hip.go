//go:build hip && !icicle

package circuits

import (
	"os"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	"github.com/consensys/gnark/constraint"

	"github.com/vocdoni/davinci-node/prover"
)

func preloadDeviceKeyImpl(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, path, sha256hex string) error {
	if os.Getenv("DVN_LOAD_KEY_FILES") != "1" || !prover.UseGPUProver || ccs == nil {
		return nil
	}
	return prover.PreloadProvingKeyFile(curve, ccs, pk, path, sha256hex)
}
