//go:build hip && !icicle

// Package prover, MI355X backend: the third alternative next to prover_cpu.go (`!icicle && !hip`) and prover_gpu.go
// (`icicle`).  Same exported surface and control flow as prover_gpu.go (Prove / defaultProver / CPUProver / GPUProver /
// ProveWithWitness / CPUProverWithWitness / GPUProverWithWitness, GPU error -> message -> CPU fallback); behind
// callGPUProver sits libdvn_prover.so (include/dvn_prover.h) instead of gnark's icicle backend.
//
// Proving keys stay gnark's plain types (*groth16_bn254.ProvingKey ...): the CPU fallback needs no unwrapping, and
// circuits/artifacts.go constructs them with groth16.NewProvingKey (see circuits/artifacts_hip.go).  Device-resident copies
// live in a side table keyed by the key pointer (prover_hip_<curve>.go).
package prover

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/dvn/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/dvn/lib -ldvn_prover -Wl,-rpath,${SRCDIR}/../third_party/dvn/lib
#include <stdlib.h>
#include "dvn_prover.h"
*/
import "C"

import (
	"fmt"
	"os"
	"runtime"
	"strconv"
	"strings"
	"sync"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend"
	"github.com/consensys/gnark/backend/groth16"
	groth16_bls12377 "github.com/consensys/gnark/backend/groth16/bls12-377"
	groth16_bls12381 "github.com/consensys/gnark/backend/groth16/bls12-381"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	groth16_bw6761 "github.com/consensys/gnark/backend/groth16/bw6-761"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
	"github.com/consensys/gnark/frontend"
)

var (
	hipOnce sync.Once
	hipCtx  *C.dvn_ctx
	hipErr  error
)

// hipContext brings the library up once per process.  DVN_DEVICES ("0,1,2,3") lists the HIP ordinals to use; with more than
// one, a statetransition / aggregator proof is spread over all of them (dvn_groth16_prove_solved_multi).
func hipContext() (*C.dvn_ctx, error) {
	hipOnce.Do(func() {
		var devs []C.int
		for _, f := range strings.Split(os.Getenv("DVN_DEVICES"), ",") {
			if f = strings.TrimSpace(f); f == "" {
				continue
			}
			d, err := strconv.Atoi(f)
			if err != nil {
				hipErr = fmt.Errorf("DVN_DEVICES: %w", err)
				return
			}
			devs = append(devs, C.int(d))
		}
		hipErr = hipCall(nil, func() C.int32_t {
			if len(devs) == 0 {
				return C.dvn_init(nil, 0, &hipCtx)
			}
			return C.dvn_init(&devs[0], C.int(len(devs)), &hipCtx)
		})
	})
	return hipCtx, hipErr
}

// hipCall runs one library call and turns a failure into an error.  The message of dvn_last_error is thread-local in the
// library and a goroutine may migrate between OS threads from one cgo call to the next, so the goroutine is pinned to its
// thread across the call AND the message read.
func hipCall(ctx *C.dvn_ctx, f func() C.int32_t) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	rc := f()
	if rc == 0 {
		return nil
	}
	msg := C.GoString(C.dvn_last_error(ctx))
	if msg == "" {
		msg = C.GoString(C.dvn_error_string(rc))
	}
	return fmt.Errorf("dvn error %d: %s", int(rc), msg)
}

// callGPUProver mirrors prover_gpu.go:24-61: per-curve type assertion of the proving key, then the curve's device prover.
func callGPUProver(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, w witness.Witness, opts []backend.ProverOption,
) (groth16.Proof, error) {
	ctx, err := hipContext()
	if err != nil {
		return nil, err
	}
	switch curve {
	case ecc.BN254:
		bn254Pk, ok := pk.(*groth16_bn254.ProvingKey)
		if !ok {
			return nil, fmt.Errorf("proving key type mismatch for BN254: expected *groth16_bn254.ProvingKey, got %T", pk)
		}
		return hipProveBN254(ctx, ccs, bn254Pk, w, opts...)

	case ecc.BLS12_377:
		bls12377Pk, ok := pk.(*groth16_bls12377.ProvingKey)
		if !ok {
			return nil, fmt.Errorf("proving key type mismatch for BLS12_377: expected *groth16_bls12377.ProvingKey, got %T", pk)
		}
		// prover_gpu.go:41 pins this hot key to the GPU (PinToGPU): device keys are always resident here
		return hipProveBLS12377(ctx, ccs, bls12377Pk, w, opts...)

	case ecc.BLS12_381:
		bls12381Pk, ok := pk.(*groth16_bls12381.ProvingKey)
		if !ok {
			return nil, fmt.Errorf("proving key type mismatch for BLS12_381: expected *groth16_bls12381.ProvingKey, got %T", pk)
		}
		return hipProveBLS12381(ctx, ccs, bls12381Pk, w, opts...)

	case ecc.BW6_761:
		bw6761Pk, ok := pk.(*groth16_bw6761.ProvingKey)
		if !ok {
			return nil, fmt.Errorf("proving key type mismatch for BW6_761: expected *groth16_bw6761.ProvingKey, got %T", pk)
		}
		return hipProveBW6761(ctx, ccs, bw6761Pk, w, opts...)

	default:
		return nil, fmt.Errorf("GPU proving not supported for curve %s", curve)
	}
}

// Prove runs the groth16.Prove algorithm.  With UseGPUProver it tries the MI355X prover first and falls back to the CPU
// prover on any error.
func Prove(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, assignment frontend.Circuit, opts ...backend.ProverOption,
) (groth16.Proof, error) {
	return prover(curve, ccs, pk, assignment, opts...)
}

func defaultProver(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, assignment frontend.Circuit, opts ...backend.ProverOption,
) (groth16.Proof, error) {
	if UseGPUProver {
		proof, err := GPUProver(curve, ccs, pk, assignment, opts...)
		if err != nil {
			// GPU proving failed, fall back to CPU
			fmt.Printf("GPU proving failed (%v), falling back to CPU\n", err)
			return CPUProver(curve, ccs, pk, assignment, opts...)
		}
		return proof, nil
	}
	return CPUProver(curve, ccs, pk, assignment, opts...)
}

// CPUProver is the standard implementation that simply calls groth16.Prove directly.
func CPUProver(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, assignment frontend.Circuit, opts ...backend.ProverOption,
) (groth16.Proof, error) {
	w, err := frontend.NewWitness(assignment, curve.ScalarField())
	if err != nil {
		return nil, fmt.Errorf("failed to create witness: %w", err)
	}
	return groth16.Prove(ccs, pk, w, opts...) // the key already is the plain gnark type: no cpuReadyProvingKey
}

// GPUProver is an implementation that uses GPU acceleration for proving.
func GPUProver(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, assignment frontend.Circuit, opts ...backend.ProverOption,
) (groth16.Proof, error) {
	w, err := frontend.NewWitness(assignment, curve.ScalarField())
	if err != nil {
		return nil, fmt.Errorf("failed to create witness: %w", err)
	}
	return callGPUProver(curve, ccs, pk, w, opts)
}

// ProveWithWitness generates a proof from an already-created witness, GPU first when enabled, CPU on any GPU error.
func ProveWithWitness(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, w witness.Witness, opts ...backend.ProverOption,
) (groth16.Proof, error) {
	if UseGPUProver {
		proof, err := GPUProverWithWitness(curve, ccs, pk, w, opts...)
		if err != nil {
			fmt.Printf("GPU proving failed (%v), falling back to CPU\n", err)
			return CPUProverWithWitness(curve, ccs, pk, w, opts...)
		}
		return proof, nil
	}
	return CPUProverWithWitness(curve, ccs, pk, w, opts...)
}

// CPUProverWithWitness proves using CPU with an already-created witness.
func CPUProverWithWitness(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, w witness.Witness, opts ...backend.ProverOption,
) (groth16.Proof, error) {
	return groth16.Prove(ccs, pk, w, opts...)
}

// GPUProverWithWitness proves using GPU with an already-created witness.
func GPUProverWithWitness(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, w witness.Witness, opts ...backend.ProverOption,
) (groth16.Proof, error) {
	return callGPUProver(curve, ccs, pk, w, opts)
}
