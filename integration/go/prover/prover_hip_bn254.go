//go:build hip && !icicle

package prover

// gnark's groth16.Prove for BN254 (backend/groth16/bn254/prove.go) with everything after the R1CS solver on the MI355X:
// the solver (hints, emulated arithmetic) stays gnark's, its BSB22 commitment hint calls dvn_pedersen_commit, and
// computeH + the five MultiExps + blinding are one call, dvn_groth16_prove_solved.  The statetransition proof runs
// through here (/root/reference/sequencer/statetransition.go:248).

/*
#include "dvn_prover.h"
*/
import "C"

import (
	"fmt"
	"math/big"
	"math/bits"
	"runtime"
	"sync"
	"unsafe"

	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark/backend"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
	cs "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/constraint/solver"
	fcs "github.com/consensys/gnark/frontend/cs"
)

// deviceKeyBN254 is the device-resident image of one proving key: one handle per device of the context (a proof is spread
// over all of them), plus the commitment bases on the first device for the solver's hint.
type deviceKeyBN254 struct {
	keys            []*C.dvn_groth16_pk
	basis, basisSig []*C.dvn_points // per commitment key, on device 0
}

var (
	pkMuBN254  sync.Mutex
	pkDevBN254 = map[*groth16_bn254.ProvingKey]*deviceKeyBN254{} // side table keyed by the PK pointer
)

func g1PtrBN254(p []curve.G1Affine) *C.uint64_t {
	if len(p) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&p[0])) // gnark's affine memory image (Montgomery limbs): zero-copy
}

func g2PtrBN254(p []curve.G2Affine) *C.uint64_t {
	if len(p) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&p[0]))
}

func frPtrBN254(v []fr.Element) *C.uint64_t {
	if len(v) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&v[0]))
}

func u32Ptr(v []uint32) *C.uint32_t {
	if len(v) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&v[0]))
}

// importKeyBN254 hands the key's slices to the library once (copied to HBM before the call returns: no Go pointer is kept).
// idxA / idxB: the wires whose points gnark kept (InfinityA / InfinityB drop the others); idxK: the private wires minus the
// committed ones minus the commitment wires themselves (gnark's filterHeap over wireValues[nbPublic:]).
func importKeyBN254(ctx *C.dvn_ctx, r1cs *cs.R1CS, pk *groth16_bn254.ProvingKey) (*deviceKeyBN254, error) {
	pkMuBN254.Lock()
	defer pkMuBN254.Unlock()
	if d, ok := pkDevBN254[pk]; ok {
		return d, nil
	}
	nbWires := len(pk.InfinityA)
	idxA := make([]uint32, 0, len(pk.G1.A))
	idxB := make([]uint32, 0, len(pk.G1.B))
	for i := 0; i < nbWires; i++ {
		if !pk.InfinityA[i] {
			idxA = append(idxA, uint32(i))
		}
		if !pk.InfinityB[i] {
			idxB = append(idxB, uint32(i))
		}
	}
	commitmentInfo := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
	skip := make(map[int]struct{})
	for _, list := range commitmentInfo.GetPrivateCommitted() {
		for _, w := range list {
			skip[w] = struct{}{}
		}
	}
	for _, w := range commitmentInfo.CommitmentIndexes() {
		skip[w] = struct{}{}
	}
	idxK := make([]uint32, 0, len(pk.G1.K))
	for i := r1cs.GetNbPublicVariables(); i < nbWires; i++ {
		if _, drop := skip[i]; !drop {
			idxK = append(idxK, uint32(i))
		}
	}
	if len(idxA) != len(pk.G1.A) || len(idxB) != len(pk.G1.B) || len(idxK) != len(pk.G1.K) {
		return nil, fmt.Errorf("proving key / constraint system mismatch: |A| %d/%d |B| %d/%d |K| %d/%d",
			len(idxA), len(pk.G1.A), len(idxB), len(pk.G1.B), len(idxK), len(pk.G1.K))
	}
	// the descriptor below is Go memory that holds Go pointers: every array it names is pinned for the duration of the calls
	// (cgo pointer rules; runtime.Pinner, Go >= 1.21)
	var pin runtime.Pinner
	defer pin.Unpin()
	for _, p := range []unsafe.Pointer{unsafe.Pointer(g1PtrBN254(pk.G1.A)), unsafe.Pointer(g1PtrBN254(pk.G1.B)), unsafe.Pointer(g1PtrBN254(pk.G1.K)),
		unsafe.Pointer(g1PtrBN254(pk.G1.Z)), unsafe.Pointer(g2PtrBN254(pk.G2.B)), unsafe.Pointer(u32Ptr(idxA)), unsafe.Pointer(u32Ptr(idxB)),
		unsafe.Pointer(u32Ptr(idxK))} {
		if p != nil {
			pin.Pin(p)
		}
	}
	pin.Pin(&pk.G1.Alpha)
	pin.Pin(&pk.G1.Beta)
	pin.Pin(&pk.G1.Delta)
	pin.Pin(&pk.G2.Beta)
	pin.Pin(&pk.G2.Delta)
	ka := C.dvn_groth16_key_arrays{
		log_n:   C.uint32_t(bits.TrailingZeros64(pk.Domain.Cardinality)),
		n_wires: C.uint64_t(nbWires),
		A:       g1PtrBN254(pk.G1.A), B: g1PtrBN254(pk.G1.B), K: g1PtrBN254(pk.G1.K), Z: g1PtrBN254(pk.G1.Z), B2: g2PtrBN254(pk.G2.B),
		nA: C.uint64_t(len(pk.G1.A)), nB: C.uint64_t(len(pk.G1.B)), nK: C.uint64_t(len(pk.G1.K)), nZ: C.uint64_t(len(pk.G1.Z)),
		idxA: u32Ptr(idxA), idxB: u32Ptr(idxB), idxK: u32Ptr(idxK),
		alpha_g1: (*C.uint64_t)(unsafe.Pointer(&pk.G1.Alpha)), beta_g1: (*C.uint64_t)(unsafe.Pointer(&pk.G1.Beta)),
		delta_g1: (*C.uint64_t)(unsafe.Pointer(&pk.G1.Delta)),
		beta_g2:  (*C.uint64_t)(unsafe.Pointer(&pk.G2.Beta)), delta_g2: (*C.uint64_t)(unsafe.Pointer(&pk.G2.Delta)),
	}
	d := &deviceKeyBN254{}
	ndev := int(C.dvn_ctx_device_count(ctx))
	for i := 0; i < ndev; i++ {
		sub := C.dvn_ctx_device(ctx, C.int32_t(i))
		var h *C.dvn_groth16_pk
		if err := hipCall(sub, func() C.int32_t { return C.dvn_groth16_pk_import(sub, &ka, &h) }); err != nil {
			d.release(ctx) // the keys already resident on the earlier devices would otherwise stay in HBM for good
			return nil, fmt.Errorf("dvn_groth16_pk_import (device %d): %w", i, err)
		}
		d.keys = append(d.keys, h)
	}
	for i := range pk.CommitmentKeys {
		var b, s *C.dvn_points
		ck := &pk.CommitmentKeys[i]
		if err := hipCall(ctx, func() C.int32_t {
			return C.dvn_points_upload_g1(ctx, g1PtrBN254(ck.Basis), C.size_t(len(ck.Basis)), &b)
		}); err != nil {
			d.release(ctx)
			return nil, err
		}
		d.basis = append(d.basis, b)
		if err := hipCall(ctx, func() C.int32_t {
			return C.dvn_points_upload_g1(ctx, g1PtrBN254(ck.BasisExpSigma), C.size_t(len(ck.BasisExpSigma)), &s)
		}); err != nil {
			d.release(ctx)
			return nil, err
		}
		d.basisSig = append(d.basisSig, s)
	}
	pkDevBN254[pk] = d
	return d, nil
}

// release frees whatever of the key is resident: the per-device handles (each on its own sub-context) and the commitment
// bases on device 0.  Used on the error paths of importKeyBN254; a successfully imported key lives as long as the process.
func (d *deviceKeyBN254) release(ctx *C.dvn_ctx) {
	for i, h := range d.keys {
		if h != nil {
			C.dvn_groth16_pk_free(C.dvn_ctx_device(ctx, C.int32_t(i)), h)
		}
	}
	for _, p := range append(d.basis, d.basisSig...) {
		if p != nil {
			C.dvn_points_free(ctx, p)
		}
	}
	d.keys, d.basis, d.basisSig = nil, nil, nil
}

// hipProveBN254 restates gnark's Prove step for step (backend/groth16/bn254/prove.go).
func hipProveBN254(ctx *C.dvn_ctx, ccs constraint.ConstraintSystem, pk *groth16_bn254.ProvingKey, fullWitness witness.Witness,
	opts ...backend.ProverOption) (*groth16_bn254.Proof, error) {
	r1cs, ok := ccs.(*cs.R1CS)
	if !ok {
		return nil, fmt.Errorf("constraint system type mismatch for BN254: expected *cs.R1CS, got %T", ccs)
	}
	opt, err := backend.NewProverConfig(opts...)
	if err != nil {
		return nil, fmt.Errorf("new prover config: %w", err)
	}
	if opt.HashToFieldFn == nil {
		return nil, fmt.Errorf("prover config carries no hash-to-field function")
	}
	dev, err := importKeyBN254(ctx, r1cs, pk)
	if err != nil {
		return nil, err
	}
	commitmentInfo := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
	proof := &groth16_bn254.Proof{Commitments: make([]curve.G1Affine, len(commitmentInfo))}
	poks := make([]curve.G1Affine, len(commitmentInfo))

	// (1) solve, with the BSB22 commitment hint overridden: Commitment and its proof of knowledge come from the device (two MSMs
	// in flight together); the challenge is hashed by gnark's own HashToFieldFn (keccak under the Solidity target option)
	solverOpts := opt.SolverOpts[:len(opt.SolverOpts):len(opt.SolverOpts)]
	bsb22ID := solver.GetHintID(fcs.Bsb22CommitmentComputePlaceholder)
	solverOpts = append(solverOpts, solver.OverrideHint(bsb22ID, func(_ *big.Int, in []*big.Int, out []*big.Int) error {
		i := int(in[0].Int64())
		in = in[1:]
		hashed := in[:len(commitmentInfo[i].PublicAndCommitmentCommitted)]
		committed := in[len(hashed):]
		values := make([]fr.Element, len(commitmentInfo[i].PrivateCommitted))
		for j, inJ := range committed {
			values[j].SetBigInt(inJ)
		}
		if err := hipCall(ctx, func() C.int32_t {
			return C.dvn_pedersen_commit(ctx, dev.basis[i], dev.basisSig[i], frPtrBN254(values), C.size_t(len(values)), 0,
				(*C.uint64_t)(unsafe.Pointer(&proof.Commitments[i])), (*C.uint64_t)(unsafe.Pointer(&poks[i])))
		}); err != nil {
			return err
		}
		opt.HashToFieldFn.Write(constraint.SerializeCommitment(proof.Commitments[i].Marshal(), hashed, (fr.Bits-1)/8+1))
		hashBts := opt.HashToFieldFn.Sum(nil)
		opt.HashToFieldFn.Reset()
		nbBuf := fr.Bytes
		if opt.HashToFieldFn.Size() < fr.Bytes {
			nbBuf = opt.HashToFieldFn.Size()
		}
		var res fr.Element
		res.SetBytes(hashBts[:nbBuf])
		res.BigInt(out[0])
		return nil
	}))
	_solution, err := r1cs.Solve(fullWitness, solverOpts...)
	if err != nil {
		return nil, err
	}
	solution := _solution.(*cs.R1CSSolution)
	wireValues := []fr.Element(solution.W)

	// (2) fold the proofs of knowledge (pedersen.BatchProve: sum_i c^i pok_i; with one commitment the fold is the identity)
	if len(poks) == 1 {
		proof.CommitmentPok = poks[0]
	} else if len(poks) > 1 {
		serialized := make([]byte, fr.Bytes*len(commitmentInfo))
		for i := range commitmentInfo {
			copy(serialized[fr.Bytes*i:], wireValues[commitmentInfo[i].CommitmentIndex].Marshal())
		}
		challenge, err := fr.Hash(serialized, []byte("G16-BSB22"), 1)
		if err != nil {
			return nil, err
		}
		var acc curve.G1Jac
		var ci big.Int
		pow := fr.One()
		for i := range poks {
			var t curve.G1Jac
			t.FromAffine(&poks[i])
			t.ScalarMultiplication(&t, pow.BigInt(&ci))
			acc.AddAssign(&t)
			pow.Mul(&pow, &challenge[0])
		}
		proof.CommitmentPok.FromJacobian(&acc)
	}

	// (5) blinding factors, as gnark draws them
	var r, s fr.Element
	if _, err := r.SetRandom(); err != nil {
		return nil, err
	}
	if _, err := s.SetRandom(); err != nil {
		return nil, err
	}

	// (3)(4)(6)(7) computeH, wire filtering, the five MultiExps, blinding and assembly: one call.  Ar | Bs | Krs come back as
	// G1Affine | G2Affine | G1Affine memory images, i.e. exactly the first three fields of the Proof struct.
	var out struct {
		Ar  curve.G1Affine
		Bs  curve.G2Affine
		Krs curve.G1Affine
	}
	n := C.size_t(len(solution.A))
	if len(solution.B) != len(solution.A) || len(solution.C) != len(solution.A) {
		return nil, fmt.Errorf("solver output vectors differ in length")
	}
	if len(dev.keys) == 1 {
		err = hipCall(ctx, func() C.int32_t {
			return C.dvn_groth16_prove_solved(ctx, dev.keys[0], frPtrBN254(wireValues), frPtrBN254(solution.A), frPtrBN254(solution.B),
				frPtrBN254(solution.C), n, (*C.uint64_t)(unsafe.Pointer(&r)), (*C.uint64_t)(unsafe.Pointer(&s)), (*C.uint64_t)(unsafe.Pointer(&out)))
		})
	} else {
		err = hipCall(ctx, func() C.int32_t {
			return C.dvn_groth16_prove_solved_multi(ctx, &dev.keys[0], C.uint32_t(len(dev.keys)), frPtrBN254(wireValues), frPtrBN254(solution.A),
				frPtrBN254(solution.B), frPtrBN254(solution.C), n, (*C.uint64_t)(unsafe.Pointer(&r)), (*C.uint64_t)(unsafe.Pointer(&s)),
				(*C.uint64_t)(unsafe.Pointer(&out)))
		})
	}
	if err != nil {
		return nil, err
	}
	proof.Ar, proof.Bs, proof.Krs = out.Ar, out.Bs, out.Krs
	return proof, nil
}
