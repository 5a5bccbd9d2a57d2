//go:build hip && !icicle

package prover

// Device keys straight from the artifact FILE.  The node keeps every proving key as a sha256-named file
// (/root/reference/circuits/artifacts.go:62-70) and decodes it with pk.UnsafeReadFrom (:391-406); importKey<Curve> then hands
// the decoded Go slices to the library.  PreloadProvingKeyFile is the other way in: the library reads the same file itself
// (mmap, content-hash check, point decompression on the device, resident cache keyed by the hash) and the handle is entered into
// the side table under the gnark key's pointer, so the first proof finds its device key ready and no second copy of the
// point arrays crosses cgo.  gnark's own key stays what the CPU fallback uses.
// Optional: circuits/artifacts_hip.go calls it when DVN_LOAD_KEY_FILES=1; any error just means the first proof imports the
// key from the Go slices as before.
// NOT COMPILED in the environment this file was written in (no Go toolchain).

/*
#include <stdlib.h>
#include "dvn_prover.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	groth16_bls12377 "github.com/consensys/gnark/backend/groth16/bls12-377"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	groth16_bw6761 "github.com/consensys/gnark/backend/groth16/bw6-761"
	"github.com/consensys/gnark/constraint"
)

// committedWires lists what the key file does not carry: the private wires whose K point gnark's Setup left out of pk.G1.K --
// the PrivateCommitted indices of every commitment together with the commitment wires themselves.
func committedWires(ccs constraint.ConstraintSystem) ([]C.uint32_t, error) {
	type withInfo interface{ GetCommitments() constraint.Commitments }
	holder, ok := ccs.(withInfo)
	if !ok {
		return nil, fmt.Errorf("constraint system %T exposes no commitment info", ccs)
	}
	info, ok := holder.GetCommitments().(constraint.Groth16Commitments)
	if !ok {
		return nil, nil
	}
	var out []C.uint32_t
	for _, list := range info.GetPrivateCommitted() {
		for _, w := range list {
			out = append(out, C.uint32_t(w))
		}
	}
	for _, w := range info.CommitmentIndexes() {
		out = append(out, C.uint32_t(w))
	}
	return out, nil
}

// PreloadProvingKeyFile loads the device-resident copy of pk from its artifact file on the first device of the context.
// sha256hex may be empty (no content check).  Multi-device contexts keep importing through the Go slices: a file load
// serves one device.
func PreloadProvingKeyFile(curve ecc.ID, ccs constraint.ConstraintSystem, pk groth16.ProvingKey, path, sha256hex string) error {
	ctx, err := hipContext()
	if err != nil {
		return err
	}
	if int(C.dvn_ctx_device_count(ctx)) != 1 {
		return fmt.Errorf("file preload serves single-device contexts")
	}
	cw, err := committedWires(ccs)
	if err != nil {
		return err
	}
	var cwp *C.uint32_t
	if len(cw) > 0 {
		cwp = &cw[0]
	}
	cpath := C.CString(path)
	defer C.free(unsafe.Pointer(cpath))
	var chash *C.char
	if sha256hex != "" {
		chash = C.CString(sha256hex)
		defer C.free(unsafe.Pointer(chash))
	}
	nPublic := C.uint32_t(ccs.GetNbPublicVariables())
	var h *C.dvn_groth16_pk
	bases := func() (b, s *C.dvn_points, err error) {
		err = hipCall(ctx, func() C.int32_t { return C.dvn_groth16_pk_commitment_bases(h, &b, &s) })
		return
	}
	switch curve {
	case ecc.BN254:
		key, ok := pk.(*groth16_bn254.ProvingKey)
		if !ok {
			return fmt.Errorf("proving key type mismatch for BN254: expected *groth16_bn254.ProvingKey, got %T", pk)
		}
		pkMuBN254.Lock()
		defer pkMuBN254.Unlock()
		if _, done := pkDevBN254[key]; done {
			return nil
		}
		if err := hipCall(ctx, func() C.int32_t {
			return C.dvn_groth16_pk_load_file(ctx, cpath, chash, nPublic, cwp, C.uint32_t(len(cw)), &h)
		}); err != nil {
			return err
		}
		d := &deviceKeyBN254{keys: []*C.dvn_groth16_pk{h}}
		if b, s, err := bases(); err == nil && b != nil {
			d.basis, d.basisSig = []*C.dvn_points{b}, []*C.dvn_points{s}
		}
		pkDevBN254[key] = d
	case ecc.BLS12_377:
		key, ok := pk.(*groth16_bls12377.ProvingKey)
		if !ok {
			return fmt.Errorf("proving key type mismatch for BLS12_377: expected *groth16_bls12377.ProvingKey, got %T", pk)
		}
		pkMuBLS12377.Lock()
		defer pkMuBLS12377.Unlock()
		if _, done := pkDevBLS12377[key]; done {
			return nil
		}
		if err := hipCall(ctx, func() C.int32_t {
			return C.dvn_bls12377_groth16_pk_load_file(ctx, cpath, chash, nPublic, cwp, C.uint32_t(len(cw)), &h)
		}); err != nil {
			return err
		}
		d := &deviceKeyBLS12377{keys: []*C.dvn_groth16_pk{h}}
		if b, s, err := bases(); err == nil && b != nil {
			d.basis, d.basisSig = []*C.dvn_points{b}, []*C.dvn_points{s}
		}
		pkDevBLS12377[key] = d
	case ecc.BW6_761:
		key, ok := pk.(*groth16_bw6761.ProvingKey)
		if !ok {
			return fmt.Errorf("proving key type mismatch for BW6_761: expected *groth16_bw6761.ProvingKey, got %T", pk)
		}
		pkMuBW6761.Lock()
		defer pkMuBW6761.Unlock()
		if _, done := pkDevBW6761[key]; done {
			return nil
		}
		if err := hipCall(ctx, func() C.int32_t {
			return C.dvn_bw6761_groth16_pk_load_file(ctx, cpath, chash, nPublic, cwp, C.uint32_t(len(cw)), &h)
		}); err != nil {
			return err
		}
		d := &deviceKeyBW6761{keys: []*C.dvn_groth16_pk{h}}
		if b, s, err := bases(); err == nil && b != nil {
			d.basis, d.basisSig = []*C.dvn_points{b}, []*C.dvn_points{s}
		}
		pkDevBW6761[key] = d
	default:
		return fmt.Errorf("no artifact loader for curve %s", curve.String())
	}
	return nil
}
