//go:build hip && !icicle

package types

// Blob KZG on the MI355X: the four methods of /root/reference/types/blobs.go:88-133 (ComputeCommitment, ComputeCellProofs,
// ComputeBlobProof, ComputeProof) bound to the dvn_kzg_* entry points of libdvn_prover.so instead of geth's kzg4844.
// BlobEvalData.Set calls them before every state-transition proof (/root/reference/crypto/blobs/blob.go:40-90).
// Byte formats are unchanged: a Blob is 4096 x 32-byte big-endian field elements, commitments and proofs are 48-byte
// compressed G1 points, the evaluation point and the claim 32 bytes big endian.  Errors come back as geth's do: a
// non-canonical field element or point is an error, never a panic.
//
// The trusted setup is the one the node already embeds (config.KZGTrustedSetup, /root/reference/config/kzg_setup.go:7-8):
// it is parsed once, its two G1 bases are decompressed on the device and stay resident.
// NOT COMPILED in the environment this file was written in (no Go toolchain): apply_hip_patch.sh moves the geth-backed
// methods of types/blobs.go behind `!hip` (blobs_geth.go) so that each method is defined exactly once per build.

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/dvn/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/dvn/lib -ldvn_prover
#include "dvn_prover.h"
*/
import "C"

import (
	"bufio"
	"bytes"
	"encoding/hex"
	"fmt"
	"math/big"
	"os"
	"runtime"
	"strconv"
	"strings"
	"sync"
	"unsafe"

	"github.com/vocdoni/davinci-node/config"
)

var (
	hipBlobOnce sync.Once
	hipCtx      *C.dvn_ctx       // this package's own context (types must not import prover); device = first of DVN_DEVICES
	kzgSetup    *C.dvn_kzg_setup // both G1 bases of the embedded setup, resident in HBM
	hipBlobErr  error
)

// hipError turns a failed library call into an error.  The message is thread-local in the library: the caller holds the OS
// thread (hipBlobCall) from the failing call to this read.
func hipError(rc C.int32_t) error {
	msg := C.GoString(C.dvn_last_error(hipCtx))
	if msg == "" {
		msg = C.GoString(C.dvn_error_string(rc))
	}
	return fmt.Errorf("dvn error %d: %s", int(rc), msg)
}

func hipBlobCall(f func() C.int32_t) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := f(); rc != 0 {
		return hipError(rc)
	}
	return nil
}

// parseTrustedSetup reads the text format of config/kzg_trusted_setup.txt: the G1 count (4096), the G2 count (65), then the
// G1 Lagrange basis, the G2 monomial points and the G1 monomial basis, one hex point per line.
func parseTrustedSetup(text []byte) (lagrange, monomial []byte, err error) {
	sc := bufio.NewScanner(bytes.NewReader(text))
	sc.Buffer(make([]byte, 1<<16), 1<<20)
	next := func() (string, error) {
		for sc.Scan() {
			if l := strings.TrimSpace(sc.Text()); l != "" {
				return l, nil
			}
		}
		return "", fmt.Errorf("trusted setup: unexpected end of file")
	}
	count := func() (int, error) {
		l, e := next()
		if e != nil {
			return 0, e
		}
		return strconv.Atoi(l)
	}
	n1, err := count()
	if err != nil || n1 != 4096 {
		return nil, nil, fmt.Errorf("trusted setup: expected 4096 G1 points, got %d (%v)", n1, err)
	}
	n2, err := count()
	if err != nil {
		return nil, nil, err
	}
	points := func(n, size int, keep bool) ([]byte, error) {
		var out []byte
		if keep {
			out = make([]byte, 0, n*size)
		}
		for i := 0; i < n; i++ {
			l, e := next()
			if e != nil {
				return nil, e
			}
			if !keep {
				continue
			}
			b, e := hex.DecodeString(l)
			if e != nil || len(b) != size {
				return nil, fmt.Errorf("trusted setup: point %d is not %d hex bytes", i, size)
			}
			out = append(out, b...)
		}
		return out, nil
	}
	if lagrange, err = points(n1, 48, true); err != nil {
		return nil, nil, err
	}
	if _, err = points(n2, 96, false); err != nil { // the G2 powers are the verifier's side
		return nil, nil, err
	}
	if monomial, err = points(n1, 48, true); err != nil {
		return nil, nil, err
	}
	return lagrange, monomial, nil
}

func hipBlobInit() error {
	hipBlobOnce.Do(func() {
		var devs []C.int
		if f := strings.TrimSpace(strings.Split(os.Getenv("DVN_DEVICES"), ",")[0]); f != "" {
			d, err := strconv.Atoi(f)
			if err != nil {
				hipBlobErr = fmt.Errorf("DVN_DEVICES: %w", err)
				return
			}
			devs = append(devs, C.int(d))
		}
		if hipBlobErr = hipBlobCall(func() C.int32_t {
			if len(devs) == 0 {
				return C.dvn_init(nil, 0, &hipCtx)
			}
			return C.dvn_init(&devs[0], 1, &hipCtx)
		}); hipBlobErr != nil {
			return
		}
		lagrange, monomial, err := parseTrustedSetup(config.KZGTrustedSetup)
		if err != nil {
			hipBlobErr = err
			return
		}
		hipBlobErr = hipBlobCall(func() C.int32_t {
			return C.dvn_kzg_setup_load(hipCtx, (*C.uint8_t)(unsafe.Pointer(&lagrange[0])), (*C.uint8_t)(unsafe.Pointer(&monomial[0])), &kzgSetup)
		})
	})
	return hipBlobErr
}

func blobPtr(b *Blob) *C.uint8_t { return (*C.uint8_t)(unsafe.Pointer(&b[0])) }

// ComputeCommitment creates a small commitment out of a data blob.
func (b *Blob) ComputeCommitment() (KZGCommitment, error) {
	if err := hipBlobInit(); err != nil {
		return KZGCommitment{}, err
	}
	var out KZGCommitment
	if err := hipBlobCall(func() C.int32_t {
		return C.dvn_kzg_blob_to_commitment(hipCtx, kzgSetup, blobPtr(b), (*C.uint8_t)(unsafe.Pointer(&out[0])))
	}); err != nil {
		return KZGCommitment{}, err
	}
	return out, nil
}

// ComputeCellProofs returns the 128 EIP-7594 cell proofs of the blob (one batched MSM on the device).
func (b *Blob) ComputeCellProofs() ([]KZGProof, error) {
	if err := hipBlobInit(); err != nil {
		return nil, err
	}
	out := make([]KZGProof, CellProofsPerBlob)
	if err := hipBlobCall(func() C.int32_t {
		return C.dvn_kzg_compute_cell_proofs(hipCtx, kzgSetup, blobPtr(b), (*C.uint8_t)(unsafe.Pointer(&out[0][0])))
	}); err != nil {
		return nil, err
	}
	return out, nil
}

// ComputeBlobProof returns the KZG proof that is used to verify the blob against the commitment.
//
// This method does not verify that the commitment is correct with respect to blob.
func (b *Blob) ComputeBlobProof(commitment KZGCommitment) (KZGProof, error) {
	if err := hipBlobInit(); err != nil {
		return KZGProof{}, err
	}
	var proof KZGProof
	if err := hipBlobCall(func() C.int32_t {
		return C.dvn_kzg_compute_blob_proof(hipCtx, kzgSetup, blobPtr(b), (*C.uint8_t)(unsafe.Pointer(&commitment[0])), (*C.uint8_t)(unsafe.Pointer(&proof[0])))
	}); err != nil {
		return KZGProof{}, err
	}
	return proof, nil
}

// ComputeProof computes the KZG proof at the given point for the polynomial represented by the blob.
//
// If the absolute value of point doesn't fit in [32]byte, returns an error.
func (b *Blob) ComputeProof(point *big.Int) (proof KZGProof, claim *big.Int, err error) {
	var z, y [32]byte
	if point.BitLen() > len(z)*8 {
		return KZGProof{}, nil, fmt.Errorf("point does not fit in %d bytes", len(z))
	}
	if err := hipBlobInit(); err != nil {
		return KZGProof{}, nil, err
	}
	point.FillBytes(z[:])
	if err := hipBlobCall(func() C.int32_t {
		return C.dvn_kzg_compute_proof(hipCtx, kzgSetup, blobPtr(b), (*C.uint8_t)(unsafe.Pointer(&z[0])), (*C.uint8_t)(unsafe.Pointer(&proof[0])), (*C.uint8_t)(unsafe.Pointer(&y[0])))
	}); err != nil {
		return KZGProof{}, nil, err
	}
	return proof, new(big.Int).SetBytes(y[:]), nil
}
