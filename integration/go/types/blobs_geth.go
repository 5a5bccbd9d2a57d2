//go:build !hip

package types

// The geth-backed blob KZG methods, as they stand in /root/reference/types/blobs.go:88-133, behind `!hip`: with the hip tag
// blobs_hip.go defines the same four methods over libdvn_prover.so, and apply_hip_patch.sh removes them from blobs.go so
// that each is defined exactly once per build.

import (
	"fmt"
	"math/big"

	gethkzg "github.com/ethereum/go-ethereum/crypto/kzg4844"
)

// ComputeCommitment creates a small commitment out of a data blob.
func (b *Blob) ComputeCommitment() (KZGCommitment, error) {
	commitment, err := gethkzg.BlobToCommitment(b.AsGeth())
	if err != nil {
		return KZGCommitment{}, err
	}
	return KZGCommitment(commitment), nil
}

// ComputeCellProofs returns the cell proofs of a data blob.
func (b *Blob) ComputeCellProofs() ([]KZGProof, error) {
	proofs, err := gethkzg.ComputeCellProofs(b.AsGeth())
	if err != nil {
		return nil, err
	}
	return SliceOf(proofs, KZGProofFromGeth), nil
}

// ComputeBlobProof returns the KZG proof that is used to verify the blob against the commitment.
func (b *Blob) ComputeBlobProof(commitment KZGCommitment) (KZGProof, error) {
	proof, err := gethkzg.ComputeBlobProof(b.AsGeth(), *commitment.AsGeth())
	if err != nil {
		return KZGProof{}, err
	}
	return KZGProof(proof), nil
}

// ComputeProof computes the KZG proof at the given point for the polynomial represented by the blob.
func (b *Blob) ComputeProof(point *big.Int) (proof KZGProof, claim *big.Int, err error) {
	var gethPoint gethkzg.Point
	if point.BitLen() > len(gethPoint)*8 {
		return KZGProof{}, nil, fmt.Errorf("point does not fit in %d bytes", len(gethPoint))
	}
	point.FillBytes(gethPoint[:])
	gethProof, gethClaim, err := gethkzg.ComputeProof(b.AsGeth(), gethPoint)
	if err != nil {
		return KZGProof{}, nil, err
	}
	return KZGProof(gethProof), new(big.Int).SetBytes(gethClaim[:]), nil
}
