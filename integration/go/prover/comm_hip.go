//go:build hip && !icicle

package prover

// One prover PROCESS per GPU (instead of one process driving all GPUs through DVN_DEVICES): the ranks run the solver and the
// commitment step replicated, compute their share of the five MSMs (dvn_groth16_prove_solved_shard), exchange the 72-word
// partial results with ONE all-gather over RCCL / xGMI (dvn_comm_all_gather) and assemble the same proof
// (dvn_groth16_prove_combine).  The reference has no counterpart: its ICICLE backend proves on one GPU
// (prover/prover_gpu.go:33); the serialisation point this shortens is sequencer/statetransition.go:80.

/*
#include <stdlib.h>
#include "dvn_prover.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// Comm is an RCCL communicator of `world` prover processes; rank 0 creates the id (CommUniqueID) and hands its 128 bytes to
// the others by any means (a file, an environment variable, the sequencer's own RPC) before every rank calls NewComm.
type Comm struct {
	h           *C.dvn_comm
	Rank, World uint32
}

func CommUniqueID() ([128]byte, error) {
	var id [128]byte
	err := hipCall(nil, func() C.int32_t { return C.dvn_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))) })
	return id, err
}

// NewComm is a collective: it returns when all `world` ranks have called it with the same id.
func NewComm(id [128]byte, rank, world uint32) (*Comm, error) {
	ctx, err := hipContext()
	if err != nil {
		return nil, err
	}
	c := &Comm{Rank: rank, World: world}
	err = hipCall(ctx, func() C.int32_t {
		return C.dvn_comm_init(ctx, (*C.uint8_t)(unsafe.Pointer(&id[0])), C.uint32_t(rank), C.uint32_t(world), &c.h)
	})
	if err != nil {
		return nil, err
	}
	return c, nil
}

// AllGather returns the rows of all ranks, rank r's at [r*len(local), (r+1)*len(local)).
func (c *Comm) AllGather(local []uint64) ([]uint64, error) {
	ctx, err := hipContext()
	if err != nil {
		return nil, err
	}
	if len(local) == 0 || len(local) > int(C.DVN_COMM_MAX_WORDS) {
		return nil, fmt.Errorf("all-gather of %d words: between 1 and %d per rank", len(local), int(C.DVN_COMM_MAX_WORDS))
	}
	all := make([]uint64, int(c.World)*len(local))
	err = hipCall(ctx, func() C.int32_t {
		return C.dvn_comm_all_gather(ctx, c.h, (*C.uint64_t)(unsafe.Pointer(&local[0])), C.size_t(len(local)), (*C.uint64_t)(unsafe.Pointer(&all[0])))
	})
	return all, err
}

func (c *Comm) Close() error {
	ctx, err := hipContext()
	if err != nil {
		return err
	}
	h := c.h
	c.h = nil
	return hipCall(ctx, func() C.int32_t { return C.dvn_comm_free(ctx, h) })
}

// proveSolvedSharedBN254 is the last step of hipProveBN254 for a process that is rank comm.Rank of comm.World: the device
// key, the wire values and the solver's a, b, c are the same on every rank.  The blinding factors must be too, and they are
// random: every rank sends its own draw along with its partial results and all of them use rank 0's.
func proveSolvedSharedBN254(ctx *C.dvn_ctx, dev *deviceKeyBN254, comm *Comm, wireValues, a, b, c []fr.Element, r, s fr.Element) (
	ar curve.G1Affine, bs curve.G2Affine, krs curve.G1Affine, err error) {
	const partialWords = 72
	row := make([]uint64, partialWords+8)
	err = hipCall(ctx, func() C.int32_t {
		return C.dvn_groth16_prove_solved_shard(ctx, dev.keys[0], frPtrBN254(wireValues), frPtrBN254(a), frPtrBN254(b), frPtrBN254(c),
			C.size_t(len(a)), C.uint32_t(comm.Rank), C.uint32_t(comm.World), (*C.uint64_t)(unsafe.Pointer(&row[0])))
	})
	if err != nil {
		return
	}
	copy(row[partialWords:partialWords+4], r[:])
	copy(row[partialWords+4:], s[:])
	all, err := comm.AllGather(row)
	if err != nil {
		return
	}
	partials := make([]uint64, int(comm.World)*partialWords)
	for k := 0; k < int(comm.World); k++ {
		copy(partials[k*partialWords:(k+1)*partialWords], all[k*len(row):k*len(row)+partialWords])
	}
	var r0, s0 fr.Element
	copy(r0[:], all[partialWords:partialWords+4])
	copy(s0[:], all[partialWords+4:partialWords+8])
	var out struct {
		Ar  curve.G1Affine
		Bs  curve.G2Affine
		Krs curve.G1Affine
	}
	err = hipCall(ctx, func() C.int32_t {
		return C.dvn_groth16_prove_combine(dev.keys[0], (*C.uint64_t)(unsafe.Pointer(&partials[0])), C.uint32_t(comm.World),
			(*C.uint64_t)(unsafe.Pointer(&r0)), (*C.uint64_t)(unsafe.Pointer(&s0)), (*C.uint64_t)(unsafe.Pointer(&out)))
	})
	return out.Ar, out.Bs, out.Krs, err
}
