/* dvn_prover.h -- C-ABI of the MI355X-native prover backend (libdvn_prover.so).
 *
 * This is the drop-in boundary: the entry points a build-tagged Go file next to
 * /root/reference/prover/prover_gpu.go would bind through cgo in place of the ICICLE
 * calls behind gpugroth16.Prove (prover/prover_gpu.go:33).  Every function is thread-safe, calls
 * hipSetDevice on entry (cgo calls arrive on arbitrary OS threads), never throws or aborts, and never keeps
 * a caller pointer after it returns (cgo pointer rule): host inputs are copied to the device before return
 * and outputs are written into caller buffers.
 *
 * Data formats are gnark-crypto's in-memory images, zero-copy from Go slices:
 *   fr.Element / fp.Element   4 x uint64 little-endian limbs, Montgomery form, R = 2^256
 *   G1Affine {X, Y}           8 x uint64, infinity = (0, 0)
 *   G2Affine {X.A0, X.A1, Y.A0, Y.A1}   16 x uint64
 *   G1Jac {X, Y, Z}           12 x uint64;  G2Jac 24 x uint64
 *
 * Return value: 0 = DVN_OK, negative = error; dvn_last_error(ctx) gives the message of the CALLING THREAD's last failure
 * (thread-local).  A cgo caller must therefore pin its goroutine to the OS thread across the failing call and the
 * dvn_last_error call (runtime.LockOSThread / UnlockOSThread, as the shim in integration/go does); dvn_error_string(code)
 * is the thread-independent fallback text of a code.
 *
 * No C++ exception crosses this boundary: every entry point that returns a code catches whatever is thrown beneath it
 * (std::bad_alloc of a host container, std::system_error of a thread that cannot start) and returns DVN_ERR_OOM, or
 * DVN_ERR_DEVICE for anything else, with the message in dvn_last_error.
 */
#ifndef DVN_PROVER_H
#define DVN_PROVER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVN_OK 0
#define DVN_ERR_INVALID_ARG (-1)   /* NULL pointer, bad range, len(points) != len(scalars) (MultiExp's error) */
#define DVN_ERR_DEVICE (-2)        /* a HIP call failed; message carries hipGetErrorString */
#define DVN_ERR_NO_DEVICE (-3)     /* no usable gfx950 device / requested ordinal missing */
#define DVN_ERR_OOM (-4)
#define DVN_ERR_UNSUPPORTED (-5)   /* e.g. curve other than BN254 (mirrors prover/prover_gpu.go:58-60) */
#define DVN_ERR_INJECTED (-6)      /* DVN_FORCE_FAIL=1 test hook: exercises the caller's GPU->CPU fallback */

typedef struct dvn_ctx dvn_ctx;
typedef struct dvn_points dvn_points;   /* device-resident affine point set (one proving-key array) */
typedef struct dvn_domain dvn_domain;   /* fft.Domain: twiddle tables for one size, device-resident */

/* ---- lifetime ---------------------------------------------------------------------------------------- */
/* Replaces icicle's device/runtime bring-up behind gpugroth16 (prover/prover_gpu.go:10-15).
 * devs/ndev: HIP device ordinals (NULL/0 = device 0).  EVERY listed ordinal gets a context: the handle returned serves
 * devs[0], dvn_ctx_device(ctx, i) is the context of devs[i] (use it for uploads / key imports on that device), and the
 * *_multi entry points spread one proof over all of them.  dvn_shutdown of the primary releases the others.  An ordinal may
 * repeat (two contexts on one GPU), which is how a one-GPU machine rehearses the multi-device path. */
int32_t dvn_init(const int* devs, int ndev, dvn_ctx** out);
void dvn_shutdown(dvn_ctx* ctx);
int32_t dvn_ctx_device_count(const dvn_ctx* ctx);
dvn_ctx* dvn_ctx_device(dvn_ctx* ctx, int32_t index);
int32_t dvn_ctx_device_ordinal(const dvn_ctx* ctx);
const char* dvn_last_error(const dvn_ctx* ctx);
const char* dvn_error_string(int32_t code);
const char* dvn_version(void);
/* Test hook, host only (no context, no device): raises the C++ exception `kind` names inside an entry point so that the
 * exception-to-code mapping of the boundary can be checked.  0 none (DVN_OK), 1 std::bad_alloc, 2 std::system_error (both
 * DVN_ERR_OOM), 3 std::runtime_error, 4 a non-standard exception (both DVN_ERR_DEVICE).  No counterpart in the reference. */
int32_t dvn_test_raise(int32_t kind);
/* Bench only, no counterpart in the reference: the chip-wide v_mad_u64_u32 issue rate of THIS device right now, in lane
 * operations per second (every SIMD saturated with independent multiply-add chains; ~10 ms).  The hot kernels are bound by
 * the issue of that instruction, so bench.py quotes their integer-issue fraction against this in-run figure. */
int32_t dvn_bench_mad_rate(dvn_ctx* ctx, double* lane_ops_per_s);

/* ---- device memory helpers (bench/test harness; also used by the host orchestration) ------------------ */
int32_t dvn_malloc(dvn_ctx* ctx, size_t bytes, void** dptr);
int32_t dvn_free(dvn_ctx* ctx, void* dptr);
int32_t dvn_memcpy_h2d(dvn_ctx* ctx, void* dst, const void* src, size_t bytes);
int32_t dvn_memcpy_d2h(dvn_ctx* ctx, void* dst, const void* src, size_t bytes);
int32_t dvn_synchronize(dvn_ctx* ctx);

/* ---- proving-key point arrays ------------------------------------------------------------------------ */
/* Upload once per proving key (what icicle_bn254.ProvingKey's lazily uploaded device pointers hold,
 * constructed at /root/reference/circuits/artifacts.go:634-637).
 * xy_mont: N x 8 (G1) or N x 16 (G2) uint64. */
int32_t dvn_points_upload_g1(dvn_ctx* ctx, const uint64_t* xy_mont, size_t n, dvn_points** out);
int32_t dvn_points_upload_g2(dvn_ctx* ctx, const uint64_t* xy_mont, size_t n, dvn_points** out);
/* Synthetic SRS generated on the device: P_i = [s0]G + i*[s1]G with (s0, s1) the first two Fr of the
 * SplitMix64 stream `seed`.  Bench/test input generator. */
int32_t dvn_points_synthetic_g1(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out);
int32_t dvn_points_synthetic_g2(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out);
/* (Re)build the set as an expanded table for window size window_bits in [11, 21] (row w = 2^(window_bits * w) * P_i,
 * 254 / window_bits + 1 rows), or collapse it back to a plain set with 0.  By default every set of >= 2^10 points, G1 or
 * G2, is expanded at upload (window 15 below 2^15 points, 16 below 2^18, 17 below 2^20, 20 from there) unless its table would exceed
 * DVN_MSM_EXPAND_MAX_GB (16 GiB); DVN_MSM_EXPAND=0 turns the default off. */
int32_t dvn_points_set_expanded(dvn_ctx* ctx, dvn_points* pts, uint32_t window_bits);
/* Synthetic scalar vector written to DEVICE memory d_out (n x 4 uint64, gnark Montgomery form): kind 0 =
 * uniform mod r from SplitMix64 stream `seed`, kind 1 = the witness-like mixture (40 % zero,
 * 25 % one, 25 % < 2^64, 10 % uniform).  Bench/test input generator. */
int32_t dvn_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out);
/* Copy points [off, off+n) back to the host in gnark's affine Montgomery image. */
int32_t dvn_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy_mont);
size_t dvn_points_len(const dvn_points* pts);
int32_t dvn_points_free(dvn_ctx* ctx, dvn_points* pts);

/* ---- MultiExp (replaces gnark-crypto G1Jac.MultiExp / G2Jac.MultiExp) -------------- */
/* result = sum_{i<n} scalars[i] * points[off+i].  scalars_mont: n x 4 uint64 on the HOST.
 * out_jac: a valid Jacobian representative (this library returns the affine normal form with Z = 1, or
 * (1,1,0) for infinity); parity with gnark is defined after affine normalisation.
 * flags: DVN_MSM_* bits.
 * Host buffers of 16 MiB and more (a scalar vector here, the witness of a proof) are read through a ring of pinned chunks
 * that the context allocates at its first large copy (32 MiB of pinned host memory per context, freed by dvn_shutdown) and
 * that four short-lived host threads fill: the caller's memory is never pinned and is free when the call returns.
 * DVN_H2D_STAGED=0 in the environment keeps the plain copy. */
#define DVN_MSM_DEFAULT 0u
#define DVN_MSM_SCALARS_ON_DEVICE 1u   /* scalars_mont is a device pointer (HBM-resident witness) */
int32_t dvn_msm_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                   uint32_t flags, uint64_t out_jac[12]);
int32_t dvn_msm_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                   uint32_t flags, uint64_t out_jac[24]);
/* `batch` MultiExps over ONE point set in one pass (many commitments over one basis): scalars_mont = batch x n x 4 uint64,
 * vector b at [b*n, (b+1)*n) (host, or device with DVN_MSM_SCALARS_ON_DEVICE); out_jac = batch x 12, each as dvn_msm_g1's.
 * batch * windows(n) * n must stay below 2^31. */
int32_t dvn_msm_g1_batched(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                           uint32_t batch, uint32_t flags, uint64_t* out_jac);
/* Asynchronous pair: _submit queues the MSM on one of the context's four lanes (own stream + workspace) and returns a
 * ticket; _collect waits for it and writes the result.  With several MSMs in flight the sort and the latency-bound tail of
 * one run under the accumulate kernel of the other (measured +10 % throughput at 2^22, +19 % at 2^20, +52 % at 2^18).
 * Host scalars are copied before _submit returns; DEVICE scalars must stay untouched until _collect.  At most four
 * tickets may be outstanding per context (one per lane), whichever threads hold them.  The synchronous entry points take their
 * lanes from the same four (an MSM one, a proof up to four) and FAIL with "too many MSMs in flight" rather than wait when
 * outstanding tickets leave too few -- the ticket holder may be the calling thread.  Callers that mix tickets and synchronous
 * calls on several threads collect their tickets first, or give the asynchronous work a context of its own. */
int32_t dvn_msm_g1_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                          uint32_t flags, int32_t* ticket);
int32_t dvn_msm_g1_collect(dvn_ctx* ctx, int32_t ticket, uint64_t out_jac[12]);
int32_t dvn_msm_g2_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                          uint32_t flags, int32_t* ticket);
int32_t dvn_msm_g2_collect(dvn_ctx* ctx, int32_t ticket, uint64_t out_jac[24]);
/* Window-sharded variant for multi-GPU: this rank accumulates windows
 * [win_lo, win_hi) of the `nwin` windows dvn_msm_plan reports and returns one extended-Jacobian point
 * (X, Y, ZZ, ZZZ; 16 x uint64 for G1) per window in window_sums; ranks exchange them (RCCL all-gather) and any rank
 * finishes with dvn_msm_combine_g1. */
int32_t dvn_msm_plan(size_t n, uint32_t* window_bits, uint32_t* nwin);
/* The plan for THIS point set: large sets are stored as an expanded table (row w = 2^(c w) * P) whose
 * windows all share one bucket set; then combine_bits = 0 and dvn_msm_g1_windows returns the partial result of the
 * window range in slot 0 (remaining slots infinity): partial results add up, dvn_msm_combine_g1(.., window_bits = 0). */
int32_t dvn_msm_plan_points(const dvn_points* pts, size_t n, uint32_t* window_bits, uint32_t* nwin, uint32_t* combine_bits);
int32_t dvn_msm_g1_windows(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                           uint32_t flags, uint32_t win_lo, uint32_t win_hi, uint64_t* window_sums /* (win_hi-win_lo) x 16 */);
/* asynchronous halves of dvn_msm_g1_windows (same lanes/tickets as dvn_msm_g1_submit): lets a rank queue the next
 * step's windows while it exchanges and combines the previous step's sums */
int32_t dvn_msm_g1_windows_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                                  uint32_t flags, uint32_t win_lo, uint32_t win_hi, int32_t* ticket);
int32_t dvn_msm_g1_windows_collect(dvn_ctx* ctx, int32_t ticket, uint64_t* window_sums /* (win_hi-win_lo) x 16 */);
int32_t dvn_msm_combine_g1(dvn_ctx* ctx, const uint64_t* window_sums /* nwin x 16 */, uint32_t nwin, uint32_t window_bits,
                           uint64_t out_jac[12]);

/* One MSM shared between the GPUs of a node BY ENTRY COUNT (north_star: "shard by Pippenger window ... all-reduce of partial
 * bucket sums"): rank `rank` of `world` returns its share as a G1Jac image; the shares of all ranks add up to the MSM
 * (dvn_msm_g1_sum, host only: RCCL has no elliptic-curve reduction, the ranks all-gather 96 bytes each).  On an expanded point
 * set a share is an equal range of the windows x n entries -- balanced for any world size, where whole windows are not (13
 * windows over 8 ranks); a set too small to be expanded is computed whole by rank 0.  Device scalars only.  Tickets are the
 * lanes' (dvn_msm_g1_submit). */
int32_t dvn_msm_g1_share_submit(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* d_scalars_mont,
                                uint32_t flags, uint32_t rank, uint32_t world, int32_t* ticket);
int32_t dvn_msm_g1_share_collect(dvn_ctx* ctx, int32_t ticket, uint64_t partial_jac[12]);
/* jacs: the NORMALISED images the library emits -- Z is 0 (infinity) or the Montgomery 1; any other Z is DVN_ERR_INVALID_ARG
 * (a general G1Jac, e.g. gnark's, must be brought to affine first) */
int32_t dvn_msm_g1_sum(const uint64_t* jacs /* count x 12 */, uint32_t count, uint64_t out_jac[12]);
/* One MSM over ALL devices of a multi-device context (dvn_init with ndev > 1), the in-process form of the shared MSM: sets[i]
 * is the same point set uploaded on dvn_ctx_device(ctx, i); a host thread per device runs that device's share (rank i of
 * ndev, as dvn_msm_g1_share_submit) on HOST scalars it uploads itself, and the ndev partial sums are added on the host.  One
 * process holds every partial result, so no collective runs.  Same result as dvn_msm_g1(ctx, sets[0], ...). */
int32_t dvn_msm_g1_multi(dvn_ctx* ctx, const dvn_points* const* sets, uint32_t nsets, size_t off, size_t n,
                         const uint64_t* scalars_mont, uint64_t out_jac[12]);

/* ---- fr/fft (replaces gnark-crypto fft.Domain) ------------------------------------- */
#define DVN_DIF 0
#define DVN_DIT 1
int32_t dvn_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out);      /* fft.NewDomain(1 << log_n) */
int32_t dvn_ntt_domain_free(dvn_ctx* ctx, dvn_domain* dom);
/* Domain.FFT (inverse = 0) / Domain.FFTInverse (inverse = 1), in place on n = 2^log_n elements.
 * data on the host (copied in and out) or, with DVN_NTT_DATA_ON_DEVICE, a device pointer. */
#define DVN_NTT_DATA_ON_DEVICE 1u
int32_t dvn_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset,
                uint32_t flags);

/* ---- groth16 computeH -------------------------------------------------------------- */
/* a, b, c: len <= n elements each (host, or device with DVN_NTT_DATA_ON_DEVICE); h_out: n elements, the
 * coefficients of h in bit-reversed order exactly as gnark's computeH leaves them. */
int32_t dvn_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b,
                              const uint64_t* c, size_t len, uint64_t* h_out, uint32_t flags);

/* ---- Groth16 orchestration: what the Go shim's hipGroth16Prove does, in C++ ------------- */
/* A sparse R1CS  (A w) o (B w) = (C w)  stands in for gnark's constraint.ConstraintSystem.  dvn_r1cs_synthetic builds
 * the seeded multiplication-gate instance (rows (w[l1] + k w[l2]) * w[r] = w[o], wire 0 =
 * constant 1, wires 1..n_public-1 public) together with its satisfying wire assignment.  Host only. */
typedef struct dvn_r1cs dvn_r1cs;
typedef struct dvn_groth16_pk dvn_groth16_pk;
int32_t dvn_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out);
/* gnark ecc.ID of the system's curve as the prover mirror numbers them: 1 BN254, 2 BLS12_377, 3 BLS12_381, 4 BW6_761 */
int32_t dvn_r1cs_curve(const dvn_r1cs* cs);
int32_t dvn_r1cs_info(const dvn_r1cs* cs, uint32_t* n_wires, uint32_t* n_public, uint32_t* n_constraints);
int32_t dvn_r1cs_witness(const dvn_r1cs* cs, uint64_t* out /* n_wires x 4, Montgomery */);
void dvn_r1cs_free(dvn_r1cs* cs);
/* prover.Setup (/root/reference/prover/setup.go:15-28 -> groth16.Setup): toxic waste (tau, alpha, beta, gamma, delta)
 * drawn from SplitMix64(seed) so the result is reproducible; Lagrange evaluation, K and bit-reversed Z, fixed-base
 * batch multiplications on the device, InfinityA/B filtering.  The key's point arrays stay resident in HBM. */
int32_t dvn_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out);
int32_t dvn_groth16_pk_info(const dvn_groth16_pk* pk, uint32_t* log_n, uint64_t sizes[5] /* |A|,|B|,|K|,|Z|,|B2| */);
int32_t dvn_groth16_pk_free(dvn_ctx* ctx, dvn_groth16_pk* pk);
/* groth16.Prove (gnark backend/groth16/bn254/prove.go) on a complete wire assignment: (a,b,c) = (Aw,Bw,Cw), computeH,
 * wire filtering, 4 G1 MSMs + 1 G2 MSM, blinding with the injected (r, s) (gnark draws them from crypto/rand).
 * witness: n_wires x 4 uint64 Montgomery (host; device pointer if flags & 1).
 * proof_out: Ar (8) | Bs (16) | Krs (8) uint64 = groth16_bn254.Proof{Ar, Bs, Krs} affine images. */
int32_t dvn_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                          const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32]);
/* Sharded proving (BASELINE.json config 5: the statetransition proof's MSMs sharded across the GPUs of one node).  Every rank
 * holds the whole key and witness; rank `rank` of `world` runs the "solve" and computeH and ITS SHARE of the five MSMs -- an
 * equal range of the (row, point) entries of every expanded key set, so 13 rows over 8 ranks are 1.625 rows each -- and
 * returns 72 uint64 of partial results (Jacobian images of the A, B1, K, Z partial sums, 12 each, and of B2, 24); the ranks
 * all-gather the 576-byte buffers (RCCL), and dvn_groth16_prove_combine (host only) adds them and assembles the proof.  A
 * key set too small to be expanded (< 2^10 points) is computed whole by rank 0. */
int32_t dvn_groth16_prove_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                uint32_t rank, uint32_t world, uint64_t partial_out[72]);
int32_t dvn_groth16_prove_combine(const dvn_groth16_pk* pk, const uint64_t* partials /* world x 72 */, uint32_t world,
                                  const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32]);
/* One proof over ALL devices of a multi-device context (dvn_init with ndev > 1): keys[i] is the same proving key resident
 * on dvn_ctx_device(ctx, i) (built by dvn_groth16_setup or imported there).  Every device runs its share of the five MSMs --
 * an equal range of the (row, point) entries of each expanded key set -- beside a replicated computeH, on a host thread of
 * its own; the five partial sums per device return through pinned host memory and are added on the host.  One process holds
 * every partial result, so no collective runs: the multi-PROCESS form (one rank per GPU, dvn_groth16_prove_shard + an RCCL
 * all-gather of 576 bytes + dvn_groth16_prove_combine) stays available.  This shortens the node's serialisation point, one
 * statetransition proof at a time (/root/reference/sequencer/statetransition.go:80).  Witness and (a, b, c) in HOST memory. */
int32_t dvn_groth16_prove_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* witness,
                                const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32]);
/* per-proof statistics of the last prove call on this key as one JSON object (any curve): device ms until h is ready / after /
 * total, device ms and pair count of each of the five MSMs, host-to-device bytes, algorithmic bytes and the achieved GB/s --
 * the device-side detail behind the reference's "proof generated took=..." log line (/root/reference/circuits/artifacts.go:
 * 572-577).  buf = NULL queries the size (needed, incl. the terminating 0). */
int32_t dvn_groth16_last_stats_json(const dvn_groth16_pk* pk, char* buf, size_t cap, size_t* needed);
/* device time of the last prove call (any curve): ms_h = start until h is ready on the main stream (witness copy, gathers,
 * "solve", computeH -- the B2, A, B1 and K MSMs already run beside it on the four lanes), ms_msm = from there to the last MSM, total */
int32_t dvn_groth16_last_ms(const dvn_groth16_pk* pk, float* ms_h, float* ms_msm, float* ms_total);

/* Pedersen commitment and its proof of knowledge (gnark-crypto fr/pedersen ProvingKey.Commit /
 * ProveKnowledge, reached from the BSB22 commitment hint of groth16.Prove): commitment = MSM(Basis, values),
 * pok = MSM(BasisExpSigma, values); with one commitment (statetransition) the BatchProve fold is the identity.
 * values: n x 4 uint64 Montgomery (host; device pointer if flags & 1).  Outputs: G1Affine images. */
int32_t dvn_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma,
                            const uint64_t* values_mont, size_t n, uint32_t flags,
                            uint64_t commitment_out[8], uint64_t pok_out[8]);
/* Solidity ABI image (uint256[8], uint256[2], uint256[2]) of a proof, as /root/reference/solidity/solidity.go:29-73,
 * 85-116 builds it for the on-chain verifier: big-endian words, G2 coordinates A1 before A0.
 * commitment / commitment_pok: G1Affine images or NULL (zeros).  Host only. */
int32_t dvn_groth16_proof_abi_encode(const uint64_t proof[32], const uint64_t commitment[8], const uint64_t commitment_pok[8],
                                     uint8_t out[384]);

/* ---- mirror of the reference's `prover` package (csrc/dvn_prover_pkg.hip) --------------------- */
/* Same control flow as /root/reference/prover/prover_gpu.go:66-123: Prove -> prover func (SetProver seam) ->
 * defaultProver: if UseGPUProver (env GPU_PROVER in {true,y,1,yes}, prover/config.go:18-26) try the GPU and on ANY
 * error print "GPU proving failed (<err>), falling back to CPU" and run the CPU prover.  The CPU prover is the
 * embedder's (gnark groth16.Prove): register it with dvn_prover_set_cpu_backend; none registered -> error, never a
 * silent CPU computation.  curve ids: 1 BN254, 2 BLS12-377, 3 BLS12-381, 4 BW6-761 -- the four legs of callGPUProver
 * (prover_gpu.go:27-57), each with a GPU path here; any other id returns "GPU proving not supported for curve ..." like
 * prover_gpu.go:58-60.  witness / r / s / proof have the widths of the curve: scalars 4 uint64 (BW6-761: 6), proofs 32 uint64
 * on BN254, 48 on the two BLS12 curves, 72 on BW6-761 -- plus 16 for a BN254 key with a BSB22 commitment, whose proof image
 * continues with Commitments[0] | CommitmentPok (dvn_groth16_proof_words gives the size for a key).  The package state (UseGPUProver, SetProver, the CPU backend) may be
 * read and written from any thread. */
typedef struct dvn_prover_request {
    int curve; dvn_ctx* ctx; const dvn_r1cs* ccs; dvn_groth16_pk* pk;
    const uint64_t* witness; const uint64_t* r; const uint64_t* s; uint64_t* proof;
} dvn_prover_request;
typedef int32_t (*dvn_prover_func)(const dvn_prover_request*, void* user);
/* the CPU prover of the embedder (gnark's groth16.Prove): it receives what the reference's CPUProverWithWitness receives --
 * the constraint system and the (plain) proving key next to the witness (prover_gpu.go:126-131) */
typedef int32_t (*dvn_cpu_backend)(int curve, const dvn_r1cs* ccs, dvn_groth16_pk* pk, const uint64_t* witness, const uint64_t* r,
                                   const uint64_t* s, uint64_t* proof, void* user);
int32_t dvn_prover_use_gpu(void);
void dvn_prover_set_use_gpu(int32_t on);
void dvn_prover_reload_env(void);                                   /* re-run the package init(): read GPU_PROVER */
void dvn_prover_set_prover(dvn_prover_func p, void* user);          /* prover.SetProver; NULL restores defaultProver */
void dvn_prover_set_cpu_backend(dvn_cpu_backend b, void* user);
int32_t dvn_prover_last_backend(void);                              /* 1 GPU, 2 CPU backend, 3 custom prover func */
int32_t dvn_prover_prove(dvn_ctx* ctx, int32_t curve, const dvn_r1cs* ccs, dvn_groth16_pk* pk, const uint64_t* witness,
                         const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out);
/* The two provers behind the seam, callable directly as the reference's exported GPUProver / CPUProver are
 * (/root/reference/prover/prover_gpu.go:87-106).  Without a device context (ctx == NULL: what a host without a GPU has after
 * dvn_init failed) dvn_prover_gpu_prover fails with "GPU prover not supported in this build", the text the reference's
 * CPU-build stubs panic with (/root/reference/prover/prover_cpu.go:41-44,61-64; prover_cpu_test.go:7-25). */
int32_t dvn_prover_gpu_prover(dvn_ctx* ctx, int32_t curve, const dvn_r1cs* ccs, dvn_groth16_pk* pk, const uint64_t* witness,
                              const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out);
int32_t dvn_prover_cpu_prover(dvn_ctx* ctx, int32_t curve, const dvn_r1cs* ccs, dvn_groth16_pk* pk, const uint64_t* witness,
                              const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out);
int32_t dvn_prover_setup(dvn_ctx* ctx, const dvn_r1cs* ccs, uint64_t seed, dvn_groth16_pk** pk);   /* prover.Setup */

/* ---- BLS12-377 ------------------------------------------------------------------------------------ */
/* The voteverifier's curve (/root/reference/spec/params/params.go:21-27; reference GPU dispatch prover/prover_gpu.go:36-43).
 * Same kernels, instantiated on 377-bit base-field elements.  gnark images: fp.Element 6 x uint64 (R = 2^384),
 * fr.Element 4 x uint64, G1Affine 12 x uint64, G2Affine 24 x uint64 (E2 = A0 + A1 u, u^2 = -5), G1Jac 18, G2Jac 36.
 * Handles are the same dvn_points / dvn_domain types (they remember their curve; mixing curves is rejected);
 * dvn_points_len / dvn_points_free / dvn_ntt_domain_free apply to them as well. */
int32_t dvn_bls12377_points_upload_g1(dvn_ctx* ctx, const uint64_t* xy_mont /* n x 12 */, size_t n, dvn_points** out);
int32_t dvn_bls12377_points_upload_g2(dvn_ctx* ctx, const uint64_t* xy_mont /* n x 24 */, size_t n, dvn_points** out);
int32_t dvn_bls12377_points_synthetic_g1(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out);
int32_t dvn_bls12377_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy_mont);
int32_t dvn_bls12377_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out);
int32_t dvn_bls12377_msm_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                            uint32_t flags, uint64_t out_jac[18]);
int32_t dvn_bls12377_msm_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                            uint32_t flags, uint64_t out_jac[36]);
int32_t dvn_bls12377_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out);      /* log_n <= 27 (two-adicity 47) */
int32_t dvn_bls12377_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset,
                         uint32_t flags);
/* computeH of gnark's backend/groth16/bls12-377/prove.go, same contract as dvn_groth16_compute_h */
int32_t dvn_bls12377_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                      size_t len, uint64_t* h_out, uint32_t flags);

/* Groth16 on BLS12-377 (the voteverifier proof: 60 per batch): the same Setup / Prove mirrors as dvn_groth16_setup /
 * dvn_groth16_prove above, over this curve.  dvn_r1cs / dvn_groth16_pk handles remember their curve (mixing is rejected);
 * dvn_r1cs_info / _witness / _free and dvn_groth16_pk_info / _pk_free / _last_ms apply to them as well.
 * proof_out: Ar (12) | Bs (24) | Krs (12) uint64 = gnark's groth16_bls12377.Proof images. */
int32_t dvn_bls12377_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out);
int32_t dvn_bls12377_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out);
int32_t dvn_bls12377_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                   const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[48]);

/* ---- BLS12-381 ------------------------------------------------------------------------------------ */
/* The fourth curve of the reference's GPU dispatch (/root/reference/prover/prover_gpu.go:45-50) and the curve of the
 * EIP-4844 blob commitment computed before every state-transition proof: a 4096-point G1 MSM over the KZG setup in
 * Lagrange form (/root/reference/crypto/blobs/blob.go:40-90, types/blobs.go:90-160).  Element images have the sizes of
 * BLS12-377 (fp 6 x uint64, fr 4 x uint64; E2 = A0 + A1 u with u^2 = -1); scalar-field NTT with coset shift 7. */
int32_t dvn_bls12381_points_upload_g1(dvn_ctx* ctx, const uint64_t* xy_mont /* n x 12 */, size_t n, dvn_points** out);
int32_t dvn_bls12381_points_upload_g2(dvn_ctx* ctx, const uint64_t* xy_mont /* n x 24 */, size_t n, dvn_points** out);
int32_t dvn_bls12381_points_synthetic_g1(dvn_ctx* ctx, uint64_t seed, size_t n, dvn_points** out);
int32_t dvn_bls12381_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy_mont);
int32_t dvn_bls12381_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out);
int32_t dvn_bls12381_msm_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                            uint32_t flags, uint64_t out_jac[18]);
int32_t dvn_bls12381_msm_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont,
                            uint32_t flags, uint64_t out_jac[36]);
int32_t dvn_bls12381_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out);      /* log_n <= 27 (two-adicity 32) */
int32_t dvn_bls12381_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data, int inverse, int decimation, int on_coset,
                         uint32_t flags);
/* computeH of gnark's backend/groth16/bls12-381/prove.go, same contract as dvn_groth16_compute_h */
int32_t dvn_bls12381_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                      size_t len, uint64_t* h_out, uint32_t flags);

int32_t dvn_bls12381_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out);
int32_t dvn_bls12381_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out);
int32_t dvn_bls12381_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                   const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[48]);

/* ---- BW6-761 -------------------------------------------------------------------------------------- */
/* The aggregator's curve (/root/reference/spec/params/params.go:24; one proof per batch; reference GPU dispatch
 * prover/prover_gpu.go:52-57).  G1 (y^2 = x^3 - 1) and G2
 * (y^2 = x^3 + 4) both live over the 761-bit base field, so one entry point serves both; `group` (1 or 2) is recorded in
 * the handle.  gnark images: fp.Element 12 x uint64 (R = 2^768), G1Affine/G2Affine 24 x uint64, G1Jac/G2Jac 36 x uint64;
 * scalars are fr.Element = 6 x uint64 (R = 2^384; the field is BLS12-377's base field), also the element type of the
 * NTT (fft.Domain of ecc/bw6-761/fr: two-adicity 46, FrMultiplicativeGen 15) and of computeH. */
int32_t dvn_bw6761_points_upload(dvn_ctx* ctx, int group, const uint64_t* xy_mont /* n x 24 */, size_t n, dvn_points** out);
/* P_i = [s0]B + i*[s1]B from one caller-supplied affine point B (bench/test input generator) */
int32_t dvn_bw6761_points_synthetic(dvn_ctx* ctx, int group, const uint64_t base_xy_mont[24], uint64_t seed, size_t n,
                                    dvn_points** out);
int32_t dvn_bw6761_points_download(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, uint64_t* xy_mont);
int32_t dvn_bw6761_scalars_synthetic(dvn_ctx* ctx, uint64_t seed, int kind, size_t n, void* d_out /* n x 48 bytes */);
int32_t dvn_bw6761_msm(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* scalars_mont /* n x 6 */,
                       uint32_t flags, uint64_t out_jac[36]);

int32_t dvn_bw6761_ntt_domain(dvn_ctx* ctx, uint32_t log_n, dvn_domain** out);                 /* log_n <= 27 */
int32_t dvn_bw6761_ntt(dvn_ctx* ctx, const dvn_domain* dom, uint64_t* data /* n x 6 */, int inverse, int decimation, int on_coset,
                       uint32_t flags);
int32_t dvn_bw6761_groth16_compute_h(dvn_ctx* ctx, const dvn_domain* dom, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                     size_t len, uint64_t* h_out /* n x 6 */, uint32_t flags);

/* Groth16 on BW6-761 (the aggregator proof): Setup / Prove mirrors as for the other curves.  witness: m x 6 uint64,
 * r_mont / s_mont: 6 uint64 each, proof_out: Ar (24) | Bs (24) | Krs (24) uint64 = gnark's groth16_bw6761.Proof images. */
int32_t dvn_bw6761_r1cs_synthetic(uint64_t seed, uint32_t n_constraints, uint32_t n_public, dvn_r1cs** out);
int32_t dvn_bw6761_groth16_setup(dvn_ctx* ctx, const dvn_r1cs* cs, uint64_t seed, dvn_groth16_pk** out);
int32_t dvn_bw6761_groth16_prove(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                 const uint64_t r_mont[6], const uint64_t s_mont[6], uint64_t proof_out[72]);

/* ---- Groth16 with one BSB22 commitment: the statetransition shape (8 public inputs + 1 commitment wire,
 * /root/reference/config/statetransition_vkey.sol:97-114; gnark Prove's commitment steps and fr/pedersen).
 * dvn_r1cs_synthetic_committed: as dvn_r1cs_synthetic plus a commitment wire after the n_public public wires and
 * n_committed committed private wires; dvn_r1cs_info then reports n_public + 1 public wires.  dvn_groth16_setup builds
 * the commitment key (Basis = K / gamma over the committed wires, BasisExpSigma = sigma * Basis).
 * dvn_groth16_prove_committed: Commitment and CommitmentPok (two MSMs), the challenge
 * keccak256(C.x || C.y || publicCommitted) mod r that the reference's Solidity verifier recomputes
 * (statetransition_vkey.sol:659-676), the completion of the witness, then the proof as dvn_groth16_prove.
 * Together with dvn_groth16_proof_abi_encode this yields the calldata of solidity/solidity.go:85-116. */
/* ---- gnark's own proving key (device handles in a side table keyed by the PK pointer) --------------------
 * dvn_groth16_pk_import takes the in-memory arrays of a *groth16_bn254.ProvingKey as the node loads it
 * (/root/reference/circuits/artifacts.go:391-406; zero-copy from the Go slices, copied to HBM before return): G1.A / G1.B /
 * G1.K / G1.Z / G2.B as gnark affine Montgomery images, the wire indices that go with A, B (= G2.B) and K in order (gnark
 * drops InfinityA / InfinityB points and keeps K for the private, non-committed wires), and the five key constants.
 * dvn_groth16_prove_solved then runs everything of groth16.Prove that follows the solver: w = all
 * wire values, a / b / c = the solver's constraint evaluations (len <= 2^log_n each, zero padded here). */
typedef struct dvn_groth16_key_arrays {
    uint32_t log_n;              /* domain size 2^log_n */
    uint64_t n_wires;            /* m */
    const uint64_t *A, *B, *K, *Z, *B2;          /* nA x 8, nB x 8, nK x 8, nZ x 8, nB x 16 uint64 */
    uint64_t nA, nB, nK, nZ;                     /* nZ >= 2^log_n - 1 */
    const uint32_t *idxA, *idxB, *idxK;          /* nA, nB, nK wire indices */
    const uint64_t *alpha_g1, *beta_g1, *delta_g1;   /* 8 uint64 each */
    const uint64_t *beta_g2, *delta_g2;              /* 16 uint64 each */
} dvn_groth16_key_arrays;
int32_t dvn_groth16_pk_import(dvn_ctx* ctx, const dvn_groth16_key_arrays* key, dvn_groth16_pk** out);
int32_t dvn_groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                 const uint64_t* c, size_t len, const uint64_t r_mont[4], const uint64_t s_mont[4],
                                 uint64_t proof_out[32]);
/* arrays of a key built by dvn_groth16_setup in the import layout: which = 0..4 A, B, K, Z, B2; 5 / 6 idxA / idxB (uint32);
 * 7 the constants alpha1 | beta1 | delta1 | beta2 | delta2.  out = NULL queries the size in uint64 words. */
int32_t dvn_groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words);
/* dvn_groth16_prove_solved over all devices of a multi-device context (see dvn_groth16_prove_multi): keys[i] imported on
 * dvn_ctx_device(ctx, i) */
int32_t dvn_groth16_prove_solved_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* w, const uint64_t* a,
                                       const uint64_t* b, const uint64_t* c, size_t len, const uint64_t r_mont[4], const uint64_t s_mont[4],
                                       uint64_t proof_out[32]);
/* ... and rank `rank` of `world`'s share of it for one-process-per-GPU deployments: partial_out as dvn_groth16_prove_shard's
 * (72 words); the ranks exchange them (dvn_comm_all_gather) and dvn_groth16_prove_combine assembles the proof.  Host w, a, b, c.
 * No counterpart in the reference (prover/prover_gpu.go:33 proves on one GPU). */
int32_t dvn_groth16_prove_solved_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                       const uint64_t* c, size_t len, uint32_t rank, uint32_t world, uint64_t partial_out[72]);
/* the same pair on the voteverifier and aggregator curves: partial results of 108 (BLS12-377) / 180 (BW6-761) uint64 per rank,
 * proofs of 48 / 72 words as dvn_<curve>_groth16_prove_solved writes them; r_mont / s_mont of 4 / 6 words */
int32_t dvn_bls12377_groth16_prove_solved_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                                const uint64_t* c, size_t len, uint32_t rank, uint32_t world, uint64_t partial_out[108]);
int32_t dvn_bls12377_groth16_prove_combine(const dvn_groth16_pk* pk, const uint64_t* partials /* world x 108 */, uint32_t world,
                                           const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[48]);
int32_t dvn_bw6761_groth16_prove_solved_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                              const uint64_t* c, size_t len, uint32_t rank, uint32_t world, uint64_t partial_out[180]);
int32_t dvn_bw6761_groth16_prove_combine(const dvn_groth16_pk* pk, const uint64_t* partials /* world x 180 */, uint32_t world,
                                         const uint64_t r_mont[6], const uint64_t s_mont[6], uint64_t proof_out[72]);
int32_t dvn_bw6761_groth16_prove_solved_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* w, const uint64_t* a,
                                              const uint64_t* b, const uint64_t* c, size_t len, const uint64_t* r_mont, const uint64_t* s_mont,
                                              uint64_t* proof_out);
/* dvn_pedersen_commit on the other curves of callGPUProver's dispatch (/root/reference/prover/prover_gpu.go:36-57): the
 * BSB22 commitment hint of the voteverifier (BLS12-377) and aggregator (BW6-761) circuits.  values: n scalars of the curve
 * (4 / 4 / 6 uint64 each); outputs: G1Affine images (12 / 12 / 24 uint64).  The challenge hash-to-field stays in gnark. */
int32_t dvn_bls12377_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma, const uint64_t* values_mont,
                                     size_t n, uint32_t flags, uint64_t commitment_out[12], uint64_t pok_out[12]);
int32_t dvn_bls12381_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma, const uint64_t* values_mont,
                                     size_t n, uint32_t flags, uint64_t commitment_out[12], uint64_t pok_out[12]);
int32_t dvn_bw6761_pedersen_commit(dvn_ctx* ctx, const dvn_points* basis, const dvn_points* basis_exp_sigma, const uint64_t* values_mont,
                                   size_t n, uint32_t flags, uint64_t commitment_out[24], uint64_t pok_out[24]);

/* the same three entry points on the other curves of prover_gpu.go's dispatch (element widths of that curve: G1 12 / 12 / 24,
 * G2 24 / 24 / 24 uint64 per affine point, scalars 4 / 4 / 6 uint64) */
int32_t dvn_bls12377_groth16_pk_import(dvn_ctx* ctx, const dvn_groth16_key_arrays* key, dvn_groth16_pk** out);
int32_t dvn_bls12377_groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                          const uint64_t* c, size_t len, const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out);
int32_t dvn_bls12377_groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words);
int32_t dvn_bls12381_groth16_pk_import(dvn_ctx* ctx, const dvn_groth16_key_arrays* key, dvn_groth16_pk** out);
int32_t dvn_bls12381_groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                          const uint64_t* c, size_t len, const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out);
int32_t dvn_bls12381_groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words);
int32_t dvn_bw6761_groth16_pk_import(dvn_ctx* ctx, const dvn_groth16_key_arrays* key, dvn_groth16_pk** out);
int32_t dvn_bw6761_groth16_prove_solved(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                                        const uint64_t* c, size_t len, const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out);
int32_t dvn_bw6761_groth16_pk_export_arrays(dvn_ctx* ctx, const dvn_groth16_pk* pk, int which, uint64_t* out, size_t cap_words, size_t* n_words);

/* ---- gnark's artifact FILES ---------------------------------------------------------------------
 * The node keeps proving keys as sha256-named files written by cmd/circuit-compile with pk.WriteTo
 * (/root/reference/cmd/circuit-compile/main.go:507-513) and loads them with pk.UnsafeReadFrom
 * (/root/reference/circuits/artifacts.go:391-406, hash check :123-133).  dvn_groth16_pk_load_file reads the same bytes
 * (compressed or raw point encodings) straight into a device-resident key for dvn_groth16_prove_solved: the points
 * are decompressed on the device, the InfinityA / InfinityB bitmaps become the wire lists of A and B, and K's wire list is
 * [n_public, nbWires) minus committed_wires -- the two facts the key file does not carry come from the constraint system
 * gnark loads beside it: ccs.GetNbPublicVariables(), and for committed_wires every private wire whose K point gnark's Setup
 * leaves out of pk.G1.K, i.e. the PrivateCommitted indices of the commitment TOGETHER WITH its CommitmentIndex wire (the wire
 * that receives the challenge: Setup moves its K point into the verifying key).  The count is checked against |G1.K|.
 * sha256_hex: the expected content hash (the file's name in the node's cache) or NULL; a mismatch is DVN_ERR_INVALID_ARG.
 * Keys are cached per context by content hash: loading the same file again returns the resident key -- provided the call names
 * the same n_public and committed_wires as the load that created it (anything else is DVN_ERR_INVALID_ARG) -- and cached keys
 * live until dvn_groth16_pk_evict or dvn_shutdown (dvn_groth16_pk_free leaves them alone).  _load_bytes: the same reader on a
 * memory image, uncached.  dvn_gnark_pk_scan (host only) reports the section sizes of an image: log_n, raw (0/1), nbWires, |A|,
 * |B|, |Z|, |K|, |G2.B|, nbCommitmentKeys, |Basis[0]|, whether the domain header carries its flag byte, whether the two []bool
 * bitmaps carry a uint32 length (both variants of each are accepted; the header's fields must agree with each other --
 * CardinalityInv * Cardinality = 1, Generator a primitive root of that order, the curve's coset generator -- so a misread
 * header is rejected rather than parsed further).  The format lives in the un-vendored gnark module: it is restated from the
 * published layout and exercised against this repository's own writer only.
 * The node loads four circuits' keys at boot (/root/reference/sequencer/circuit_artifacts.go:39-76): the voteverifier's
 * (BLS12-377: 48-byte coordinates, three flag bits, Tonelli-Shanks roots) and the aggregator's (BW6-761: 96-byte coordinates,
 * both groups over Fp) have the same three calls under their curve's prefix, feeding dvn_<curve>_groth16_prove_solved. */
int32_t dvn_gnark_pk_scan(const uint8_t* data, size_t len, uint64_t info[12]);
int32_t dvn_groth16_pk_evict(dvn_ctx* ctx, dvn_groth16_pk* pk);      /* drop a cached key from the cache and free it */
int32_t dvn_bls12377_gnark_pk_scan(const uint8_t* data, size_t len, uint64_t info[12]);
int32_t dvn_bls12377_groth16_pk_load_bytes(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed_wires,
                                           uint32_t n_committed_wires, dvn_groth16_pk** out);
int32_t dvn_bls12377_groth16_pk_load_file(dvn_ctx* ctx, const char* path, const char* sha256_hex, uint32_t n_public,
                                          const uint32_t* committed_wires, uint32_t n_committed_wires, dvn_groth16_pk** out);
int32_t dvn_bw6761_gnark_pk_scan(const uint8_t* data, size_t len, uint64_t info[12]);
int32_t dvn_bw6761_groth16_pk_load_bytes(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed_wires,
                                         uint32_t n_committed_wires, dvn_groth16_pk** out);
int32_t dvn_bw6761_groth16_pk_load_file(dvn_ctx* ctx, const char* path, const char* sha256_hex, uint32_t n_public,
                                        const uint32_t* committed_wires, uint32_t n_committed_wires, dvn_groth16_pk** out);
int32_t dvn_groth16_pk_load_bytes(dvn_ctx* ctx, const uint8_t* data, size_t len, uint32_t n_public, const uint32_t* committed_wires,
                                  uint32_t n_committed_wires, dvn_groth16_pk** out);
int32_t dvn_groth16_pk_load_file(dvn_ctx* ctx, const char* path, const char* sha256_hex, uint32_t n_public,
                                 const uint32_t* committed_wires, uint32_t n_committed_wires, dvn_groth16_pk** out);
/* Basis / BasisExpSigma of a loaded key's commitment (NULL handles when it has none), owned by the key: the arguments of
 * dvn_pedersen_commit in the solver's commitment hint */
int32_t dvn_groth16_pk_commitment_bases(const dvn_groth16_pk* pk, const dvn_points** basis, const dvn_points** basis_exp_sigma);

/* The verifying key of a Setup (the reference's Setup returns (pk, vk), /root/reference/prover/setup.go:15-28) as affine
 * Montgomery images of the key's curve: alpha (G1) | beta, gamma, delta (G2) | IC[n_public] (G1: CONSTANT and PUB_i of the
 * generated Solidity verifier) | with a commitment: Pedersen G, GSigmaNeg (G2).  out = NULL queries the size. */
int32_t dvn_groth16_vk_export(const dvn_groth16_pk* pk, uint64_t* out, size_t cap_words, size_t* n_words);
int32_t dvn_r1cs_synthetic_committed(uint64_t seed, uint32_t n_constraints, uint32_t n_public, uint32_t n_committed, dvn_r1cs** out);
int32_t dvn_groth16_prove_committed(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                    const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32],
                                    uint64_t commitment_out[8], uint64_t pok_out[8], uint64_t challenge_out[4]);
/* The committed proof shared between GPUs -- the statetransition proof is the node's serialisation point
 * (/root/reference/sequencer/statetransition.go:80,248) and its key carries exactly one commitment
 * (/root/reference/config/statetransition_vkey.sol:97-114,659-676).  Rank `rank` of `world` runs the commitment hint, the
 * challenge and the witness completion whole (the commitment bases are small; replicating them spares an exchange before the
 * challenge) and then its entry share of the five MSMs: partial_out as dvn_groth16_prove_shard, assembled by
 * dvn_groth16_prove_combine; every rank returns the same commitment, PoK and challenge.  _multi does the same over the devices
 * of a multi-device context from one call (keys[i] = the key on dvn_ctx_device(ctx, i)). */
int32_t dvn_groth16_prove_committed_shard(dvn_ctx* ctx, dvn_groth16_pk* pk, const uint64_t* witness, uint32_t flags,
                                          uint32_t rank, uint32_t world, uint64_t partial_out[72],
                                          uint64_t commitment_out[8], uint64_t pok_out[8], uint64_t challenge_out[4]);
int32_t dvn_groth16_prove_committed_multi(dvn_ctx* ctx, dvn_groth16_pk* const* keys, uint32_t nkeys, const uint64_t* witness,
                                          const uint64_t r_mont[4], const uint64_t s_mont[4], uint64_t proof_out[32],
                                          uint64_t commitment_out[8], uint64_t pok_out[8], uint64_t challenge_out[4]);
/* uint64 words of the proof image dvn_prover_prove / dvn_prover_gpu_prover write for this key: Ar | Bs | Krs of the key's curve
 * (32 on BN254, 48 on the BLS12 curves, 72 on BW6-761), followed by Commitments[0] | CommitmentPok (two G1Affine images, +16
 * on BN254) when the key carries a BSB22 commitment -- gnark's Proof holds those two fields beside Ar, Bs, Krs, and
 * callGPUProver routes such a key to the committed prover. */
int32_t dvn_groth16_proof_words(const dvn_groth16_pk* pk, size_t* words);
int32_t dvn_keccak256(const uint8_t* data, size_t len, uint8_t out[32]);     /* host only: Solidity's keccak256 */

/* ---- Blob KZG on BLS12-381 (EIP-4844 / EIP-7594) ------------------------------------------------
 * The step before every state-transition proof, /root/reference/crypto/blobs/blob.go:40-90 (BlobEvalData.Set):
 *   dvn_kzg_blob_to_commitment   replaces Blob.ComputeCommitment   /root/reference/types/blobs.go:88-95
 *   dvn_kzg_compute_proof        replaces Blob.ComputeProof        /root/reference/types/blobs.go:120-133  (proof, claim y)
 *   dvn_kzg_compute_cell_proofs  replaces Blob.ComputeCellProofs   /root/reference/types/blobs.go:97-104   (128 proofs)
 *   dvn_kzg_evaluate_barycentric mirrors EvaluateBarycentricNative /root/reference/crypto/blobs/barycentric.go:24-170
 * Byte formats are the reference's: blob = 4096 x 32-byte big-endian field elements (types.Blob), z / y = 32 bytes big
 * endian, commitments and proofs = 48-byte compressed G1 (types.KZGCommitment / KZGProof).  Errors as geth's kzg4844:
 * a non-canonical blob element or evaluation point -> DVN_ERR_INVALID_ARG.
 * dvn_kzg_setup_load takes the two G1 bases of /root/reference/config/kzg_trusted_setup.txt (config/kzg_setup.go:7-8) as
 * 4096 x 48 compressed bytes each, in the file's (natural) order; g1_monomial may be NULL (then no cell proofs).  The setup
 * is decompressed on the device and stays resident in HBM; it is bound to `ctx`. */
typedef struct dvn_kzg_setup dvn_kzg_setup;
int32_t dvn_kzg_setup_load(dvn_ctx* ctx, const uint8_t* g1_lagrange, const uint8_t* g1_monomial, dvn_kzg_setup** out);
int32_t dvn_kzg_setup_free(dvn_ctx* ctx, dvn_kzg_setup* setup);
int32_t dvn_kzg_blob_to_commitment(dvn_ctx* ctx, const dvn_kzg_setup* setup, const uint8_t* blob /* 131072 */, uint8_t out[48]);
int32_t dvn_kzg_compute_proof(dvn_ctx* ctx, const dvn_kzg_setup* setup, const uint8_t* blob, const uint8_t z[32],
                              uint8_t proof[48], uint8_t y[32]);
/* Blob.ComputeBlobProof (/root/reference/types/blobs.go:108-117): opening proof at the Fiat-Shamir point of EIP-4844's
 * compute_blob_kzg_proof; commitment is NOT checked against the blob (as in the reference). */
int32_t dvn_kzg_compute_blob_proof(dvn_ctx* ctx, const dvn_kzg_setup* setup, const uint8_t* blob, const uint8_t commitment[48],
                                   uint8_t proof[48]);
int32_t dvn_sha256(const uint8_t* data, size_t len, uint8_t out[32]);        /* host only */
int32_t dvn_kzg_compute_cell_proofs(dvn_ctx* ctx, const dvn_kzg_setup* setup, const uint8_t* blob, uint8_t* proofs /* 128 x 48 */);
int32_t dvn_kzg_evaluate_barycentric(dvn_ctx* ctx, const dvn_kzg_setup* setup, const uint8_t* blob, const uint8_t z[32], uint8_t y[32]);

/* ---- the exchange step for hosts that run one PROCESS per GPU (csrc/dvn_comm.hip) -----------------------------------
 * No counterpart in the reference, whose ICICLE backend drives one GPU (prover/prover_gpu.go:24-61); the serialisation point
 * a sharded proof shortens is sequencer/statetransition.go:80.  An RCCL communicator over xGMI: rank 0 makes an id
 * (dvn_comm_unique_id) and hands the 128 bytes to the other processes by any means; every rank calls dvn_comm_init with it
 * (a collective: it returns when all `world` ranks have called); dvn_comm_all_gather then gives every rank the `words` uint64
 * of every rank (rank r's at all[r * words ...]) -- the 12-word partial sums of dvn_msm_g1_share_collect, added with
 * dvn_msm_g1_sum, or the 72-word partial results of dvn_groth16_prove_shard, assembled with dvn_groth16_prove_combine.
 * RCCL has no elliptic-curve reduction operator, so the "all-reduce" of partial sums is this all-gather plus the local sum.
 * Host buffers in and out; words <= DVN_COMM_MAX_WORDS.  librccl is opened at the first call of this group, not linked. */
typedef struct dvn_comm dvn_comm;
#define DVN_COMM_MAX_WORDS 4096
int32_t dvn_comm_unique_id(uint8_t id[128]);
int32_t dvn_comm_init(dvn_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t world, dvn_comm** out);
int32_t dvn_comm_all_gather(dvn_ctx* ctx, dvn_comm* comm, const uint64_t* local, size_t words, uint64_t* all /* world x words */);
int32_t dvn_comm_free(dvn_ctx* ctx, dvn_comm* comm);

/* ---- the entry point that allocates nothing ------------------------------------------------------------------------
 * No counterpart in the reference: this one is for callers that cannot call hipMalloc -- tight serving loops, harnesses
 * that forbid allocation, code that must leave the allocator alone.  (It is NOT capturable into a HIP graph: the call
 * synchronises its stream and finishes the result on the host.)  Same MSM and same result as dvn_msm_g1, but every scratch array
 * of the pipeline (digits, sort buffers, bucket sums, edge partials) lives in a fixed 64 MiB device array that the library
 * holds on each GPU it runs on until the process ends, and memory is cleared by fill kernels.  Limits: n <=
 * DVN_MSM_STATIC_MAX_N, scalars in DEVICE memory (flags must carry DVN_MSM_SCALARS_ON_DEVICE), BN254 G1; anything beyond
 * is DVN_ERR_INVALID_ARG with a message that names the limit.  Calls no hipMalloc / hipFree.  Every call on one GPU shares
 * that GPU's array; the library serialises its own calls per GPU (a process-wide lock per device ordinal, so sibling contexts
 * of one dvn_init cannot corrupt each other) -- they queue, they do not overlap. */
#define DVN_MSM_STATIC_MAX_N 65536u
int32_t dvn_msm_g1_static(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const uint64_t* d_scalars_mont,
                          uint32_t flags, uint64_t out_jac[12]);

/* ---- timing of the last call's kernels, measured with HIP events on the library's own stream ---------- */
/* which: 0 = whole device section of the last MSM/NTT call, 1 = its dominant kernel (MSM: bucket
 * accumulation; NTT: sum of butterfly passes). */
int32_t dvn_last_kernel_ms(dvn_ctx* ctx, int which, float* ms);

#ifdef __cplusplus
}
#endif
#endif
