// C++ mirror of the reference's `prover` package: the seam the sequencer calls.
//
//   reference (Go)                                   here
//   prover/config.go:16-26   UseGPUProver + init()   prover::UseGPUProver, prover::init_from_env()  (GPU_PROVER in {true,y,1,yes})
//   prover/config.go:29-56   ProverFunc, SetProver   prover::ProverFunc, prover::SetProver
//   prover/prover_gpu.go:66  Prove                   prover::Prove            -> prover(...)
//   prover/prover_gpu.go:71  defaultProver           prover::defaultProver    GPU, on ANY error print + CPU fallback
//   prover/prover_gpu.go:87  CPUProver(+WithWitness) prover::CPUProver        -> the registered CPU backend (gnark in production)
//   prover/prover_gpu.go:99  GPUProver(+WithWitness) prover::GPUProver        -> callGPUProver: per-curve dispatch
//   prover/prover_gpu.go:24  callGPUProver           prover::callGPUProver    all four curves of the reference; anything else: "GPU proving not supported for curve"
//   prover/setup.go:15       Setup                   prover::Setup            -> dvn_groth16_setup, timing log
// The CPU prover is NOT part of this library: in the node it is gnark's groth16.Prove.  CPUProver therefore calls a
// backend registered by the embedder (dvn_prover_set_cpu_backend); with none registered it returns an error -- it never
// silently computes on the CPU.
#include <atomic>
#include <chrono>
#include "dvn_internal.h"

namespace prover {

enum CurveID { UNKNOWN = 0, BN254 = 1, BLS12_377 = 2, BLS12_381 = 3, BW6_761 = 4 };
static const char* curve_name(int c) {
    switch (c) { case BN254: return "bn254"; case BLS12_377: return "bls12_377"; case BLS12_381: return "bls12_381"; case BW6_761: return "bw6_761"; default: return "unknown"; }
}

// request = (curve, ccs, pk, complete wire assignment, blinding factors r and s that gnark draws inside Prove)
typedef dvn_prover_request Request;
typedef dvn_prover_func ProverFunc;
typedef dvn_cpu_backend CpuBackend;

// Package-level state, as in prover/config.go: read on every Prove (from any thread / goroutine), written by SetProver, the
// test hooks and init().  The flag is atomic; the two (function, user) pairs are replaced under a mutex and copied out under it,
// so a Prove racing a SetProver sees either the old pair or the new one, never a mix.
std::atomic<bool> UseGPUProver{false};
static std::mutex hooks_mu;
static ProverFunc custom_prover = nullptr; static void* custom_user = nullptr;
static CpuBackend cpu_backend = nullptr; static void* cpu_user = nullptr;
static thread_local int last_backend = 0;   // 1 = GPU, 2 = CPU, 3 = custom

static bool env_wants_gpu() {
    const char* v = getenv("GPU_PROVER");
    return v && (!strcmp(v, "true") || !strcmp(v, "y") || !strcmp(v, "1") || !strcmp(v, "yes"));
}
void init_from_env() {
    UseGPUProver.store(env_wants_gpu());
    fprintf(stderr, "[prover] GPU prover usage enabled=%s\n", UseGPUProver.load() ? "true" : "false");
}
struct EnvInit { EnvInit() { UseGPUProver.store(env_wants_gpu()); } };
static EnvInit env_init;                     // Go's package init()

int32_t CPUProverWithWitness(const Request* q) {
    last_backend = 2;
    CpuBackend fn; void* user;
    { std::lock_guard<std::mutex> lk(hooks_mu); fn = cpu_backend; user = cpu_user; }
    if (!fn) return fail(DVN_ERR_UNSUPPORTED, "CPU prover not linked: register gnark's groth16.Prove with dvn_prover_set_cpu_backend");
    // the reference hands the CPU prover (ccs, cpuReadyProvingKey(pk), witness) (prover_gpu.go:126-131): so does this call
    return fn(q->curve, q->ccs, q->pk, q->witness, q->r, q->s, q->proof, user);
}
int32_t callGPUProver(const Request* q) {
    switch (q->curve) {
        case BN254:
            if (!q->pk) return fail(DVN_ERR_INVALID_ARG, "proving key type mismatch for BN254: expected a device-resident dvn_groth16_pk, got NULL");
            {   // the statetransition key carries one BSB22 commitment: its proof image continues with Commitments[0] | CommitmentPok
                size_t words = 0;
                if (dvn_groth16_proof_words(q->pk, &words) == DVN_OK && words == 48)
                    return dvn_groth16_prove_committed(q->ctx, q->pk, q->witness, 0, q->r, q->s, q->proof, q->proof + 32, q->proof + 40, nullptr);
            }
            return dvn_groth16_prove(q->ctx, q->pk, q->witness, 0, q->r, q->s, q->proof);
        case BLS12_377:          // the reference also sets PinToGPU here (prover_gpu.go:41): keys are always device-resident in this library
            if (!q->pk) return fail(DVN_ERR_INVALID_ARG, "proving key type mismatch for BLS12_377: expected a device-resident dvn_groth16_pk, got NULL");
            return dvn_bls12377_groth16_prove(q->ctx, q->pk, q->witness, 0, q->r, q->s, q->proof);
        case BLS12_381:
            if (!q->pk) return fail(DVN_ERR_INVALID_ARG, "proving key type mismatch for BLS12_381: expected a device-resident dvn_groth16_pk, got NULL");
            return dvn_bls12381_groth16_prove(q->ctx, q->pk, q->witness, 0, q->r, q->s, q->proof);
        case BW6_761:
            if (!q->pk) return fail(DVN_ERR_INVALID_ARG, "proving key type mismatch for BW6_761: expected a device-resident dvn_groth16_pk, got NULL");
            return dvn_bw6761_groth16_prove(q->ctx, q->pk, q->witness, 0, q->r, q->s, q->proof);
        default:
            return fail(DVN_ERR_UNSUPPORTED, "GPU proving not supported for curve %s", curve_name(q->curve));
    }
}
void note_backend(int b) { last_backend = b; }
int32_t GPUProverWithWitness(const Request* q) { last_backend = 1; return callGPUProver(q); }
// frontend.NewWitness(assignment, field) has no counterpart here: the assignment already is the wire vector
int32_t CPUProver(const Request* q) { return CPUProverWithWitness(q); }
int32_t GPUProver(const Request* q) { return GPUProverWithWitness(q); }

int32_t ProveWithWitness(const Request* q) {
    if (UseGPUProver.load()) {
        int32_t rc = GPUProverWithWitness(q);
        if (rc != DVN_OK) {
            printf("GPU proving failed (%s), falling back to CPU\n", tl_error.c_str());
            fflush(stdout);
            return CPUProverWithWitness(q);
        }
        return rc;
    }
    return CPUProverWithWitness(q);
}
int32_t defaultProver(const Request* q) { return ProveWithWitness(q); }
int32_t Prove(const Request* q) {
    ProverFunc fn; void* user;
    { std::lock_guard<std::mutex> lk(hooks_mu); fn = custom_prover; user = custom_user; }
    if (fn) { last_backend = 3; return fn(q, user); }
    return defaultProver(q);
}
void SetProver(ProverFunc p, void* user) { std::lock_guard<std::mutex> lk(hooks_mu); custom_prover = p; custom_user = user; }
void SetCpuBackend(CpuBackend b, void* user) { std::lock_guard<std::mutex> lk(hooks_mu); cpu_backend = b; cpu_user = user; }

int32_t Setup(dvn_ctx* ctx, const dvn_r1cs* ccs, uint64_t seed, dvn_groth16_pk** pk) {
    uint32_t nc = 0; dvn_r1cs_info(ccs, nullptr, nullptr, &nc);
    fprintf(stderr, "[prover] generating circuit keys gpu=%s constraints=%u\n", UseGPUProver.load() ? "true" : "false", nc);
    auto t0 = std::chrono::steady_clock::now();
    const int32_t cid = dvn_r1cs_curve(ccs);
    int32_t rc = cid == BLS12_377 ? dvn_bls12377_groth16_setup(ctx, ccs, seed, pk) : cid == BLS12_381 ? dvn_bls12381_groth16_setup(ctx, ccs, seed, pk)
               : cid == BW6_761 ? dvn_bw6761_groth16_setup(ctx, ccs, seed, pk) : dvn_groth16_setup(ctx, ccs, seed, pk);
    if (rc == DVN_OK) {
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[prover] circuit keys setup done took=%.3fs gpu=%s constraints=%u\n", s, UseGPUProver.load() ? "true" : "false", nc);
    }
    return rc;
}
}  // namespace prover

// ---- C-ABI of the package mirror -----------------------------------------------------------------------------
extern "C" int32_t dvn_prover_use_gpu(void) try { return prover::UseGPUProver.load() ? 1 : 0; } DVN_CATCH
extern "C" void dvn_prover_set_use_gpu(int32_t on) { prover::UseGPUProver.store(on != 0); }
extern "C" void dvn_prover_reload_env(void) { prover::init_from_env(); }
extern "C" void dvn_prover_set_prover(prover::ProverFunc p, void* user) { prover::SetProver(p, user); }
extern "C" void dvn_prover_set_cpu_backend(prover::CpuBackend b, void* user) { prover::SetCpuBackend(b, user); }
extern "C" int32_t dvn_prover_last_backend(void) try { return prover::last_backend; } DVN_CATCH
extern "C" int32_t dvn_prover_prove(dvn_ctx* ctx, int32_t curve, const dvn_r1cs* ccs, dvn_groth16_pk* pk, const uint64_t* witness,
                                    const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) try {
    prover::Request q{curve, ctx, ccs, pk, witness, r_mont, s_mont, proof_out};
    return prover::Prove(&q);
} DVN_CATCH
extern "C" int32_t dvn_prover_gpu_prover(dvn_ctx* ctx, int32_t curve, const dvn_r1cs* ccs, dvn_groth16_pk* pk, const uint64_t* witness,
                                         const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) try {
    // a process that holds no device context is this library's "CPU build": the reference's exported stubs there panic with this
    // text whatever their arguments (prover/prover_cpu.go:41-44, 61-64; pinned by prover/prover_cpu_test.go:7-25)
    if (!ctx) { prover::note_backend(1); return fail(DVN_ERR_UNSUPPORTED, "GPU prover not supported in this build"); }
    prover::Request q{curve, ctx, ccs, pk, witness, r_mont, s_mont, proof_out};
    return prover::GPUProver(&q);
} DVN_CATCH
extern "C" int32_t dvn_prover_cpu_prover(dvn_ctx* ctx, int32_t curve, const dvn_r1cs* ccs, dvn_groth16_pk* pk, const uint64_t* witness,
                                         const uint64_t* r_mont, const uint64_t* s_mont, uint64_t* proof_out) try {
    prover::Request q{curve, ctx, ccs, pk, witness, r_mont, s_mont, proof_out};
    return prover::CPUProver(&q);
} DVN_CATCH
extern "C" int32_t dvn_prover_setup(dvn_ctx* ctx, const dvn_r1cs* ccs, uint64_t seed, dvn_groth16_pk** pk) try { return prover::Setup(ctx, ccs, seed, pk); } DVN_CATCH
