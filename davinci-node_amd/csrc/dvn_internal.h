// Internal declarations shared by the translation units of libdvn_prover.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/dvn_prover.h"
#include "bn254_curve.h"
#include "host_fast.h"

using namespace dvn;

// ------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------
extern thread_local std::string tl_error;

// One in-flight MSM: its own stream, workspace and pinned result buffer.  Two lanes let the sort / latency-bound tail
// of one MSM run under the accumulate kernel of another (measured: +10 % at 2^22, +19 % at 2^20, +52 % at 2^18).
struct MsmLane {
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    void* ws = nullptr; size_t ws_bytes = 0;
    // caller-independent workspace for the allocation-free entry point: when set, the MSM carves its scratch out of this
    // fixed buffer instead of the grow-only `ws` (never hipMalloc / hipFree) and clears memory with a fill kernel
    void* ext_ws = nullptr; size_t ext_bytes = 0;
    u32* h_sums = nullptr;        // pinned host buffer for the per-window sums
    bool busy = false;
    int curve = 0, suite = 0; bool expanded = false; u32 c = 0, nwin = 0, W = 0; u32 seq = 0;
    u32 red_kh = 0, red_kl = 0;   // expanded sets: h_sums holds the pieces of the marginal-sum bucket reduction (msm_driver.h bitsum_combine)
    float ms[2] = {0.f, 0.f};
};
constexpr int DVN_LANES = 4;
constexpr size_t DVN_HSUMS_WORDS = 64 * 128;    // up to 64 windows of the widest xyzz image (BLS12-377 G2: 96 words)
// staged host-to-device copies (dvn_h2d): ring of pinned chunks; DVN_H2D_BUFS is a multiple of the thread count
constexpr int DVN_H2D_BUFS = 8, DVN_H2D_THREADS = 4;
constexpr size_t DVN_H2D_CHUNK = (size_t)4 << 20, DVN_H2D_MIN = (size_t)16 << 20;

struct dvn_ctx {
    int device = 0;
    MsmLane lane[DVN_LANES];
    hipEvent_t ev_dep = nullptr;
    bool dep_hold = false;    // lanes acquired while set wait for ev_dep as it was last recorded (not for later main-stream work)
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // 0/1: whole device section, 2/3: dominant kernel
    float last_ms[2] = {0.f, 0.f};
    void* ws = nullptr;       // grow-only workspace
    size_t ws_bytes = 0;
    std::mutex mu;
    bool force_fail = false;
    // this device's copy of the library's static scratch array (resolved once at dvn_init with hipGetSymbolAddress)
    void* static_ws = nullptr; size_t static_ws_bytes = 0;
    // dvn_init(devs, ndev) with ndev > 1: one sub-context per further device ordinal (devs[1..]); this context serves devs[0].
    // A sub-context is a complete context of its own (streams, lanes, workspaces on its device), owned by the primary.
    std::vector<dvn_ctx*> peers;
    dvn_ctx* primary = nullptr;          // set in sub-contexts
    // sha256 -> proving key loaded from an artifact file (dvn_groth16_pk_load_file); keys stay owned by the cache
    std::vector<std::pair<std::string, struct dvn_groth16_pk*>> key_cache;
    // large caller buffers (a witness, a scalar vector: pageable memory of the Go heap) go up through a ring of pinned chunks that
    // a few host threads fill and DMA drains on a stream of its own (dvn_h2d, dvn_core.hip); created at the first large copy
    void* h2d_pinned[DVN_H2D_BUFS] = {};
    hipEvent_t h2d_done[DVN_H2D_BUFS] = {};
    hipStream_t h2d_stream = nullptr;
    hipEvent_t h2d_ev = nullptr;
    int h2d_state = 0;                   // 0 = not set up yet, 1 = ready, -1 = unavailable (DVN_H2D_STAGED=0 or no pinned memory)
};
struct dvn_points {
    int suite = 0;            // 0 = BN254, 1 = BLS12-377
    int curve = 1;            // 1 = G1, 2 = G2
    size_t n = 0;
    u32* d = nullptr;         // internal affine image: nexp rows of n points; row w holds 2^(expand_c * w) * P_i
    u32 expand_c = 0;         // 0 = plain set (one row); otherwise the window size the rows were built for
    u32 nexp = 1;
};
// Default expansion policy (measured on MI355X, tools/gpu_expand_sweep.py, G1 and G2 alike): with the staged sort the
// one-bucket-set layout wins 10-30 % from 2^15 points up -- window 16 below 2^18, 17 below 2^20, 20 from there (18, 19 and
// 21 lose: their top window holds 2-7 bits of a 254-bit scalar and piles a quarter of all entries into a few buckets).
// Small sets (2^10 .. 2^15 points: commitment bases, small keys) take it too, with 15-bit windows: their cost is the latency of
// the bucket reduction, and ONE bucket set reduced by marginal sums needs 0.31-0.41 ms where the per-window running sums of the
// plain layout need 0.5-0.6 (tools/gpu_small_msm.py).
constexpr u32 DVN_EXPAND_MIN_LOG = 10;
// A set is left plain when its table would exceed DVN_MSM_EXPAND_MAX_GB (default 16 GiB; HBM is 288 GB, a Groth16 key has
// five sets).  DVN_MSM_EXPAND=0 disables expansion globally.
static inline void expand_plan(size_t n, u32 scalar_bits, size_t point_bytes, u32* c, u32* nexp) {
    static const int enabled = []() { const char* v = getenv("DVN_MSM_EXPAND"); return (v && v[0] == '0') ? 0 : 1; }();
    static const size_t max_bytes = []() { const char* v = getenv("DVN_MSM_EXPAND_MAX_GB"); return (size_t)(v ? atoi(v) : 16) << 30; }();
    *c = 0; *nexp = 1;
    if (!enabled) return;
    dvn_host::expand_plan_pure(n, scalar_bits, point_bytes, DVN_EXPAND_MIN_LOG, max_bytes, c, nexp);
}
struct dvn_domain {
    int suite = 0;            // 0 = BN254 Fr, 1 = BLS12-377 Fr
    u32 log_n = 0;
    u32* tw_fwd = nullptr;    // per-stage tables, n-1 entries of NL limbs in planes, entry (n - (n>>s)) + k = w^(k<<s)
    u32* tw_inv = nullptr;
    u32* tw_fwd_c = nullptr;  // the same with stage s scaled by g^(2^s): the DIT network evaluates on the coset g<w>
    u32* tw_inv_c = nullptr;  // w^-1 and g^-(2^s): the DIF network returns the inverse coset transform (times n)
    // coset power tables (exponent split lo/hi), internal form
    u32 lo_bits = 0;
    u32 *g_lo = nullptr, *g_hi = nullptr;            // g^j, g^(j<<lo_bits)
    u32 *gi_lo = nullptr, *gi_hi = nullptr;          // g^-j / n, g^-(j<<lo_bits)
    u32* card_inv = nullptr;                         // 1/n (one element)
    u32* h_consts = nullptr;                         // computeH: k1 = den * 2^5 / n^3, k2 = den / n^2
};

int32_t fail(int32_t code, const char* fmt, ...);
// maps the exception in flight to an error code + message (dvn_core.hip); every extern "C" entry point ends in DVN_CATCH
int32_t dvn_guard_fail() noexcept;
#define DVN_CATCH catch (...) { return dvn_guard_fail(); }

// rocprofv3 / roctracer ranges around the stages of a proof (H2D, gathers, each MSM, computeH and its transforms, collect):
// resolved with dlopen at first use and only when DVN_ROCTX=1, so the library has no link-time dependency on a profiler
// and the markers cost nothing otherwise.
void dvn_roctx_push(const char* name);
void dvn_roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char* name) { dvn_roctx_push(name); }
    ~RoctxRange() { dvn_roctx_pop(); }
    RoctxRange(const RoctxRange&) = delete;
};
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) {                                                                               \
            int32_t code = (_e == hipErrorOutOfMemory) ? DVN_ERR_OOM : DVN_ERR_DEVICE;                        \
            (void)hipGetLastError();                                                                          \
            return fail(code, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);      \
        }                                                                                                     \
    } while (0)
#define TRY(expr) do { int32_t _rc = (expr); if (_rc != DVN_OK) return _rc; } while (0)

// Host memory -> device memory, ordered on `consumer`: what `consumer` has queued before the call is finished before dst is
// written, what it queues afterwards sees the data.  Returns when the caller's buffer has been READ completely (it may be
// freed or reused: the cgo rule), not when the data has arrived.  Buffers of DVN_H2D_MIN bytes and more are staged through
// pinned chunks by DVN_H2D_THREADS host threads (measured on the MI355X box: pageable hipMemcpy 30 GB/s, pinned DMA 57 GB/s,
// a host copy on four threads 50 GB/s); smaller ones and a context without pinned memory take hipMemcpyAsync.
// The caller holds ctx->mu (one staged copy per context at a time).
int32_t dvn_h2d(dvn_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t consumer);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline int32_t enter(dvn_ctx* ctx) {
    if (!ctx) return fail(DVN_ERR_INVALID_ARG, "ctx is NULL");
    if (ctx->force_fail) return fail(DVN_ERR_INJECTED, "DVN_FORCE_FAIL is set: injected device failure");
    HIP_TRY(hipSetDevice(ctx->device));
    return DVN_OK;
}
// Zero `bytes` (a multiple of 4) at p (4-byte aligned) on a stream, with a kernel of small workgroups: the runtime's memset
// kernel wants a whole CU's worth of free wave slots and registers, so queued behind a long VALU-bound kernel on another stream
// it (and everything after it on its stream) waits until that kernel drains; 256-thread groups of a dozen registers slip in.
static __global__ void k_zero_fill(u32* __restrict__ p, size_t words) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t head = std::min<size_t>(words, ((16 - ((size_t)p & 15)) & 15) / 4);       // words before the first 16-byte boundary
    uint4* q = reinterpret_cast<uint4*>(p + head);
    const size_t n16 = (words - head) / 4;
    for (size_t i = t; i < n16; i += stride) q[i] = make_uint4(0, 0, 0, 0);
    if (t < head) p[t] = 0;
    const size_t tail0 = head + n16 * 4;
    if (t < words - tail0) p[tail0 + t] = 0;
}
static inline int32_t dvn_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return DVN_OK;
    const size_t words = bytes / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((words / 4 + 255) / 256 + 1, 4096);
    k_zero_fill<<<blocks, 256, 0, st>>>((u32*)p, words);
    HIP_TRY(hipGetLastError());
    return DVN_OK;
}
static inline int32_t ensure_ws(dvn_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return DVN_OK;
    if (ctx->ws) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    size_t want = align_up(bytes + bytes / 8, 1 << 20);
    HIP_TRY(hipMalloc(&ctx->ws, want));
    ctx->ws_bytes = want;
    return DVN_OK;
}
static inline int32_t lane_ensure_ws(MsmLane& L, size_t bytes) {
    if (L.ext_ws) {
        if (bytes > L.ext_bytes) return fail(DVN_ERR_INVALID_ARG, "MSM needs %zu bytes of scratch, the static scratch buffer holds %zu", bytes, L.ext_bytes);
        return DVN_OK;
    }
    if (bytes <= L.ws_bytes) return DVN_OK;
    if (L.ws) { HIP_TRY(hipStreamSynchronize(L.stream)); HIP_TRY(hipFree(L.ws)); L.ws = nullptr; L.ws_bytes = 0; }
    size_t want = align_up(bytes + bytes / 8, 1 << 20);
    HIP_TRY(hipMalloc(&L.ws, want));
    L.ws_bytes = want;
    return DVN_OK;
}
static inline void* lane_ws_base(const MsmLane& L) { return L.ext_ws ? L.ext_ws : L.ws; }
struct Carver {
    char* base; size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <class T> T* take(size_t count) { off = align_up(off, 256); T* p = (T*)(base + off); off += count * sizeof(T); return p; }
};
struct CarverSize {
    size_t off = 0;
    void take(size_t bytes) { off = align_up(off, 256); off += bytes; }
};

// ------------------------------------------------------------------------------------------------------
// host helpers on the shared arithmetic headers
// ------------------------------------------------------------------------------------------------------
static inline u64 splitmix_at(u64 seed, u64 index) {
    u64 z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static const u32 FR_MOD_W[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
static inline bool geq8(const u32* a, const u32* b) {
    for (int i = 7; i >= 0; --i) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; }
    return true;
}
static inline void sub8(u32* a, const u32* b) {
    u64 br = 0;
    for (int i = 0; i < 8; ++i) { u64 d = (u64)a[i] - b[i] - br; a[i] = (u32)d; br = (d >> 32) & 1; }
}
// i-th canonical Fr of SplitMix64 stream `seed` (oracle/bn254_py.py fr_at)
static inline void fr_at_words(u64 seed, u64 i, u32 out[8]) {
    for (int k = 0; k < 4; ++k) { u64 v = splitmix_at(seed, 4 * i + k); out[2 * k] = (u32)v; out[2 * k + 1] = (u32)(v >> 32); }
    while (geq8(out, FR_MOD_W)) sub8(out, FR_MOD_W);
}
template <class Cv> static inline typename Cv::X host_scalar_mul(const typename Cv::A& p, const u32 k[8]) {
    typename Cv::X acc = Cv::infinity();
    for (int b = 255; b >= 0; --b) {
        acc = Cv::dbl(acc);
        if ((k[b >> 5] >> (b & 31)) & 1) acc = Cv::add_mixed(acc, p);
    }
    return acc;
}
static inline G1::A g1_generator() {
    u32 one[8] = {1, 0, 0, 0, 0, 0, 0, 0}, two[8] = {2, 0, 0, 0, 0, 0, 0, 0};
    return G1::A{fe_from_canonical<FpCfg>(fe_unpack<FpCfg>(one)), fe_from_canonical<FpCfg>(fe_unpack<FpCfg>(two))};
}
static inline G2::A g2_generator() {   // EIP-197 generator, gnark Montgomery image (oracle/bn254_cpu.c G2_GEN_MONT)
    static const u64 g[16] = {
        0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL,
        0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL, 0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL,
        0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL,
        0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL};
    return G2::affine_from_ext((const u32*)g);
}


// resolves this device's address of the static scratch array (dvn_msm_g1.hip); called by dvn_init with the device current
int32_t dvn_internal_static_scratch_init(dvn_ctx* ctx);

// ---- internal cross-TU entry points (caller holds ctx->mu) -------------------------------------------------
// MSM with DEVICE scalars, result as a host (4x64-limb, R = 2^256) extended-Jacobian point (dvn_msm_g1.hip / dvn_msm_g2.hip)
int32_t dvn_internal_msm_g1_dev(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, dvn_host::HG1::X* out);
int32_t dvn_internal_msm_g2_dev(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, dvn_host::HG2::X* out);
// asynchronous halves of the two above (at most DVN_LANES MSMs in flight)
int32_t dvn_internal_msm_submit_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, MsmLane** lane);
int32_t dvn_internal_msm_submit_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, MsmLane** lane);
int32_t dvn_internal_msm_collect_g1(dvn_ctx* ctx, MsmLane* lane, dvn_host::HG1::X* out);
int32_t dvn_internal_msm_collect_g2(dvn_ctx* ctx, MsmLane* lane, dvn_host::HG2::X* out);
// rows [lo, hi) of an expanded set: one partial result per call (sharded Groth16)
int32_t dvn_internal_msm_share_submit_g1(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, u32 rank, u32 world, MsmLane** lane);
int32_t dvn_internal_msm_share_submit_g2(dvn_ctx* ctx, const dvn_points* pts, size_t off, size_t n, const u32* d_scalars, u32 rank, u32 world, MsmLane** lane);
// computeH on three device vectors of n = 2^log_n elements (already zero padded); result left in da (dvn_ntt.hip)
int32_t dvn_internal_compute_h_dev(dvn_ctx* ctx, const dvn_domain* dom, u32* da, u32* db, u32* dc);
uint32_t dvn_internal_domain_log_n(const dvn_domain* dom);
