// libdvn_prover.so -- context lifetime, memory helpers, error plumbing (include/dvn_prover.h).
// Host side of the C-ABI boundary; stands where gnark's icicle backend stands behind
// /root/reference/prover/prover_gpu.go:24-61.
#include <dlfcn.h>
#include "dvn_internal.h"
#include <stdexcept>
#include <system_error>
#include <atomic>
#include <thread>

thread_local std::string tl_error;
// ---- profiler markers (DVN_ROCTX=1): roctxRangePushA / roctxRangePop of whichever roctx library is installed -------------
namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)();
struct Roctx {
    roctx_push_t push = nullptr; roctx_pop_t pop = nullptr;
    Roctx() {
        const char* v = getenv("DVN_ROCTX");
        if (!v || v[0] != '1') return;
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (roctx_push_t)dlsym(h, "roctxRangePushA"); pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void dvn_roctx_push(const char* name) { Roctx& r = roctx(); if (r.push) (void)r.push(name); }
void dvn_roctx_pop() { Roctx& r = roctx(); if (r.pop) (void)r.pop(); }

int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    try { tl_error = buf; } catch (...) { /* out of memory while recording the message: the code still says what happened */ }
    return code;
}

// The one place where a C++ exception becomes an error code: every extern "C" entry point that returns a code is a
// function-try-block ending in DVN_CATCH (dvn_internal.h), so nothing thrown below it -- std::bad_alloc from a vector or
// string, std::system_error from a thread or std::async that cannot start -- unwinds into the C (cgo, ctypes) caller.
int32_t dvn_guard_fail() noexcept {
    try { throw; }
    catch (const std::bad_alloc&) { return fail(DVN_ERR_OOM, "host allocation failed (std::bad_alloc)"); }
    catch (const std::system_error& e) { return fail(DVN_ERR_OOM, "host resource exhausted: %s", e.what()); }
    catch (const std::exception& e) { return fail(DVN_ERR_DEVICE, "internal error: %s", e.what()); }
    catch (...) { return fail(DVN_ERR_DEVICE, "internal error: unknown exception"); }
}
// Test hook (host only, no context needed): throws the exception class `kind` names from inside a guarded entry point, so that
// the mapping above can be exercised without exhausting memory.  0: nothing, 1: bad_alloc, 2: system_error, 3: runtime_error, 4: int.
extern "C" int32_t dvn_test_raise(int32_t kind) try {
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "thread could not start");
    if (kind == 3) throw std::runtime_error("injected");
    if (kind == 4) throw 4;
    return DVN_OK;
} DVN_CATCH

// ------------------------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------------------------
extern "C" const char* dvn_version(void) { return "dvn-prover 0.1 (gfx950, BN254, 9x29-bit Montgomery)"; }
extern "C" const char* dvn_last_error(const dvn_ctx*) { return tl_error.c_str(); }

static int32_t ctx_create(int dev, dvn_ctx** out) {
    dvn_ctx* ctx = new (std::nothrow) dvn_ctx();
    if (!ctx) return fail(DVN_ERR_OOM, "host allocation failed");
    ctx->device = dev;
    const char* ff = getenv("DVN_FORCE_FAIL");
    ctx->force_fail = ff && (ff[0] == '1' || ff[0] == 'y' || ff[0] == 't');
    hipError_t e2 = hipSetDevice(dev);
    if (e2 == hipSuccess) e2 = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e2 == hipSuccess; ++i) e2 = hipEventCreate(&ctx->ev[i]);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&ctx->ev_dep, hipEventDisableTiming);
    for (int l = 0; l < DVN_LANES && e2 == hipSuccess; ++l) {
        MsmLane& L = ctx->lane[l];
        e2 = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
        for (int i = 0; i < 4 && e2 == hipSuccess; ++i) e2 = hipEventCreate(&L.ev[i]);
        if (e2 == hipSuccess) e2 = hipHostMalloc((void**)&L.h_sums, DVN_HSUMS_WORDS * 4, hipHostMallocDefault);
    }
    if (e2 != hipSuccess) { delete ctx; (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "context setup failed on device %d: %s", dev, hipGetErrorString(e2)); }
    { int32_t rc = dvn_internal_static_scratch_init(ctx); if (rc != DVN_OK) { dvn_shutdown(ctx); return rc; } }
    *out = ctx;
    return DVN_OK;
}
// devs / ndev: every listed ordinal gets a context.  The handle returned serves devs[0]; the contexts of devs[1..] hang off it
// (dvn_ctx_device) and are what the *_multi entry points spread one proof over.  An ordinal may repeat (two contexts sharing
// one GPU): how a one-GPU box rehearses the multi-device path.
extern "C" int32_t dvn_init(const int* devs, int ndev, dvn_ctx** out) try {
    if (!out) return fail(DVN_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (ndev < 0 || ndev > 64 || (ndev > 0 && !devs)) return fail(DVN_ERR_INVALID_ARG, "bad device list");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) { (void)hipGetLastError(); return fail(DVN_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e)); }
    const int n = ndev > 0 ? ndev : 1;
    for (int i = 0; i < n; ++i) {
        const int dev = ndev > 0 ? devs[i] : 0;
        if (dev < 0 || dev >= count) return fail(DVN_ERR_NO_DEVICE, "device ordinal %d out of range (%d devices)", dev, count);
    }
    dvn_ctx* primary = nullptr;
    TRY(ctx_create(ndev > 0 ? devs[0] : 0, &primary));
    for (int i = 1; i < n; ++i) {
        dvn_ctx* sub = nullptr;
        int32_t rc = ctx_create(devs[i], &sub);
        if (rc != DVN_OK) { std::string keep = tl_error; dvn_shutdown(primary); tl_error = keep; return rc; }
        sub->primary = primary;
        primary->peers.push_back(sub);
    }
    (void)hipSetDevice(primary->device);
    *out = primary;
    return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_ctx_device_count(const dvn_ctx* ctx) try { return ctx ? 1 + (int32_t)ctx->peers.size() : 0; } DVN_CATCH
extern "C" dvn_ctx* dvn_ctx_device(dvn_ctx* ctx, int32_t index) {
    if (!ctx || index < 0 || index > (int32_t)ctx->peers.size()) { (void)fail(DVN_ERR_INVALID_ARG, "device index %d outside the context's %d devices", index, ctx ? 1 + (int)ctx->peers.size() : 0); return nullptr; }
    return index == 0 ? ctx : ctx->peers[(size_t)index - 1];
}
extern "C" int32_t dvn_ctx_device_ordinal(const dvn_ctx* ctx) try { return ctx ? ctx->device : -1; } DVN_CATCH
extern "C" const char* dvn_error_string(int32_t code) {
    switch (code) {
        case DVN_OK: return "ok";
        case DVN_ERR_INVALID_ARG: return "invalid argument";
        case DVN_ERR_DEVICE: return "HIP call failed";
        case DVN_ERR_NO_DEVICE: return "no usable gfx950 device";
        case DVN_ERR_OOM: return "out of memory";
        case DVN_ERR_UNSUPPORTED: return "unsupported";
        case DVN_ERR_INJECTED: return "injected failure (DVN_FORCE_FAIL)";
        default: return "unknown error code";
    }
}
// ---- staged host-to-device copies (declared in dvn_internal.h) ----------------------------------------------------------
static bool h2d_ready(dvn_ctx* ctx) {
    if (ctx->h2d_state != 0) return ctx->h2d_state > 0;
    ctx->h2d_state = -1;
    const char* v = getenv("DVN_H2D_STAGED");
    if (v && v[0] == '0') return false;
    bool ok = hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking) == hipSuccess
              && hipEventCreateWithFlags(&ctx->h2d_ev, hipEventDisableTiming) == hipSuccess;
    for (int b = 0; ok && b < DVN_H2D_BUFS; ++b)
        ok = hipHostMalloc(&ctx->h2d_pinned[b], DVN_H2D_CHUNK, hipHostMallocDefault) == hipSuccess
             && hipEventCreateWithFlags(&ctx->h2d_done[b], hipEventDisableTiming) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return false; }             // what was created is released with the context
    ctx->h2d_state = 1;
    return true;
}
static void h2d_release(dvn_ctx* ctx) {
    if (ctx->h2d_stream) (void)hipStreamSynchronize(ctx->h2d_stream);
    for (int b = 0; b < DVN_H2D_BUFS; ++b) {
        if (ctx->h2d_pinned[b]) (void)hipHostFree(ctx->h2d_pinned[b]);
        if (ctx->h2d_done[b]) (void)hipEventDestroy(ctx->h2d_done[b]);
    }
    if (ctx->h2d_ev) (void)hipEventDestroy(ctx->h2d_ev);
    if (ctx->h2d_stream) (void)hipStreamDestroy(ctx->h2d_stream);
}
int32_t dvn_h2d(dvn_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t consumer) {
    if (bytes == 0) return DVN_OK;
    if (bytes < DVN_H2D_MIN || !h2d_ready(ctx)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, consumer));      // pageable source: returns once it is staged
        return DVN_OK;
    }
    static_assert(DVN_H2D_BUFS % DVN_H2D_THREADS == 0, "a ring buffer must always be refilled by the thread that filled it before");
    // the copy stream starts behind what the consumer has queued so far (earlier readers and writers of dst)
    HIP_TRY(hipEventRecord(ctx->h2d_ev, consumer));
    HIP_TRY(hipStreamWaitEvent(ctx->h2d_stream, ctx->h2d_ev, 0));
    const size_t nchunk = (bytes + DVN_H2D_CHUNK - 1) / DVN_H2D_CHUNK;
    std::atomic<int> err{(int)hipSuccess};
    auto work = [&](int tid) {
        hipError_t e = hipSetDevice(ctx->device);
        for (size_t i = (size_t)tid; e == hipSuccess && i < nchunk && err.load() == (int)hipSuccess; i += DVN_H2D_THREADS) {
            const int b = (int)(i % DVN_H2D_BUFS);
            const size_t off = i * DVN_H2D_CHUNK, len = std::min(DVN_H2D_CHUNK, bytes - off);
            e = hipEventSynchronize(ctx->h2d_done[b]);              // the chunk this buffer carried before has left it (also a previous call's)
            if (e != hipSuccess) break;
            memcpy(ctx->h2d_pinned[b], (const char*)src + off, len);
            e = hipMemcpyAsync((char*)dst + off, ctx->h2d_pinned[b], len, hipMemcpyHostToDevice, ctx->h2d_stream);
            if (e == hipSuccess) e = hipEventRecord(ctx->h2d_done[b], ctx->h2d_stream);
        }
        if (e != hipSuccess) err.store((int)e);
    };
    std::thread th[DVN_H2D_THREADS - 1];
    int started = 0;
    try {
        for (; started < DVN_H2D_THREADS - 1; ++started) th[started] = std::thread(work, started + 1);
    } catch (...) { /* no more threads: this one takes their shares below */ }
    work(0);
    for (int t = started; t < DVN_H2D_THREADS - 1; ++t) work(t + 1);
    for (int t = 0; t < started; ++t) th[t].join();
    if (err.load() != (int)hipSuccess) {
        (void)hipGetLastError();
        return fail(DVN_ERR_DEVICE, "staged host-to-device copy failed: %s", hipGetErrorString((hipError_t)err.load()));
    }
    HIP_TRY(hipEventRecord(ctx->h2d_ev, ctx->h2d_stream));
    HIP_TRY(hipStreamWaitEvent(consumer, ctx->h2d_ev, 0));
    return DVN_OK;
}

int32_t dvn_internal_key_cache_clear(dvn_ctx* ctx);      // dvn_artifacts.hip
extern "C" void dvn_shutdown(dvn_ctx* ctx) {
    if (!ctx) return;
    for (dvn_ctx* sub : ctx->peers) { sub->primary = nullptr; dvn_shutdown(sub); }
    ctx->peers.clear();
    (void)hipSetDevice(ctx->device);
    (void)dvn_internal_key_cache_clear(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    h2d_release(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (int l = 0; l < DVN_LANES; ++l) {
        MsmLane& L = ctx->lane[l];
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        if (L.ws) (void)hipFree(L.ws);
        if (L.h_sums) (void)hipHostFree(L.h_sums);
        for (int i = 0; i < 4; ++i) if (L.ev[i]) (void)hipEventDestroy(L.ev[i]);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (ctx->ev_dep) (void)hipEventDestroy(ctx->ev_dep);
    for (int i = 0; i < 4; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
extern "C" int32_t dvn_malloc(dvn_ctx* ctx, size_t bytes, void** dptr) try {
    TRY(enter(ctx));
    if (!dptr) return fail(DVN_ERR_INVALID_ARG, "dptr is NULL");
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_free(dvn_ctx* ctx, void* dptr) try { TRY(enter(ctx)); if (dptr) HIP_TRY(hipFree(dptr)); return DVN_OK; } DVN_CATCH
extern "C" int32_t dvn_memcpy_h2d(dvn_ctx* ctx, void* dst, const void* src, size_t bytes) try {
    TRY(enter(ctx)); HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_memcpy_d2h(dvn_ctx* ctx, void* dst, const void* src, size_t bytes) try {
    TRY(enter(ctx)); HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return DVN_OK;
} DVN_CATCH
extern "C" int32_t dvn_synchronize(dvn_ctx* ctx) try { TRY(enter(ctx)); HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipDeviceSynchronize()); return DVN_OK; } DVN_CATCH
extern "C" int32_t dvn_last_kernel_ms(dvn_ctx* ctx, int which, float* ms) try {
    if (!ctx || !ms || which < 0 || which > 1) return fail(DVN_ERR_INVALID_ARG, "bad argument");
    *ms = ctx->last_ms[which];
    return DVN_OK;
} DVN_CATCH



// ------------------------------------------------------------------------------------------------------
// Bench-only: this GPU's v_mad_u64_u32 issue rate, measured in the run that quotes it.
// The hot kernels of the library (bucket accumulation, NTT butterflies) are bound by the issue of this one instruction; their
// "fraction of the integer peak" is only meaningful against the rate of the box they ran on (it differs by ~20 % between
// boxes and with clocks).  Every SIMD is saturated (8 waves each) with 8 independent multiply-add chains per lane; the rate
// is taken from the difference of a long and a short launch, so launch overhead cancels.
// ------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_mad_chain(u64* __restrict__ out, u32 seed, int rounds) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[8]; u32 w[8];
    u32 m = seed * 2654435761u + t;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = t + i; w[i] = t + 7 * i + 1; }
    for (int it = 0; it < 2048 * rounds; ++it) {
        asm volatile("" : "+v"(m));                    // opaque: the recurrence has no closed form for the compiler
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(w[i]), "v"(m) : "vcc");
    }
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
    out[t] = r;
}
extern "C" int32_t dvn_bench_mad_rate(dvn_ctx* ctx, double* lane_ops_per_s) try {
    TRY(enter(ctx));
    if (!lane_ops_per_s) return fail(DVN_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    const int blocks = prop.multiProcessorCount * 8;           // 8 blocks of 4 waves per CU: 8 waves per SIMD
    u64* out = nullptr;
    HIP_TRY(hipMalloc((void**)&out, (size_t)blocks * 256 * 8));
    float ms[2] = {0.f, 0.f};
    hipError_t e = hipSuccess;
    for (int r = 0; r < 2 && e == hipSuccess; ++r) {
        const int rounds = r ? 9 : 1;
        k_mad_chain<<<blocks, 256, 0, ctx->stream>>>(out, 1, rounds);            // warm
        e = hipEventRecord(ctx->ev[0], ctx->stream);
        k_mad_chain<<<blocks, 256, 0, ctx->stream>>>(out, 2, rounds);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev[1], ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms[r], ctx->ev[0], ctx->ev[1]);
    }
    (void)hipFree(out);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(DVN_ERR_DEVICE, "mad-rate probe failed: %s", hipGetErrorString(e)); }
    const double lane_ops = (double)blocks * 256.0 * 2048.0 * 8.0 /* rounds difference */ * 8.0 /* chains */;
    *lane_ops_per_s = lane_ops / ((double)(ms[1] - ms[0]) * 1e-3);
    return DVN_OK;
} DVN_CATCH
