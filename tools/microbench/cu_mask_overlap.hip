// Can a memory-bound kernel run beside a VALU-bound one when each has its own CUs?  Two streams created with
// hipExtStreamCreateWithCUMask: `few` CUs for a streaming copy, the rest for an integer multiply-add loop.  The few-CU mask
// selects the same number of CUs in every XCD whether mask bit i means (XCD i % 8) or (XCD i / 32).
// Prints each kernel's time alone and the time of both together, masked and unmasked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

__global__ __launch_bounds__(128) void k_valu(u64* out, unsigned seed, int iters) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[8]; unsigned m = seed * 2654435761u + t;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = t + i;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(m));
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (u64)(unsigned)a[i] * m + a[(i + 1) & 7];
    }
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
    out[t] = r;
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv) {
    const int per_xcd = argc > 1 ? atoi(argv[1]) : 2;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs; %d CUs per XCD for the copy stream\n", p.name, cus, per_xcd); fflush(stdout);
    if (cus != 256 || per_xcd < 1 || per_xcd > 3) { printf("unexpected shape\n"); return 1; }
    std::vector<uint32_t> few(8, 0), rest(8, 0);
    for (int x = 0; x < 8; ++x) for (int j = 0; j < per_xcd; ++j) few[x] |= 1u << (x + 8 * j);
    for (int x = 0; x < 8; ++x) rest[x] = ~few[x];
    hipStream_t s_few, s_rest, s_a, s_b;
    CK(hipExtStreamCreateWithCUMask(&s_few, 8, few.data()));
    CK(hipExtStreamCreateWithCUMask(&s_rest, 8, rest.data()));
    CK(hipStreamCreateWithFlags(&s_a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_b, hipStreamNonBlocking));
    uint32_t got[8]; CK(hipExtStreamGetCUMask(s_few, 8, got));
    printf("few mask read back: "); for (int i = 0; i < 8; ++i) printf("%08x ", got[i]); printf("\n"); fflush(stdout);
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    uint4 *src, *dst; u64* out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, (size_t)cus * 24 * 128 * 8));
    CK(hipMemset(src, 1, bytes));
    const int vblocks = cus * 24, iters = 6000;
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    auto run = [&](const char* what, hipStream_t sv, hipStream_t sc, bool valu, bool copy) -> int {
        CK(hipDeviceSynchronize());
        if (valu) { CK(hipEventRecord(e0, sv)); k_valu<<<vblocks, 128, 0, sv>>>(out, 1, iters); CK(hipEventRecord(e1, sv)); }
        if (copy) { CK(hipEventRecord(f0, sc)); for (int r = 0; r < 4; ++r) k_copy<<<4096, 256, 0, sc>>>(src, dst, n16); CK(hipEventRecord(f1, sc)); }
        CK(hipDeviceSynchronize());
        float tv = 0, tc = 0;
        if (valu) CK(hipEventElapsedTime(&tv, e0, e1));
        if (copy) CK(hipEventElapsedTime(&tc, f0, f1));
        printf("%-28s valu %7.3f ms   copy(4 x 1 GiB) %7.3f ms = %6.0f GB/s\n", what, tv, tc, copy ? 8.0 * bytes / tc / 1e6 : 0.0); fflush(stdout);
        return 0;
    };
    for (int rep = 0; rep < 2; ++rep) {
        if (run("unmasked: valu alone", s_a, s_b, true, false)) return 1;
        if (run("unmasked: copy alone", s_a, s_b, false, true)) return 1;
        if (run("unmasked: both", s_a, s_b, true, true)) return 1;
        if (run("masked: valu alone (rest)", s_rest, s_few, true, false)) return 1;
        if (run("masked: copy alone (few)", s_rest, s_few, false, true)) return 1;
        if (run("masked: both", s_rest, s_few, true, true)) return 1;
    }
    printf("done\n");
    return 0;
}
