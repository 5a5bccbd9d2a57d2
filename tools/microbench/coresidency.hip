// Does a memory-bound kernel make progress BESIDE a register-heavy VALU-bound one (three waves of ~156 VGPRs per SIMD, a grid
// many times the chip, i.e. the shape of the MSM's bucket accumulation)?  Depends on the workgroup shape of the memory-bound
// kernel: a 1024-thread group needs four free wave slots AND 4 x its registers on every SIMD of one CU, a 256-thread group of
// <= 40 registers fits into what three 156-register waves leave free (512 - 468 = 44).
// Prints, for both shapes: each kernel alone, and both started together on two streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

// 72 independent 64-bit accumulators per lane: ~150 VGPRs, three waves per SIMD
__global__ __launch_bounds__(128) void k_valu_fat(u64* out, unsigned seed, int iters) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[72]; unsigned m = seed * 2654435761u + t;
#pragma unroll
    for (int i = 0; i < 72; ++i) a[i] = t + i;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(m));
#pragma unroll
        for (int i = 0; i < 72; ++i) a[i] = (u64)(unsigned)a[i] * m + a[i];
    }
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < 72; ++i) r += a[i];
    out[t] = r;
}
template <int T> __global__ __launch_bounds__(T) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n; i += (size_t)gridDim.x * T) dst[i] = src[i];
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    uint4 *src, *dst; u64* out;
    const int vblocks = cus * 6 * 8;                     // 8 times what is resident at three waves per SIMD (6 two-wave groups per CU)
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, (size_t)vblocks * 128 * 8));
    CK(hipMemset(src, 1, bytes));
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    const int iters = 400;
    auto run = [&](const char* what, bool valu, int copy_threads) -> int {
        CK(hipDeviceSynchronize());
        if (valu) { CK(hipEventRecord(e0, sa)); k_valu_fat<<<vblocks, 128, 0, sa>>>(out, 1, iters); CK(hipEventRecord(e1, sa)); }
        if (copy_threads) {
            CK(hipEventRecord(f0, sb));
            for (int r = 0; r < 2; ++r) {
                if (copy_threads == 256) k_copy<256><<<8192, 256, 0, sb>>>(src, dst, n16);
                else k_copy<1024><<<2048, 1024, 0, sb>>>(src, dst, n16);
            }
            CK(hipEventRecord(f1, sb));
        }
        CK(hipDeviceSynchronize());
        float tv = 0, tc = 0;
        if (valu) CK(hipEventElapsedTime(&tv, e0, e1));
        if (copy_threads) CK(hipEventElapsedTime(&tc, f0, f1));
        printf("%-40s valu %7.3f ms   copy(2 x 1 GiB, %4d-thread groups) %7.3f ms\n", what, tv, copy_threads, tc); fflush(stdout);
        return 0;
    };
    for (int rep = 0; rep < 2; ++rep) {
        if (run("valu alone", true, 0)) return 1;
        if (run("copy alone, 256", false, 256)) return 1;
        if (run("copy alone, 1024", false, 1024)) return 1;
        if (run("both, copy in 256-thread groups", true, 256)) return 1;
        if (run("both, copy in 1024-thread groups", true, 1024)) return 1;
    }
    printf("done\n");
    return 0;
}
