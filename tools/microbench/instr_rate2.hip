// Second instruction-rate microbenchmark: the non-multiply instructions of the 29-bit Montgomery multiplication.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int CH = 8, UNROLL = 16;
template <int OP> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc[CH]; uint32_t b = seed ^ (tid * 2246822519u) | 1u, sh = (seed & 3) + 27;
    for (int c = 0; c < CH; ++c) acc[c] = ((uint64_t)(tid + c * 977u) << 20) | seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                uint64_t x = acc[c];
                if constexpr (OP == 0) { asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(x) : "v"(x)); }
                else if constexpr (OP == 1) { asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(x) : "v"(x), "v"(acc[(c + 1) % CH])); }
                else if constexpr (OP == 2) { uint32_t lo = (uint32_t)x, r; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(b)); x = r; }
                else if constexpr (OP == 3) { uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32), r; asm volatile("v_alignbit_b32 %0, %1, %2, 29" : "=v"(r) : "v"(hi), "v"(lo)); x = ((uint64_t)hi << 32) | r; }
                else if constexpr (OP == 4) { uint32_t lo = (uint32_t)x, r; asm volatile("v_and_b32 %0, 0x1fffffff, %1" : "=v"(r) : "v"(lo)); x = r; }
                else if constexpr (OP == 5) { uint32_t lo = (uint32_t)x, r; asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(b), "v"(sh)); x = r; }
                else if constexpr (OP == 6) { uint64_t r; uint32_t lo = (uint32_t)x; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(b), "v"(x) : "vcc"); x = r; }
                else if constexpr (OP == 7) { uint32_t lo = (uint32_t)x, r; asm volatile("v_ashrrev_i32 %0, 29, %1" : "=v"(r) : "v"(lo)); x = r; }
                else if constexpr (OP == 8) { uint32_t lo = (uint32_t)x, r; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(lo), "v"(b) : ); x = r; }
                else if constexpr (OP == 9) { uint32_t lo = (uint32_t)x, r; asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(b)); x = r; }
                else if constexpr (OP == 10) { uint32_t lo = (uint32_t)x, r; asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(b), "v"(sh)); x = r; }
                acc[c] = x;
            }
        }
    }
    uint64_t r = 0; for (int c = 0; c < CH; ++c) r ^= acc[c];
    if (r == 0x123456789abcdefull) out[tid] = r;
}
template <int OP> int run(const char* name) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount, blocks = cus * 4;
    uint64_t* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, 256>>>(out, 12345, 10); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(out, 12345, 2000); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 256 * 2000.0 * UNROLL * CH, pc = ops / (ms * 1e6) / cus;
    printf("%-22s 4 waves/SIMD %8.3f ms  %6.2f lane-op/ns/CU  (@2.4GHz: %5.2f cyc per wave64-instr per SIMD)\n", name, ms, pc, 256.0 / (pc / 2.4));
    CHECK(hipFree(out)); return 0;
}
int main() {
    run<6>("v_mad_u64_u32"); run<0>("v_lshrrev_b64"); run<1>("v_lshl_add_u64"); run<2>("v_mul_lo_u32"); run<3>("v_alignbit_b32");
    run<4>("v_and_b32"); run<5>("v_add3_u32"); run<7>("v_ashrrev_i32"); run<8>("v_cndmask_b32"); run<9>("v_mul_u32_u24"); run<10>("v_mad_u32_u24");
    return 0;
}
