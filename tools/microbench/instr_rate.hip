// Instruction-rate microbenchmark for gfx950: which multiplier primitive should the
// BN254 Montgomery multiplication be built on?  Not part of the product library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int CH = 8;      // independent chains per lane
constexpr int UNROLL = 16; // ops per chain per loop iteration

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc[CH];
    double   dac[CH];
    uint32_t a = seed * 2654435761u + tid * 40503u + 1u;
    uint32_t b = seed ^ (tid * 2246822519u) | 1u;
    double da = 1.0 + (double)(tid & 1023) * 1e-9, db = 0.999999 + (double)(seed & 7) * 1e-9;
    for (int c = 0; c < CH; ++c) { acc[c] = a + c * 977u; dac[c] = 1.0 + c; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (OP == 0) {        // v_mad_u64_u32
                    acc[c] = (uint64_t)(uint32_t)acc[c] * b + acc[c];
                } else if constexpr (OP == 1) { // v_mul_lo_u32
                    acc[c] = (uint32_t)acc[c] * b;
                } else if constexpr (OP == 2) { // v_mul_hi_u32
                    acc[c] = __umulhi((uint32_t)acc[c], b) | 0x80000001u;
                } else if constexpr (OP == 3) { // v_fma_f64
                    dac[c] = __builtin_fma(dac[c], db, da);
                } else if constexpr (OP == 4) { // v_mad_u32_u24
                    acc[c] = ((uint32_t)acc[c] & 0xffffffu) * (b & 0xffffffu) + 7u;
                } else if constexpr (OP == 5) { // v_add_co / v_addc (64-bit add)
                    acc[c] = acc[c] + (((uint64_t)b << 32) | a) + (acc[c] >> 63);
                } else if constexpr (OP == 6) { // v_fma_f32
                    float f = __builtin_fmaf((float)dac[c], (float)db, (float)da); dac[c] = f;
                } else if constexpr (OP == 7) { // 32-bit add
                    acc[c] = (uint32_t)acc[c] + b;
                } else if constexpr (OP == 8) { // v_mul_hi_u32_u24 + v_mul_u32_u24
                    uint32_t x = (uint32_t)acc[c] & 0xffffffu;
                    uint64_t pr = (uint64_t)x * (uint64_t)(b & 0xffffffu); acc[c] = (uint32_t)(pr >> 32) + (uint32_t)pr;
                }
            }
        }
    }
    uint64_t r = 0;
    for (int c = 0; c < CH; ++c) r ^= acc[c] ^ (uint64_t)__double_as_longlong(dac[c]);
    if (r == 0x123456789abcdefull) out[tid] = r; // practically never; keeps the chains live
}

// f32 chain variant kept separate so the double<->float converts do not pollute OP 6
__global__ __launch_bounds__(256) void rate_f32(float* out, float seed, int iters) {
    float acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = seed + c;
    float b = 0.99999f, a = 1e-7f * threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_fmaf(acc[c], b, a);
    float r = 0; for (int c = 0; c < CH; ++c) r += acc[c];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int OP>
int run(const char* name, int waves_per_simd) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd; // 256 threads = 4 waves = 1 wave per SIMD per block
    uint64_t* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    rate_kernel<OP><<<blocks, 256>>>(out, 12345, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, 256>>>(out, 12345, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 256 * (double)iters * UNROLL * CH;
    double per_cu_per_ns = ops / (ms * 1e6) / cus;
    printf("%-28s waves/SIMD=%d  %8.3f ms  %8.2f Glane-op/s  %6.2f lane-op/ns/CU  (@2.4GHz: %5.2f cyc per wave64-instr per SIMD)\n",
           name, waves_per_simd, ms, ops / (ms * 1e6), per_cu_per_ns,
           (4.0 * 64.0) / (per_cu_per_ns / 2.4));
    CHECK(hipFree(out));
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    for (int w : {1, 2, 4}) {
        run<0>("v_mad_u64_u32", w);
        run<1>("v_mul_lo_u32", w);
        run<2>("v_mul_hi_u32(+or)", w);
        run<3>("v_fma_f64", w);
        run<4>("v_mul_u32_u24+add+shift", w);
        run<5>("add64 x2 + shift", w);
        run<7>("v_add_u32", w);
        run<8>("mulhi_u24+mul_u24+add+and", w);
    }
    {
        float* o; CHECK(hipMalloc(&o, 1024));
        int cus = prop.multiProcessorCount;
        for (int w : {1, 2, 4}) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            rate_f32<<<cus * w, 256>>>(o, 1.0f, 10); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            rate_f32<<<cus * w, 256>>>(o, 1.0f, 2000);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double ops = (double)cus * w * 256 * 2000.0 * UNROLL * CH;
            double pc = ops / (ms * 1e6) / cus;
            printf("%-28s waves/SIMD=%d  %8.3f ms  %8.2f Glane-op/s  %6.2f lane-op/ns/CU  (@2.4GHz: %5.2f cyc per wave64-instr per SIMD)\n",
                   "v_fma_f32", w, ms, ops / (ms * 1e6), pc, 256.0 / (pc / 2.4));
        }
    }
    return 0;
}
