// Issue cost of the instructions a wide-integer multiplication can be built from, in cycles per wave64 instruction per
// SIMD with every SIMD saturated: v_mad_u64_u32, v_fma_f64, v_add_f64, v_lshl_add_u64, v_mul_lo_u32, v_mul_hi_u32,
// v_mad_u32_u24, v_add3_u32 (VOP3 encodings) and v_add_u32, v_and_b32, v_lshrrev_b32 (VOP2 encodings).  Each kernel runs ITER rounds of 8 independent chains of one operation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
constexpr int ITER = 4096;   // per unit of `rounds`

template <int OP> __global__ __launch_bounds__(256) void k(u64* out, u32 seed, int rounds) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[8]; double d[8]; u32 w[8];
    u32 m = seed * 2654435761u + t, m2 = m ^ 0x9E3779B9u;
    double fm = (double)(m & 0xFFFFF) + 1.0, fc = 4503599627370496.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = t + i; d[i] = (double)(t + i); w[i] = t + i; }
    for (int it = 0; it < ITER * rounds; ++it) {
        asm volatile("" : "+v"(m), "+v"(m2));          // opaque: no closed form for the recurrence
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(w[i]), "v"(m) : "vcc");
            if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(fm), "v"(fc));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(fc));
            if (OP == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[i]) : "v"(m));
            if (OP == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(w[i]) : "v"(m));
            if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(w[i]) : "v"(m));
            if (OP == 7) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(m));
            if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(w[i]) : "v"(m));
            if (OP == 9) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(a[i]));
            if (OP == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(fm));
            if (OP == 11) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(w[i]) : "v"(m));           // VOP2 encodings
            if (OP == 12) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(w[i]) : "v"(m));
            if (OP == 13) asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(w[i]));
            if (OP == 15) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(m));
        }
    }
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + (u64)d[i] + w[i];
    out[t] = r;
}

template <int OP> static void run(const char* name, u64* out, int cus, double mhz) {
    const int blocks = cus * 8;                      // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int r = 0; r < 2; ++r) {
        const int rounds = r ? 9 : 1;
        k<OP><<<blocks, 256>>>(out, 1, rounds); hipDeviceSynchronize();
        hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 2, rounds); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[r], e0, e1);
    }
    const double insts_per_simd = 8.0 /*waves*/ * ITER * 8 * 8 /*rounds difference*/;
    printf("%-16s %8.3f / %8.3f ms -> %.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", name, ms[0], ms[1],
           (ms[1] - ms[0]) * 1e-3 * mhz * 1e6 / insts_per_simd, mhz);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1000.0;
    printf("%s: %d CUs, %.0f MHz\n", p.name, cus, mhz);
    u64* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    run<0>("mad_u64_u32", out, cus, mhz); run<1>("fma_f64", out, cus, mhz); run<2>("add_f64", out, cus, mhz); run<3>("lshl_add_u64", out, cus, mhz);
    run<4>("mul_lo_u32", out, cus, mhz); run<5>("mul_hi_u32", out, cus, mhz); run<6>("mad_u32_u24", out, cus, mhz); run<7>("add3_u32", out, cus, mhz);
    run<8>("mul_hi_u32_u24", out, cus, mhz); run<9>("lshrrev_b64", out, cus, mhz); run<10>("mul_f64", out, cus, mhz);
    run<11>("add_u32 (VOP2)", out, cus, mhz); run<12>("and_b32 (VOP2)", out, cus, mhz); run<13>("lshrrev_b32 (VOP2)", out, cus, mhz);
    run<15>("fma_f32", out, cus, mhz); run<0>("mad_u64_u32 again", out, cus, mhz);
    hipFree(out); return 0;
}
