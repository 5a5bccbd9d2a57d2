#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned lds[];
__global__ __launch_bounds__(512) void k(unsigned* out) { lds[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = lds[(threadIdx.x + 1) & 511]; }
int main() {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kb : {32, 48, 56, 64, 72, 76, 80, 96}) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k, 512, (size_t)kb * 1024);
        printf("lds %d KiB -> %d blocks/CU (%s)\n", kb, nb, hipGetErrorString(e));
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerBlock %zu\n", p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock);
}
