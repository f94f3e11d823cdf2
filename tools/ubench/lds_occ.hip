// How many 512-thread blocks fit a CU for a given dynamic LDS size (hipOccupancyMaxActiveBlocksPerMultiprocessor).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_occ.hip -o tools/ubench/lds_occ.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(uint32_t *o)
{
    extern __shared__ uint32_t s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    o[threadIdx.x] = s[(threadIdx.x * 7) & 511];
}
int main()
{
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int bytes = 52 * 1024; bytes <= 56 * 1024; bytes += 128) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 512, bytes);
        printf("%d B -> %d blocks\n", bytes, n);
    }
    return 0;
}
