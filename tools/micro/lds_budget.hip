// How many workgroups of T threads with X bytes of LDS does a CU hold?  (runtime's occupancy answer + a measured one)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_budget.hip -o /tmp/lds_budget && /tmp/lds_budget
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
extern __shared__ char dyn[];
__global__ void k(unsigned long long* t, int spin) {
    if (threadIdx.x == 0) t[2 * blockIdx.x] = wall_clock64();
    dyn[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) t[2 * blockIdx.x + 1] = wall_clock64() + dyn[(threadIdx.x + 1) & 63];
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d, sharedMemPerBlock %zu, sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.multiProcessorCount,
           p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    unsigned long long* t; hipMalloc(&t, 16 * 8192);
    std::vector<unsigned long long> h(2 * 8192);
    for (int threads : {256, 512, 1024}) for (int kb : {20, 26, 32, 33, 39, 40, 48, 50, 53, 58, 64, 67, 72, 76, 78, 80}) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k, threads, (size_t)kb * 1024);
        // measured: launch 8 blocks per CU, spin 20 us each; blocks whose start lies within 5 us of the first = resident together
        int blocks = p.multiProcessorCount * 8;
        hipMemset(t, 0, 16 * 8192);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), (size_t)kb * 1024, 0, t, 2000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), t, 16 * blocks, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull; for (int b = 0; b < blocks; b++) t0 = std::min(t0, h[2 * b]);
        int first = 0; for (int b = 0; b < blocks; b++) if (h[2 * b] - t0 < 500) first++;
        printf("threads %4d lds %2d KB: runtime says %d blocks/CU, measured %.2f blocks/CU in the first generation\n", threads, kb, nb,
               (double)first / p.multiProcessorCount);
    }
    return 0;
}
