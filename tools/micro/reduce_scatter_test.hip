// Standalone check of the wave64 8-value reduce-scatter used by rasterize_bwd.
// build: hipcc -O3 --offload-arch=gfx950 reduce_scatter_test.hip -o reduce_scatter_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "../../pipeline-pointcloud_amd/csrc/common.h"
int mi_set_error(const char*, hipError_t, const char*, int) { return 1; }
int mi_set_error_msg(const char*) { return 2; }
void mi_prof_begin(const char*, hipStream_t) {}
void mi_prof_end(hipStream_t) {}

__global__ void k(float* o8, float* o9, const float* in) {
    float v[9];
    for (int i = 0; i < 9; i++) v[i] = in[threadIdx.x * 9 + i];
    wave_reduce_scatter8_plus1(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]);
    o8[threadIdx.x] = v[0];
    o9[threadIdx.x] = v[8];
}
int main() {
    float h[64 * 9], *d, *o8, *o9, r8[64], r9[64];
    double sum[9] = {0};
    for (int l = 0; l < 64; l++) for (int i = 0; i < 9; i++) { h[l * 9 + i] = (float)((l * 7 + i * 13) % 31) + 0.25f * i; sum[i] += h[l * 9 + i]; }
    hipMalloc(&d, sizeof(h)); hipMalloc(&o8, 256); hipMalloc(&o9, 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o8, o9, d);
    hipMemcpy(r8, o8, 256, hipMemcpyDeviceToHost); hipMemcpy(r9, o9, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) if (fabs(r8[l] - sum[l >> 3]) > 1e-3) { bad++; if (bad < 6) printf("lane %d got %f want %f\n", l, r8[l], sum[l >> 3]); }
    if (fabs(r9[63] - sum[8]) > 1e-3) { bad++; printf("v8 lane63 got %f want %f\n", r9[63], sum[8]); }
    printf(bad ? "FAIL %d\n" : "PASS: lane l holds the total of value l>>3; value 8 in lane 63\n", bad);
    return bad != 0;
}
