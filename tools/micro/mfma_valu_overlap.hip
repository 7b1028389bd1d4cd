// Does a v_mfma_f32_16x16x4_f32 stream of one wave run BESIDE the vector instructions of another wave on the same SIMD?
// Blocks of 512 threads = 8 waves = 2 per SIMD; per SIMD one wave runs role A, the other role B (0 idle, 1 VALU fma chain,
// 2 f32 MFMA 16x16x4, 3 f32 MFMA 32x32x2, 4 ds_write_b32 stream, 5 bf16 MFMA 32x32x16).  Time of (A, B) together against (A, idle), (idle, B).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed, float* lds) {
    float r = seed;
    if (ROLE == 1) {
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {        // 64 independent-ish fmas per iteration
                a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f);
                a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
                a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f);
                a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
            }
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (ROLE == 2) {
        f4v d0 = {0, 0, 0, 0}, d1 = d0;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {        // 16 MFMAs per iteration = 512 pipe cycles
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0001f, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.9999f, d1, 0, 0, 0);
            }
        }
        r = d0[0] + d1[1] + d0[2] + d1[3];
    } else if (ROLE == 3) {
        f16v d0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d1 = d0;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {        // 8 MFMAs per iteration = 512 pipe cycles
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0001f, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.9999f, d1, 0, 0, 0);
            }
        }
        r = d0[0] + d1[1] + d0[2] + d1[3];
    } else if (ROLE == 5) {
        f16v d0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d1 = d0;
        const u4v a = {0x3f803f80u + (unsigned)seed, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {        // 16 MFMAs per iteration = 512 pipe cycles
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, a), d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, a), d1, 0, 0, 0);
            }
        }
        r = d0[0] + d1[1] + d0[2] + d1[3];
    } else if (ROLE == 6 || ROLE == 7 || ROLE == 8) {
        // ONE wave alternating: an MFMA, then independent v_fma that fit into its pipe time (6 / 14 / 6)
        f4v d4 = {0, 0, 0, 0};
        f16v d16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const u4v a = {0x3f803f80u + (unsigned)seed, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < (ROLE == 7 ? 8 : 16); u++) {
                if (ROLE == 6) d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0001f, d4, 0, 0, 0);
                if (ROLE == 7) d16 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0001f, d16, 0, 0, 0);
                if (ROLE == 8) d16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, a), d16, 0, 0, 0);
                a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f);
                a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
                a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f);
                if (ROLE == 7) {
                    a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a0 = __builtin_fmaf(a0, 1.0001f, 0.5f);
                    a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f);
                    a3 = __builtin_fmaf(a3, 1.0001f, 0.5f); a4 = __builtin_fmaf(a4, 1.0001f, 0.5f);
                    a5 = __builtin_fmaf(a5, 1.0001f, 0.5f); a6 = __builtin_fmaf(a6, 1.0001f, 0.5f);
                }
                asm volatile("" : "+v"(a0), "+v"(a5));
            }
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + d4[0] + d16[0];
    } else if (ROLE == 4) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) lds[u * 68] = seed + (float)u;
            asm volatile("" ::: "memory");
        }
        r = lds[0];
    }
    return r;
}

template <int A, int B>
__global__ __launch_bounds__(512) void k(int iters, float* out) {
    __shared__ float lds[8][16 * 68];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // waves go to SIMDs in a cyclic order: waves w and w + 4 share a SIMD
    float r;
    if (wv < 4) r = run_role<A>(iters, (float)lane, &lds[wv][lane]);
    else r = run_role<B>(iters, (float)lane, &lds[wv][lane]);
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int A, int B>
float time_k(int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int it = 20000;
    printf("per iteration: VALU role = 64 v_fma (256 issue cycles), MFMA roles = 512 matrix-pipe cycles, LDS role = 16 ds_write_b32\n");
    printf("valu alone         %8.1f us\n", time_k<1, 0>(it, out));
    printf("valu + valu        %8.1f us\n", time_k<1, 1>(it, out));
    printf("mfma16 alone       %8.1f us\n", time_k<2, 0>(it, out));
    printf("mfma16 + mfma16    %8.1f us\n", time_k<2, 2>(it, out));
    printf("mfma16 + valu      %8.1f us\n", time_k<2, 1>(it, out));
    printf("mfma32 alone       %8.1f us\n", time_k<3, 0>(it, out));
    printf("mfma32 + valu      %8.1f us\n", time_k<3, 1>(it, out));
    printf("bf16 32x32x16 alone %8.1f us\n", time_k<5, 0>(it, out));
    printf("bf16 32x32x16 + valu %7.1f us\n", time_k<5, 1>(it, out));
    printf("ONE wave, f32 16x16x4 + 6 v_fma per gap (512 pipe cycles + 96 fma per iteration)  %8.1f us\n", time_k<6, 0>(it, out));
    printf("ONE wave, f32 32x32x2 + 14 v_fma per gap (512 pipe cycles + 112 fma per iteration) %8.1f us\n", time_k<7, 0>(it, out));
    printf("ONE wave, bf16 32x32x16 + 6 v_fma per gap (512 pipe cycles + 96 fma per iteration) %8.1f us\n", time_k<8, 0>(it, out));
    printf("that f32 16x16x4 wave + a valu wave  %8.1f us\n", time_k<6, 1>(it, out));
    printf("that bf16 wave + a valu wave         %8.1f us\n", time_k<8, 1>(it, out));
    printf("ldsw alone         %8.1f us\n", time_k<4, 0>(it, out));
    printf("ldsw + ldsw        %8.1f us\n", time_k<4, 4>(it, out));
    printf("ldsw + valu        %8.1f us\n", time_k<4, 1>(it, out));
    printf("ldsw + mfma16      %8.1f us\n", time_k<4, 2>(it, out));
    return 0;
}
