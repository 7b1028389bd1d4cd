// LDS-array cost of the operations the rasterisers lean on, per wave-instruction, with 4 waves per CU (one per SIMD) and 12.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_ops.hip -o tools/micro/lds_ops && tools/micro/lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>

struct alignas(16) F4 { float x, y, z, w; };
struct F3 { float x, y, z; };

template <int OP>
__global__ __launch_bounds__(768) void k(int iters, int active, float* out) {
    __shared__ F4 buf[12][800];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* f = reinterpret_cast<float*>(&buf[wv][0]);
    float acc = 0.f;
    const float v = (float)lane;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) { if (lane < active) atomicAdd(&f[u * 68 + lane], v); }                    // ds_add_f32, lanes on distinct banks
            if (OP == 1) { if (lane < active) atomicAdd(&f[(lane & 3) * 12 * 4 + (lane >> 2) + u * 200], v); }   // 4 rows x columns, as the flush slots
            if (OP == 2) { asm volatile("" ::: "memory"); const F4 t = buf[wv][u]; acc += t.x + t.y + t.z + t.w; }          // ds_read_b128, broadcast
            if (OP == 3) { asm volatile("" ::: "memory"); const F3 t = *reinterpret_cast<const F3*>(&f[u * 4]); acc += t.x + t.y + t.z; }                              // 3 x ds_read_b32 (or b32 + read2), broadcast
            if (OP == 4) { f[u * 196 + lane] = v; }                                                  // ds_write_b32
            if (OP == 5) { asm volatile("" ::: "memory"); const F4 t = buf[wv][(lane & 15) * 17 + (lane >> 4) + u]; acc += t.x + t.y + t.z + t.w; }  // ds_read_b128, per-lane rows
            if (OP == 6) { asm volatile("" ::: "memory"); if (lane < active) f[u * 68 + lane] += v; }                              // read-modify-write without atomics
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int OP>
void run(const char* name, int waves, int active, float* out) {
    const int it = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(64 * waves), 0, 0, it, active, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(64 * waves), 0, 0, it, active, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // cycles of the CU's LDS per wave-instruction at 2.4 GHz
    printf("%-44s waves/CU %2d active %2d: %7.1f us  = %6.1f cycles per wave-instruction (CU-wide)\n", name, waves, active, ms * 1e3,
           ms * 1e-3 * 2.4e9 / ((double)it * 16 * waves));
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4096);
    for (int waves : {4, 12}) {
        for (int active : {64, 18, 9, 1}) run<0>("ds_add_f32 distinct banks", waves, active, out);
        run<1>("ds_add_f32 4 rows x 5 columns", waves, 18, out);
        run<6>("plain read-modify-write", waves, 18, out);
        run<2>("ds_read_b128 broadcast", waves, 64, out);
        run<3>("3 dwords broadcast (b32s)", waves, 64, out);
        run<4>("ds_write_b32", waves, 64, out);
        run<5>("ds_read_b128 per-lane rows", waves, 64, out);
    }
    return 0;
}
