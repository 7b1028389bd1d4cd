// Hardware check of the MFMA evaluation used by csrc/rasterize_mfma.hip: log2 alpha of 256 pixels x 256 splats
// through quad_coefs -> v_mfma_f32_32x32x2_f32 x 3 -> v_permlane32_swap, against a float64 evaluation on the host.
//   hipcc --offload-arch=gfx950 -O3 -I pipeline-pointcloud_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 \
//         tools/micro/mfma_quad_test.hip -o /tmp/mfma_quad_test && /tmp/mfma_quad_test
#include "rasterize_mfma.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
int mi_set_error(const char*, hipError_t, const char*, int) { return 1; }
int mi_set_error_msg(const char*) { return 1; }
void mi_prof_begin(const char*, hipStream_t) {}
void mi_prof_end(hipStream_t) {}
int mi_rasterize_bwd_mm(int, int, int, int, int, const float*, const int32_t*, const int32_t*, const int32_t*, const float*, const float*,
                        const int32_t*, const float*, const float*, int, float*, int, hipStream_t) { return 1; }

using namespace mfma_raster;
__global__ __launch_bounds__(BLOCK) void probe(const float* splats, int tx, int ty, float* out /*[256 slots][256 px]*/) {
    __shared__ Staged L;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const float xc = tx * 16 + 8.f, yc = ty * 16 + 8.f;
    stage_splat(L, threadIdx.x, load_rec(splats, (int)threadIdx.x), xc, yc);
    __syncthreads();
    const Basis b = make_basis(wv, lane);
    int lx, ly;
    pixel_of_lane(wv, lane, lx, ly);
    for (int sb = 0; sb < 8; sb++) {
        float s[SUB];
        eval_sub_batch(L, sb, lane, b, s);
#pragma unroll
        for (int i = 0; i < SUB; i++) out[(sb * SUB + i) * 256 + ly * 16 + lx] = s[i];
    }
}

int main() {
    std::vector<float> sp(256 * SPLAT_STRIDE, 0.f);
    srand(1);
    auto rnd = [] { return rand() / (float)RAND_MAX; };
    const int tx = 3, ty = 2;
    for (int i = 0; i < 256; i++) {
        float* r = &sp[i * SPLAT_STRIDE];
        r[SP_X] = tx * 16 + 8 + (rnd() - 0.5f) * 60.f;
        r[SP_Y] = ty * 16 + 8 + (rnd() - 0.5f) * 60.f;
        float a = 0.02f + rnd() * 2.f, c = 0.02f + rnd() * 2.f;
        r[SP_CA] = a; r[SP_CC] = c; r[SP_CB] = (rnd() - 0.5f) * 1.8f * sqrtf(a * c);
        r[SP_OPA] = 0.01f + 0.98f * rnd();
    }
    float *d_sp, *d_out;
    hipMalloc(&d_sp, sp.size() * 4);
    hipMalloc(&d_out, 256 * 256 * 4);
    hipMemcpy(d_sp, sp.data(), sp.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d_sp, tx, ty, d_out);
    std::vector<float> out(256 * 256);
    hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst_rel = 0;
    int bad = 0;
    for (int s = 0; s < 256; s++)
        for (int p = 0; p < 256; p++) {
            const float* r = &sp[s * SPLAT_STRIDE];
            double px = tx * 16 + (p & 15) + 0.5, py = ty * 16 + (p >> 4) + 0.5;
            double dx = r[SP_X] - px, dy = r[SP_Y] - py;
            double sigma = 0.5 * (r[SP_CA] * dx * dx + r[SP_CC] * dy * dy) + r[SP_CB] * dx * dy;
            double ref = log2((double)r[SP_OPA]) - 1.4426950408889634 * sigma;
            double e = fabs(out[s * 256 + p] - ref);
            if (e > worst) worst = e;
            if (ref > -12 && e > worst_rel) worst_rel = e;      // where the pair can matter (alpha > 2^-12)
            if (e > 1e-2 * (1 + fabs(ref) * 1e-3)) bad++;
        }
    printf("max |log2 alpha error| over 65536 pairs: %.3g (pairs with alpha > 2^-12: %.3g); bad = %d\n", worst, worst_rel, bad);
    return bad != 0;
}
