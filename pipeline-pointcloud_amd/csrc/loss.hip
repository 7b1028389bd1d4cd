// Photometric loss of the 3DGS training step, fused: (1-l)*L1 + l*(1-SSIM) forward sums and
// the gradient image.  gfx950 only.
//
// Replaces the torch L1 + SSIM (torchmetrics in splatfacto, fused-ssim in gsplat's
// simple_trainer) that the reference reaches through main.py:1312 / main.py:1343
// (SURVEY.md 8a row a11).  SSIM: 11x11 Gaussian window, sigma 1.5, zero 'same' padding,
// C1 = 0.01^2, C2 = 0.03^2, mean over C*H*W*3.
//
// One 256-thread workgroup per 16x16 pixel tile and camera; the 26x26 halo of both images
// (3 channels) is staged in LDS once and the separable convolution runs out of LDS.
// Bound: HBM streaming, about 24 B read + 36 B written per pixel forward, 60 B read + 12 B
// written backward.
#include "common.h"

namespace {

constexpr int LT = 16;            // tile edge
constexpr int HALO = 5;
constexpr int LW = LT + 2 * HALO;  // 26
constexpr float C1 = 0.01f * 0.01f;
constexpr float C2 = 0.03f * 0.03f;

__constant__ float GW[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                             0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                             0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                             0.0075987582094967365f, 0.001028380123898387f};

__device__ __forceinline__ float block_sum(float v, float* lds4) {
    v = wave_sum_all(v);
    if (lane_id() == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return s;
}

// forward: loss sums + the three partial-derivative maps needed by the backward
__global__ __launch_bounds__(256) void loss_fwd_kernel(int H, int W, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, float* __restrict__ dm_dmu1,
                                                       float* __restrict__ dm_dsig1, float* __restrict__ dm_dsig12,
                                                       float* __restrict__ sums /* [0]=L1 sum, [1]=SSIM sum */) {
    __shared__ float t1[3][LW][LW + 1], t2[3][LW][LW + 1];
    __shared__ float xc[5][LW][LT + 1];
    __shared__ float red[4];
    int cam = blockIdx.z;
    int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const float* a = img1 + (size_t)cam * H * W * 3;
    const float* b = img2 + (size_t)cam * H * W * 3;
    for (int i = threadIdx.x; i < LW * LW; i += 256) {
        int r = i / LW, c = i - r * LW;
        int y = y0 + r - HALO, x = x0 + c - HALO;
        bool in = y >= 0 && y < H && x >= 0 && x < W;
        size_t p = ((size_t)y * W + x) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            t1[ch][r][c] = in ? a[p + ch] : 0.f;
            t2[ch][r][c] = in ? b[p + ch] : 0.f;
        }
    }
    __syncthreads();
    int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    int px = x0 + lx, py = y0 + ly;
    bool inside = px < W && py < H;
    float l1 = 0.f, ss = 0.f;
    for (int ch = 0; ch < 3; ch++) {
        for (int i = threadIdx.x; i < LW * LT; i += 256) {
            int r = i / LT, c = i - r * LT;
            float s1 = 0.f, s2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                float w = GW[k], u = t1[ch][r][c + k], v = t2[ch][r][c + k];
                s1 += w * u; s2 += w * v; s11 += w * u * u; s22 += w * v * v; s12 += w * u * v;
            }
            xc[0][r][c] = s1; xc[1][r][c] = s2; xc[2][r][c] = s11; xc[3][r][c] = s22; xc[4][r][c] = s12;
        }
        __syncthreads();
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            float w = GW[k];
            mu1 += w * xc[0][ly + k][lx]; mu2 += w * xc[1][ly + k][lx]; e11 += w * xc[2][ly + k][lx];
            e22 += w * xc[3][ly + k][lx]; e12 += w * xc[4][ly + k][lx];
        }
        __syncthreads();
        if (inside) {
            float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            float sg1 = e11 - mu1s, sg2 = e22 - mu2s, sg12 = e12 - mu12;
            float A = mu1s + mu2s + C1, B = sg1 + sg2 + C2, Cc = 2.f * mu12 + C1, D = 2.f * sg12 + C2;
            float m = (Cc * D) / (A * B);
            ss += m;
            float u = t1[ch][ly + HALO][lx + HALO], v = t2[ch][ly + HALO][lx + HALO];
            l1 += fabsf(u - v);
            size_t p = (((size_t)cam * H + py) * W + px) * 3 + ch;
            dm_dmu1[p] = (mu2 * 2.f * D) / (A * B) - (mu2 * 2.f * Cc) / (A * B) - (mu1 * 2.f * Cc * D) / (A * A * B) +
                         (mu1 * 2.f * Cc * D) / (A * B * B);
            dm_dsig1[p] = (-Cc * D) / (A * B * B);
            dm_dsig12[p] = (2.f * Cc) / (A * B);
        }
    }
    l1 = block_sum(l1, red);
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], l1);
        atomicAdd(&sums[1], ss);
    }
}

// backward: v_img1 = w_l1*sign(a-b) + w_ssim*(conv(dmu1) + 2a*conv(dsig1) + b*conv(dsig12))
__global__ __launch_bounds__(256) void loss_bwd_kernel(int H, int W, const float* __restrict__ img1,
                                                       const float* __restrict__ img2,
                                                       const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_dsig1,
                                                       const float* __restrict__ dm_dsig12, float w_l1, float w_ssim,
                                                       float* __restrict__ v_img1) {
    __shared__ float t[3][LW][LW + 1];
    __shared__ float xc[3][LW][LT + 1];
    int cam = blockIdx.z;
    int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    int px = x0 + lx, py = y0 + ly;
    bool inside = px < W && py < H;
    size_t base = (size_t)cam * H * W * 3;
    for (int ch = 0; ch < 3; ch++) {
        __syncthreads();
        for (int i = threadIdx.x; i < LW * LW; i += 256) {
            int r = i / LW, c = i - r * LW;
            int y = y0 + r - HALO, x = x0 + c - HALO;
            bool in = y >= 0 && y < H && x >= 0 && x < W;
            size_t p = base + ((size_t)y * W + x) * 3 + ch;
            t[0][r][c] = in ? dm_dmu1[p] : 0.f;
            t[1][r][c] = in ? dm_dsig1[p] : 0.f;
            t[2][r][c] = in ? dm_dsig12[p] : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < LW * LT; i += 256) {
            int r = i / LT, c = i - r * LT;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                float w = GW[k];
                s0 += w * t[0][r][c + k]; s1 += w * t[1][r][c + k]; s2 += w * t[2][r][c + k];
            }
            xc[0][r][c] = s0; xc[1][r][c] = s1; xc[2][r][c] = s2;
        }
        __syncthreads();
        if (inside) {
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                float w = GW[k];
                c0 += w * xc[0][ly + k][lx]; c1 += w * xc[1][ly + k][lx]; c2 += w * xc[2][ly + k][lx];
            }
            size_t p = base + ((size_t)py * W + px) * 3 + ch;
            float u = img1[p], v = img2[p];
            float d = u - v;
            float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            v_img1[p] = w_l1 * sgn + w_ssim * (c0 + 2.f * u * c1 + v * c2);
        }
    }
}

}  // namespace

// sums[2] must be zeroed by the caller (hipMemsetAsync on the same stream) before the call.
extern "C" int mi3dgs_loss_fwd(int C, int height, int width, const float* render, const float* target,
                               float* dm_dmu1, float* dm_dsigma1, float* dm_dsigma12, float* sums, void* stream) {
    MI_REQUIRE(C > 0 && height > 0 && width > 0, "loss_fwd: bad sizes");
    dim3 grid(mi_div_up(width, LT), mi_div_up(height, LT), C);
    MI_LAUNCH("loss_fwd", loss_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, height, width, render, target, dm_dmu1,
                       dm_dsigma1, dm_dsigma12, sums);
    MI_LAUNCH_CHECK();
    return 0;
}

// v_render = d/d(render) [ (1-l) * mean|r-t| + l * (1 - mean SSIM) ] * loss_scale
extern "C" int mi3dgs_loss_bwd(int C, int height, int width, const float* render, const float* target,
                               const float* dm_dmu1, const float* dm_dsigma1, const float* dm_dsigma12,
                               float ssim_lambda, float loss_scale, float* v_render, void* stream) {
    MI_REQUIRE(C > 0 && height > 0 && width > 0, "loss_bwd: bad sizes");
    float M = (float)C * (float)height * (float)width * 3.f;
    float w_l1 = loss_scale * (1.f - ssim_lambda) / M;
    float w_ssim = -loss_scale * ssim_lambda / M;
    dim3 grid(mi_div_up(width, LT), mi_div_up(height, LT), C);
    MI_LAUNCH("loss_bwd", loss_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, height, width, render, target, dm_dmu1,
                       dm_dsigma1, dm_dsigma12, w_l1, w_ssim, v_render);
    MI_LAUNCH_CHECK();
    return 0;
}
