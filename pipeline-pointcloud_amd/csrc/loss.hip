// Photometric loss of the 3DGS training step, fused: (1-l)*L1 + l*(1-SSIM) forward sums and
// the gradient image.  gfx950 only.
//
// Replaces the torch L1 + SSIM (torchmetrics in splatfacto, fused-ssim in gsplat's
// simple_trainer) that the reference reaches through main.py:1312 / main.py:1343
// (SURVEY.md 8a row a11).  SSIM: 11x11 Gaussian window, sigma 1.5, zero 'same' padding,
// C1 = 0.01^2, C2 = 0.03^2, mean over C*H*W*3.
//
// One 256-thread workgroup per 32x32 pixel tile and camera; the 42x42 halo of both images
// (3 channel planes) is staged in LDS once and the separable convolution runs out of LDS
// with register sliding windows (4 outputs per thread and pass).
// Algorithmic bytes: 24 B read + 36 B written per pixel forward, 60 B read + 12 B written
// backward; the kernels are LDS/VALU-bound, not HBM-bound.
#include "common.h"

namespace {

constexpr int HALO = 5;
constexpr float C1 = 0.01f * 0.01f;
constexpr float C2 = 0.03f * 0.03f;

__constant__ float GW[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                             0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                             0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                             0.0075987582094967365f, 0.001028380123898387f};

// One block per 32 x 32 tile (rounds 1-2; still the product kernels, see the streaming variant below).
constexpr int LT = 32;             // output tile edge
constexpr int LW = LT + 2 * HALO;  // 42 staged rows / columns
constexpr int LS = 45;             // staged row stride: odd, so 4 rows x 8 column groups of a read hit 32 banks
constexpr int HS = LT + 1;         // row stride of the horizontally blurred planes, same reason
constexpr int NT = 512;            // threads per block: LDS allows 2 blocks/CU, so waves come from block size
constexpr int VO = LT * LT / NT;   // output rows per thread in the vertical pass
constexpr int NM = 4;              // blurred moments of the forward: u, v, u^2 + v^2, u v
constexpr int ROW3 = LW * 3;       // floats of one staged image row in memory (interleaved RGB)

__device__ __forceinline__ float block_sum(float v, float* lds4) {
    v = wave_sum_all(v);
    if (lane_id() == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) s += lds4[i];
    __syncthreads();
    return s;
}

// Staging of the 42x42 halo region of an interleaved [H,W,3] image into 3 channel planes (zero
// padded), in two steps: ALL global loads of a thread first (into registers), the LDS stores after.
// The one-step version compiled to load -> s_waitcnt vmcnt(0) -> ds_write per element, i.e. eleven
// full memory latencies in series per image and block.
// Thread -> (row group, float of the row): 126 floats of a staged row fit 128 lanes, so a thread keeps ONE column (its
// channel, its LDS column and its bounds test against the image width are computed once) and walks rows rgrp, rgrp + 4, ...:
// per element an add, two compares and the load.  The first version flattened (row, float) into one index and paid a division
// by 126, a division by 3 and five compares per element: 17 % of loss_fwd's and 40 % of loss_bwd's vector instructions.
constexpr int ST_COLS = 128;                             // lanes per staged row (126 used)
constexpr int ST_RGRP = NT / ST_COLS;                    // rows in flight per iteration
constexpr int ST_ITERS = (LW + ST_RGRP - 1) / ST_RGRP;   // iterations per image
static_assert(ROW3 <= ST_COLS && NT % ST_COLS == 0, "staging layout");

struct StageCol { int j, rgrp, xj, col, ch; bool xok; };

// the target image comes as float32 in [0, 1] or straight from the uint8 image cache (value * scale, scale = 1 / 255: the same
// product mi3dgs_image_u8_to_f32 forms, so both routes give the same bits; the conversion pass it saves was a launch of its own
// in every training step)
__device__ __forceinline__ float px_load(const float* p, int off, float) { return p[off]; }
__device__ __forceinline__ float px_load(const uint8_t* p, int off, float scale) { return (float)p[off] * scale; }

__device__ __forceinline__ StageCol stage_col(int W, int x0) {
    StageCol c;
    c.j = (int)threadIdx.x & (ST_COLS - 1);
    c.rgrp = (int)threadIdx.x / ST_COLS;
    c.xj = (x0 - HALO) * 3 + c.j;                        // float index inside the image row: 3 x + channel
    c.col = c.j / 3;
    c.ch = c.j - 3 * c.col;
    c.xok = c.j < ROW3 && c.xj >= 0 && c.xj < W * 3;
    return c;
}

template <typename PT>
__device__ __forceinline__ void stage_load(const PT* __restrict__ img, int H, int W, int y0, const StageCol& c,
                                           float (&v)[ST_ITERS], float scale = 1.f) {
#pragma unroll
    for (int it = 0; it < ST_ITERS; it++) {
        const int y = y0 + it * ST_RGRP + c.rgrp - HALO;
        const bool ok = c.xok && y >= 0 && y < H;
        const int off = ok ? y * (W * 3) + c.xj : 0;      // 32-bit offset from a uniform base: saddr addressing, one VGPR
        const float val = px_load(img, off, scale);
        v[it] = ok ? val : 0.f;
    }
}

__device__ __forceinline__ void stage_store(float (*plane)[LW][LS], const StageCol& c, const float (&v)[ST_ITERS]) {
#pragma unroll
    for (int it = 0; it < ST_ITERS; it++) {
        const int r = it * ST_RGRP + c.rgrp;
        if (c.j < ROW3 && r < LW) plane[c.ch][r][c.col] = v[it];
    }
}

// Forward.  One 256-thread block per 32x32 output tile and camera.  Per channel:
//   horizontal pass: task (row r of 42, group of 4 columns) slides an 11-tap window over 14
//     staged values -> 4 outputs x 4 moments (u, v, uu + vv, uv), written to hz[4][42][32];
//   vertical pass: thread (column, group of 4 rows) slides over 14 rows of hz -> 4 pixels.
// The first version (16x16 tiles, one output per thread, 55 LDS reads per pixel-channel in
// the vertical pass, 2-way bank conflicts on half its LDS cycles) took 232 us; the halo
// overhead drops from 2.6x to 1.7x and LDS reads per output from 77 to 25.
template <typename TT>
__global__ __launch_bounds__(NT, 4) void loss_fwd_kernel(int H, int W, const float* __restrict__ img1,
                                                       const TT* __restrict__ img2, float t_scale, float* __restrict__ dm_dmu1,
                                                       float* __restrict__ dm_dsig1, float* __restrict__ dm_dsig12,
                                                       float* __restrict__ sums /* [0]=L1 sum, [1]=SSIM sum */) {
    __shared__ float pu[3][LW][LS], pv[3][LW][LS];
    __shared__ float hz[NM][LW][HS];
    __shared__ float red[NT / 64];
    const int cam = blockIdx.z;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const size_t base = (size_t)cam * H * W * 3;
    {
        const StageCol sc = stage_col(W, x0);
        float ra[ST_ITERS], rb[ST_ITERS];
        stage_load(img1 + base, H, W, y0, sc, ra);
        stage_load(img2 + base, H, W, y0, sc, rb, t_scale);
        stage_store(pu, sc, ra);
        stage_store(pv, sc, rb);
    }
    __syncthreads();
    const int vc = threadIdx.x & 31, vg = threadIdx.x >> 5;      // vertical pass: column, row group
    float l1 = 0.f, ss = 0.f;
    for (int ch = 0; ch < 3; ch++) {
        for (int t = threadIdx.x; t < LW * (LT / 4); t += NT) {
            int r = t >> 3, c0 = (t & 7) * 4;
            float u[14], v[14], sq[14], uv[14];
#pragma unroll
            for (int k = 0; k < 14; k++) { u[k] = pu[ch][r][c0 + k]; v[k] = pv[ch][r][c0 + k]; }
            // the products once per staged value, not once per (output, tap) pair; SSIM and its three derivative maps only
            // ever use sigma1^2 + sigma2^2, so u^2 + v^2 is blurred as ONE plane (four moments instead of five)
#pragma unroll
            for (int k = 0; k < 14; k++) { sq[k] = __builtin_fmaf(u[k], u[k], v[k] * v[k]); uv[k] = u[k] * v[k]; }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                float s1 = 0.f, s2 = 0.f, ssq = 0.f, s12 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const float w = GW[k];
                    s1 += w * u[o + k]; s2 += w * v[o + k]; ssq += w * sq[o + k]; s12 += w * uv[o + k];
                }
                hz[0][r][c0 + o] = s1; hz[1][r][c0 + o] = s2; hz[2][r][c0 + o] = ssq; hz[3][r][c0 + o] = s12;
            }
        }
        __syncthreads();
        {
            float acc[VO][NM];
#pragma unroll
            for (int o = 0; o < VO; o++)
#pragma unroll
                for (int q = 0; q < NM; q++) acc[o][q] = 0.f;
#pragma unroll
            for (int k = 0; k < VO + 10; k++) {
                float h[NM];
#pragma unroll
                for (int q = 0; q < NM; q++) h[q] = hz[q][vg * VO + k][vc];
#pragma unroll
                for (int o = 0; o < VO; o++) {
                    int tap = k - o;
                    if (tap >= 0 && tap < 11) {
#pragma unroll
                        for (int q = 0; q < NM; q++) acc[o][q] += GW[tap] * h[q];
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < VO; o++) {
                int px = x0 + vc, py = y0 + vg * VO + o;
                if (px < W && py < H) {
                    float mu1 = acc[o][0], mu2 = acc[o][1];
                    float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
                    float sg_sum = (acc[o][2] - mu1s) - mu2s, sg12 = acc[o][3] - mu12;      // sigma1^2 + sigma2^2, sigma12
                    float A = mu1s + mu2s + C1, B = sg_sum + C2, Cc = 2.f * mu12 + C1, D = 2.f * sg12 + C2;
                    float rAB = 1.f / (A * B);
                    ss += Cc * D * rAB;
                    float a = pu[ch][vg * VO + o + HALO][vc + HALO], b = pv[ch][vg * VO + o + HALO][vc + HALO];
                    l1 += fabsf(a - b);
                    size_t p = base + ((size_t)py * W + px) * 3 + ch;
                    dm_dmu1[p] = 2.f * rAB * (mu2 * (D - Cc) + mu1 * Cc * D * (1.f / B - 1.f / A));
                    dm_dsig1[p] = -Cc * D * rAB / B;
                    dm_dsig12[p] = 2.f * Cc * rAB;
                }
            }
        }
        __syncthreads();
    }
    if (sums == nullptr) return;             // the caller does not read the loss value of this step (block-uniform)
    l1 = block_sum(l1, red);
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], l1);
        atomicAdd(&sums[1], ss);
    }
}

// backward: v_img1 = w_l1*sign(a-b) + w_ssim*(conv(dmu1) + 2a*conv(dsig1) + b*conv(dsig12))
template <typename TT>
__global__ __launch_bounds__(NT) void loss_bwd_kernel(int H, int W, const float* __restrict__ img1,
                                                       const TT* __restrict__ img2, float t_scale,
                                                       const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_dsig1,
                                                       const float* __restrict__ dm_dsig12, float w_l1, float w_ssim,
                                                       float* __restrict__ v_img1) {
    __shared__ float pl[3][LW][LS];          // one map, 3 channels
    __shared__ float hz[3][3][LW][HS];       // [map][channel] horizontally blurred
    const int cam = blockIdx.z;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const size_t base = (size_t)cam * H * W * 3;
    // every global read of the block goes out first: the three derivative maps and the two image
    // values of each output pixel
    const StageCol sc = stage_col(W, x0);
    float rm[3][ST_ITERS];
    stage_load(dm_dmu1 + base, H, W, y0, sc, rm[0]);
    stage_load(dm_dsig1 + base, H, W, y0, sc, rm[1]);
    stage_load(dm_dsig12 + base, H, W, y0, sc, rm[2]);
    const int vc = threadIdx.x & 31, vg = threadIdx.x >> 5;
    float pu_[3][VO], pv_[3][VO];
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
#pragma unroll
        for (int o = 0; o < VO; o++) {
            int px = x0 + vc, py = y0 + vg * VO + o;
            bool in = px < W && py < H;
            int off = ((in ? py : 0) * W + (in ? px : 0)) * 3 + ch;
            pu_[ch][o] = (img1 + base)[off];
            pv_[ch][o] = px_load(img2 + base, off, t_scale);
        }
#pragma unroll
    for (int m = 0; m < 3; m++) {
        __syncthreads();
        stage_store(pl, sc, rm[m]);
        __syncthreads();
        for (int t = threadIdx.x; t < 3 * LW * (LT / 4); t += NT) {
            int ch = t / (LW * (LT / 4));
            int tt = t - ch * (LW * (LT / 4));
            int r = tt >> 3, c0 = (tt & 7) * 4;
            float u[14];
#pragma unroll
            for (int k = 0; k < 14; k++) u[k] = pl[ch][r][c0 + k];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                float s0 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) s0 += GW[k] * u[o + k];
                hz[m][ch][r][c0 + o] = s0;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float acc[VO][3];
#pragma unroll
        for (int o = 0; o < VO; o++) { acc[o][0] = 0.f; acc[o][1] = 0.f; acc[o][2] = 0.f; }
#pragma unroll
        for (int k = 0; k < VO + 10; k++) {
            float h0 = hz[0][ch][vg * VO + k][vc], h1 = hz[1][ch][vg * VO + k][vc], h2 = hz[2][ch][vg * VO + k][vc];
#pragma unroll
            for (int o = 0; o < VO; o++) {
                int tap = k - o;
                if (tap >= 0 && tap < 11) { acc[o][0] += GW[tap] * h0; acc[o][1] += GW[tap] * h1; acc[o][2] += GW[tap] * h2; }
            }
        }
#pragma unroll
        for (int o = 0; o < VO; o++) {
            int px = x0 + vc, py = y0 + vg * VO + o;
            if (px < W && py < H) {
                size_t p = base + ((size_t)py * W + px) * 3 + ch;
                float u = pu_[ch][o], v = pv_[ch][o];
                float d = u - v;
                float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                v_img1[p] = w_l1 * sgn + w_ssim * (acc[o][0] + 2.f * u * acc[o][1] + v * acc[o][2]);
            }
        }
    }
}

#ifdef MI3DGS_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------------
// Wave-private strips (round 3; experiments build, MI3DGS_LOSS_STREAM=1, until it beats the tile kernels).
// An image row is a 1-D array of W * 3 floats (interleaved RGB): the 11-tap blur along x touches floats c - 15, c - 12, ..., c + 15
// whatever channel c is, and the blur along y stays inside one float-column.  ONE WAVE owns 126 float-columns and walks down a
// strip of R rows; nothing is shared between waves, so there is no block barrier anywhere:
//   staging: 39 lanes load 16 bytes of each image row, form (u, v, u^2 + v^2, u v) per float and store them as float4s in the
//     wave's LDS row (LDS serves a wave in order: the row is complete before the wave's own reads of it);
//   horizontal: a lane owns TWO outputs of one channel at neighbouring pixels (floats c and c + 3) and reads the 12 taps they
//     share once;
//   vertical, without LDS: the horizontally blurred moments are added, weighted, to the eleven output rows they belong to
//     (2 x 44 register accumulators, shifted by one row per input row -- for free, an FMA names its destination); the output row
//     that just received its last tap is finished: SSIM and its three derivative maps, stored.
// First attempt of the round (one output per thread, 256-thread blocks, one barrier per row): 69 / 65 us against the tile
// kernels' 73 / 51 (profiles/r03_loss_stream_ab.txt) -- every address, wait and barrier of a row was paid per output.
constexpr int SHALO = 15;                      // 5 pixels x 3 interleaved channels
constexpr int WO = 2;                          // outputs per lane
constexpr int WCOLS = 63 * WO;                 // float-columns per wave (lane 63 idles: 63 = 21 pixel pairs x 3 channels)
constexpr int WROW = WCOLS + 2 * SHALO;        // 156 staged floats per row
constexpr int WLOAD = WROW / 4;                // 39 lanes load a float4 each
static_assert(WROW % 4 == 0, "staging by float4");
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte load from a 4-byte-aligned address

struct WaveStrip {
    int lane, c0, y_in0, n_out, n_in, sc, ob;
    bool s_on, s_vec, o_on[WO];
};

__device__ __forceinline__ WaveStrip wave_strip(int H, int W3, int R) {
    WaveStrip g;
    g.lane = threadIdx.x;
    g.c0 = blockIdx.x * WCOLS;
    const int y0 = blockIdx.y * R;
    g.y_in0 = y0 - HALO;
    g.n_out = min(R, H - y0);
    g.n_in = g.n_out + 2 * HALO;
    g.sc = g.c0 - SHALO + 4 * g.lane;                       // first of the four floats this lane stages
    g.s_on = g.lane < WLOAD;
    g.s_vec = g.s_on && g.sc >= 0 && g.sc + 3 < W3;
    const int grp = g.lane / 3;
    g.ob = 6 * grp + (g.lane - 3 * grp);                    // first output float, relative to c0; taps: staged floats ob + 3 m
#pragma unroll
    for (int o = 0; o < WO; o++) g.o_on[o] = g.lane < 63 && g.c0 + g.ob + 3 * o < W3;
    return g;
}

// the four floats of input row i this lane stages (zero outside the image: 'same' zero padding)
__device__ __forceinline__ float4 strip_load4(const float* __restrict__ img, const WaveStrip& g, int H, int W3, int i) {
    const int y = g.y_in0 + i;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < g.n_in && y >= 0 && y < H) {                    // wave-uniform
        const float* p = img + (size_t)y * W3;
        if (g.s_vec) {
            const f4u v = *reinterpret_cast<const f4u*>(p + g.sc);
            r = make_float4(v.x, v.y, v.z, v.w);
        } else if (g.s_on) {                                // the lanes at the row's ends
            if (g.sc >= 0 && g.sc < W3) r.x = p[g.sc];
            if (g.sc + 1 >= 0 && g.sc + 1 < W3) r.y = p[g.sc + 1];
            if (g.sc + 2 >= 0 && g.sc + 2 < W3) r.z = p[g.sc + 2];
            if (g.sc + 3 >= 0 && g.sc + 3 < W3) r.w = p[g.sc + 3];
        }
    }
    return r;
}

__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // ordering only: LDS serves a wave's accesses in order
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float rcp_newton(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.f), r);
}

__device__ __forceinline__ void fma4(float4& a, float w, const float4& x) {
    a.x = __builtin_fmaf(w, x.x, a.x); a.y = __builtin_fmaf(w, x.y, a.y); a.z = __builtin_fmaf(w, x.z, a.z); a.w = __builtin_fmaf(w, x.w, a.w);
}

__global__ __launch_bounds__(64) void loss_fwd_wave_kernel(int H, int W3, int R, const float* __restrict__ img1,
                                                           const float* __restrict__ img2, float* __restrict__ dm_dmu1,
                                                           float* __restrict__ dm_dsig1, float* __restrict__ dm_dsig12,
                                                           float* __restrict__ sums) {
    __shared__ float4 rowbuf[2][WROW];
    const size_t base = (size_t)blockIdx.z * H * W3;
    const float* a1 = img1 + base;
    const float* a2 = img2 + base;
    const WaveStrip g = wave_strip(H, W3, R);
    float4 acc[11][WO];
#pragma unroll
    for (int j = 0; j < 11; j++)
#pragma unroll
        for (int o = 0; o < WO; o++) acc[j][o] = make_float4(0.f, 0.f, 0.f, 0.f);
    float l1 = 0.f, ss = 0.f;
    float4 u4 = strip_load4(a1, g, H, W3, 0), v4 = strip_load4(a2, g, H, W3, 0);
    for (int i = 0; i < g.n_in; i++) {
        {
            {
            float4* row = rowbuf[i & 1];
            if (g.s_on) {
                row[4 * g.lane + 0] = make_float4(u4.x, v4.x, __builtin_fmaf(u4.x, u4.x, v4.x * v4.x), u4.x * v4.x);
                row[4 * g.lane + 1] = make_float4(u4.y, v4.y, __builtin_fmaf(u4.y, u4.y, v4.y * v4.y), u4.y * v4.y);
                row[4 * g.lane + 2] = make_float4(u4.z, v4.z, __builtin_fmaf(u4.z, u4.z, v4.z * v4.z), u4.z * v4.z);
                row[4 * g.lane + 3] = make_float4(u4.w, v4.w, __builtin_fmaf(u4.w, u4.w, v4.w * v4.w), u4.w * v4.w);
            }
            // the next row's values travel while this one is blurred
            u4 = strip_load4(a1, g, H, W3, i + 1);
            v4 = strip_load4(a2, g, H, W3, i + 1);
            wave_lds_handoff();
            float4 h[WO];
#pragma unroll
            for (int o = 0; o < WO; o++) h[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int oc = i - HALO;                           // the output row this input row is the centre of
            const bool l1_row = oc >= 0 && oc < g.n_out;
#pragma unroll
            for (int m = 0; m < 10 + WO; m++) {
                const float4 tap = row[g.ob + 3 * m];
#pragma unroll
                for (int o = 0; o < WO; o++) {
                    const int k = m - o;
                    if (k >= 0 && k < 11) fma4(h[o], GW[k], tap);
                    if (k == HALO && l1_row && g.o_on[o]) l1 += fabsf(tap.x - tap.y);
                }
            }
            wave_lds_handoff();                                // the next row's stores stay behind these reads
            // acc[j]: the output row j rows above this input row, taps 0..j in.  An FMA writes where it likes, so the shift
            // acc[j] = acc[j - 1] + G[j] h costs nothing, every index is static and the loop body exists once.
#pragma unroll
            for (int j = 10; j >= 0; j--) {
                const float w = GW[j];
#pragma unroll
                for (int o = 0; o < WO; o++) {
                    if (j == 0) acc[0][o] = make_float4(w * h[o].x, w * h[o].y, w * h[o].z, w * h[o].w);      // a new output row starts
                    else acc[j][o] = make_float4(__builtin_fmaf(w, h[o].x, acc[j - 1][o].x), __builtin_fmaf(w, h[o].y, acc[j - 1][o].y),
                                                 __builtin_fmaf(w, h[o].z, acc[j - 1][o].z), __builtin_fmaf(w, h[o].w, acc[j - 1][o].w));
                }
            }
            const int orow = i - 2 * HALO;                     // the output row that just got its last tap
            if (orow >= 0) {
#pragma unroll
                for (int o = 0; o < WO; o++) {
                    if (!g.o_on[o]) continue;
                    const float4 A4 = acc[10][o];
                    const float mu1 = A4.x, mu2 = A4.y;
                    const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
                    const float sg_sum = (A4.z - mu1s) - mu2s, sg12 = A4.w - mu12;      // sigma1^2 + sigma2^2, sigma12
                    const float A = mu1s + mu2s + C1, B = sg_sum + C2, Cc = 2.f * mu12 + C1, D = 2.f * sg12 + C2;
                    // two reciprocals (hardware estimate + one Newton step: within an ulp of the quotient) instead of the four
                    // IEEE division sequences of the formulas as written
                    const float rA = rcp_newton(A), rB = rcp_newton(B);
                    const float rAB = rA * rB;
                    ss += Cc * D * rAB;
                    const size_t p = base + (size_t)(blockIdx.y * R + orow) * W3 + (g.c0 + g.ob + 3 * o);
                    dm_dmu1[p] = 2.f * rAB * (mu2 * (D - Cc) + mu1 * Cc * D * (rB - rA));
                    dm_dsig1[p] = -Cc * D * rAB * rB;
                    dm_dsig12[p] = 2.f * Cc * rAB;
                }
            }
            }
        }
    }
    if (sums == nullptr) return;
    l1 = wave_sum_all(l1);
    ss = wave_sum_all(ss);
    if (g.lane == 0) {
        atomicAdd(&sums[0], l1);
        atomicAdd(&sums[1], ss);
    }
}

// backward: v_img1 = w_l1*sign(a-b) + w_ssim*(conv(dmu1) + 2a*conv(dsig1) + b*conv(dsig12)), the three maps blurred the same way
__global__ __launch_bounds__(64) void loss_bwd_wave_kernel(int H, int W3, int R, const float* __restrict__ img1,
                                                           const float* __restrict__ img2, const float* __restrict__ dm_dmu1,
                                                           const float* __restrict__ dm_dsig1, const float* __restrict__ dm_dsig12,
                                                           float w_l1, float w_ssim, float* __restrict__ v_img1) {
    __shared__ float4 rowbuf[2][WROW];
    const size_t base = (size_t)blockIdx.z * H * W3;
    const float* m1 = dm_dmu1 + base;
    const float* m2 = dm_dsig1 + base;
    const float* m3 = dm_dsig12 + base;
    const WaveStrip g = wave_strip(H, W3, R);
    float4 acc[11][WO];           // (.w unused)
#pragma unroll
    for (int j = 0; j < 11; j++)
#pragma unroll
        for (int o = 0; o < WO; o++) acc[j][o] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 x1 = strip_load4(m1, g, H, W3, 0), x2 = strip_load4(m2, g, H, W3, 0), x3 = strip_load4(m3, g, H, W3, 0);
    for (int i = 0; i < g.n_in; i++) {
        {
            {
            float4* row = rowbuf[i & 1];
            if (g.s_on) {
                row[4 * g.lane + 0] = make_float4(x1.x, x2.x, x3.x, 0.f);
                row[4 * g.lane + 1] = make_float4(x1.y, x2.y, x3.y, 0.f);
                row[4 * g.lane + 2] = make_float4(x1.z, x2.z, x3.z, 0.f);
                row[4 * g.lane + 3] = make_float4(x1.w, x2.w, x3.w, 0.f);
            }
            x1 = strip_load4(m1, g, H, W3, i + 1);
            x2 = strip_load4(m2, g, H, W3, i + 1);
            x3 = strip_load4(m3, g, H, W3, i + 1);
            // the two image values of the output row this iteration finishes
            const int orow = i - 2 * HALO;
            float a[WO], b[WO];
            size_t p[WO];
#pragma unroll
            for (int o = 0; o < WO; o++) {
                p[o] = base + (size_t)(blockIdx.y * R + (orow >= 0 ? orow : 0)) * W3 + (g.o_on[o] ? g.c0 + g.ob + 3 * o : 0);
                a[o] = 0.f; b[o] = 0.f;
                if (orow >= 0) { a[o] = img1[p[o]]; b[o] = img2[p[o]]; }
            }
            wave_lds_handoff();
            float4 h[WO];         // (.w unused)
#pragma unroll
            for (int o = 0; o < WO; o++) h[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < 10 + WO; m++) {
                const float4 tap = row[g.ob + 3 * m];
#pragma unroll
                for (int o = 0; o < WO; o++) {
                    const int k = m - o;
                    if (k >= 0 && k < 11) fma4(h[o], GW[k], tap);
                }
            }
            wave_lds_handoff();
#pragma unroll
            for (int j = 10; j >= 0; j--) {
                const float w = GW[j];
#pragma unroll
                for (int o = 0; o < WO; o++) {
                    if (j == 0) acc[0][o] = make_float4(w * h[o].x, w * h[o].y, w * h[o].z, 0.f);
                    else acc[j][o] = make_float4(__builtin_fmaf(w, h[o].x, acc[j - 1][o].x), __builtin_fmaf(w, h[o].y, acc[j - 1][o].y),
                                                 __builtin_fmaf(w, h[o].z, acc[j - 1][o].z), 0.f);
                }
            }
            if (orow >= 0) {
#pragma unroll
                for (int o = 0; o < WO; o++) {
                    if (!g.o_on[o]) continue;
                    const float A0 = acc[10][o].x, A1 = acc[10][o].y, A2 = acc[10][o].z;
                    const float dd = a[o] - b[o];
                    const float sgn = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
                    v_img1[p[o]] = w_l1 * sgn + w_ssim * (A0 + 2.f * a[o] * A1 + b[o] * A2);
                }
            }
            }
        }
    }
}

// rows per strip: about 1 800 waves (256 CUs x 4 SIMDs, not quite two each), at most 64 rows, at least 16 (10 extra rows are
// blurred horizontally per strip, at about half the cost of a full row)
static int strip_rows(int C, int H, int W3) {
    const long long waves_per_row = (long long)C * mi_div_up(W3, WCOLS);
    long long R = mi_div_up((long long)H * waves_per_row, 1800ll);
    return (int)(R < 16 ? 16 : R > 64 ? 64 : R);
}

static bool loss_stream() {
    static const bool v = [] { const char* e = MI_EXPERIMENT_ENV("MI3DGS_LOSS_STREAM"); return e && atoi(e) != 0; }();
    return v;
}
#endif  // MI3DGS_EXPERIMENTS

}  // namespace

// sums[2] must be zeroed by the caller (hipMemsetAsync on the same stream) before the call; nullable: a training step that does
// not report its loss value needs neither the clear nor the reduction (the gradient pass only uses the three maps).
extern "C" int mi3dgs_loss_fwd(int C, int height, int width, const float* render, const float* target,
                               float* dm_dmu1, float* dm_dsigma1, float* dm_dsigma12, float* sums, void* stream) {
    MI_REQUIRE(C > 0 && height > 0 && width > 0, "loss_fwd: bad sizes");
#ifdef MI3DGS_EXPERIMENTS
    if (loss_stream()) {
        MI_REQUIRE((long long)height * width * 3 < (1ll << 31), "loss_fwd: image too large for 32-bit offsets");
        const int W3 = width * 3, R = strip_rows(C, height, W3);
        dim3 grid(mi_div_up(W3, WCOLS), mi_div_up(height, R), C);
        MI_LAUNCH("loss_fwd", loss_fwd_wave_kernel, grid, dim3(64), 0, (hipStream_t)stream, height, W3, R, render, target, dm_dmu1,
                  dm_dsigma1, dm_dsigma12, sums);
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    dim3 grid(mi_div_up(width, LT), mi_div_up(height, LT), C);
    MI_LAUNCH("loss_fwd", loss_fwd_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, height, width, render, target, 1.f, dm_dmu1,
              dm_dsigma1, dm_dsigma12, sums);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_loss_fwd_u8(int C, int height, int width, const float* render, const uint8_t* target_u8, float scale,
                                  float* dm_dmu1, float* dm_dsigma1, float* dm_dsigma12, float* sums, void* stream) {
    MI_REQUIRE(C > 0 && height > 0 && width > 0, "loss_fwd_u8: bad sizes");
    dim3 grid(mi_div_up(width, LT), mi_div_up(height, LT), C);
    MI_LAUNCH("loss_fwd", loss_fwd_kernel<uint8_t>, grid, dim3(NT), 0, (hipStream_t)stream, height, width, render, target_u8, scale,
              dm_dmu1, dm_dsigma1, dm_dsigma12, sums);
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
#ifdef MI3DGS_EXPERIMENTS
    if (loss_stream()) {
        MI_REQUIRE((long long)height * width * 3 < (1ll << 31), "loss_bwd: image too large for 32-bit offsets");
        const int W3 = width * 3, R = strip_rows(C, height, W3);
        dim3 grid(mi_div_up(W3, WCOLS), mi_div_up(height, R), C);
        MI_LAUNCH("loss_bwd", loss_bwd_wave_kernel, grid, dim3(64), 0, (hipStream_t)stream, height, W3, R, render, target, dm_dmu1,
                  dm_dsigma1, dm_dsigma12, w_l1, w_ssim, v_render);
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    dim3 grid(mi_div_up(width, LT), mi_div_up(height, LT), C);
    MI_LAUNCH("loss_bwd", loss_bwd_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, height, width, render, target, 1.f, dm_dmu1,
              dm_dsigma1, dm_dsigma12, w_l1, w_ssim, v_render);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_loss_bwd_u8(int C, int height, int width, const float* render, const uint8_t* target_u8, float scale,
                                  const float* dm_dmu1, const float* dm_dsigma1, const float* dm_dsigma12, float ssim_lambda,
                                  float loss_scale, float* v_render, void* stream) {
    MI_REQUIRE(C > 0 && height > 0 && width > 0, "loss_bwd_u8: bad sizes");
    float M = (float)C * (float)height * (float)width * 3.f;
    float w_l1 = loss_scale * (1.f - ssim_lambda) / M;
    float w_ssim = -loss_scale * ssim_lambda / M;
    dim3 grid(mi_div_up(width, LT), mi_div_up(height, LT), C);
    MI_LAUNCH("loss_bwd", loss_bwd_kernel<uint8_t>, grid, dim3(NT), 0, (hipStream_t)stream, height, width, render, target_u8, scale,
              dm_dmu1, dm_dsigma1, dm_dsigma12, w_l1, w_ssim, v_render);
    MI_LAUNCH_CHECK();
    return 0;
}
