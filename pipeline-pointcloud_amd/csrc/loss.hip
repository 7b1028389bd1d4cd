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

__device__ __forceinline__ void stage_load(const float* __restrict__ img, int H, int W, int y0, const StageCol& c,
                                           float (&v)[ST_ITERS]) {
#pragma unroll
    for (int it = 0; it < ST_ITERS; it++) {
        const int y = y0 + it * ST_RGRP + c.rgrp - HALO;
        const bool ok = c.xok && y >= 0 && y < H;
        const int off = ok ? y * (W * 3) + c.xj : 0;      // 32-bit offset from a uniform base: saddr addressing, one VGPR
        const float val = img[off];
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
__global__ __launch_bounds__(NT, 4) void loss_fwd_kernel(int H, int W, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, float* __restrict__ dm_dmu1,
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
        stage_load(img2 + base, H, W, y0, sc, rb);
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
    l1 = block_sum(l1, red);
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], l1);
        atomicAdd(&sums[1], ss);
    }
}

// backward: v_img1 = w_l1*sign(a-b) + w_ssim*(conv(dmu1) + 2a*conv(dsig1) + b*conv(dsig12))
__global__ __launch_bounds__(NT) void loss_bwd_kernel(int H, int W, const float* __restrict__ img1,
                                                       const float* __restrict__ img2,
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
            pv_[ch][o] = (img2 + base)[off];
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
// Streaming kernels (round 3).  An image row is a 1-D array of W * 3 floats (interleaved RGB); the 11-tap blur along x touches
// floats c - 15, c - 12, ..., c + 15 whatever channel c belongs to, and the blur along y stays inside one float-column.  A block
// owns SCOLS float-columns and walks down a strip of R rows:
//   per input row: every thread stages (u, v, u^2 + v^2, u v) of one float as one float4 in LDS (double-buffered row, ONE
//     barrier per row), reads the eleven taps of its own column as float4s, and has the horizontally blurred moments h;
//   vertical blur without LDS: h is added, weighted, to the eleven output rows it belongs to (44 accumulators in registers,
//     rotating -- the loop is unrolled by 11 so that their indices are static); the output row that just received its last
//     tap is finished: SSIM and its three derivative maps, stored coalesced.
// Experiments build only (MI3DGS_LOSS_STREAM=1): measured against the tile kernels on one box, 1080p, it is no faster --
// forward 69 vs 73 us, backward 65 vs 51 us (profiles/r03_loss_stream_ab.txt).  Both designs issue ~255-275 instructions per
// pixel-channel against 44 packed FMAs of arithmetic; with ONE output per thread and row, every address, wait and barrier of a
// row is paid per output, and a strip is a long serial chain (391 blocks for 256 CUs at 1080p).  What would pay is four
// outputs per thread (14 tap reads per 4 outputs, 176 accumulators) in wave-private rows; not built.
constexpr int SCOLS = 256;                  // float-columns (threads) per block
constexpr int SHALO = 15;                   // 5 pixels x 3 interleaved channels
constexpr int SROW = SCOLS + 2 * SHALO;

struct StripGeom {
    int c, cs, chalo, y_in0, n_out, n_in;
    bool ok_s, ok_h, ok_o;
};

__device__ __forceinline__ StripGeom strip_geom(int H, int W3, int R) {
    StripGeom g;
    const int c0 = blockIdx.x * SCOLS, y0 = blockIdx.y * R, t = threadIdx.x;
    g.c = c0 + t;
    g.cs = c0 - SHALO + t;                        // the float this thread stages
    g.chalo = c0 - SHALO + SCOLS + t;             // threads 0..29 stage the right halo too
    g.ok_s = g.cs >= 0 && g.cs < W3;
    g.ok_h = t < 2 * SHALO && g.chalo < W3;
    g.ok_o = g.c < W3;
    g.y_in0 = y0 - HALO;
    g.n_out = min(R, H - y0);
    g.n_in = g.n_out + 2 * HALO;
    return g;
}

// value of input row i of the strip at float-column col (zero outside the image: 'same' zero padding)
__device__ __forceinline__ float strip_load(const float* __restrict__ img, const StripGeom& g, int H, int W3, int i, int col, bool col_ok) {
    const int y = g.y_in0 + i;
    const bool ok = col_ok && y >= 0 && y < H && i < g.n_in;
    const float v = img[ok ? y * W3 + col : 0];
    return ok ? v : 0.f;
}

__device__ __forceinline__ float rcp_newton(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.f), r);
}

__device__ __forceinline__ float4 blur_row(const float4* __restrict__ row, int t) {
    float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float4 tap = row[t + 3 * k];
        const float w = GW[k];
        h.x = __builtin_fmaf(w, tap.x, h.x); h.y = __builtin_fmaf(w, tap.y, h.y);
        h.z = __builtin_fmaf(w, tap.z, h.z); h.w = __builtin_fmaf(w, tap.w, h.w);
    }
    return h;
}

__global__ __launch_bounds__(SCOLS) void loss_fwd_stream_kernel(int H, int W3, int R, const float* __restrict__ img1,
                                                                 const float* __restrict__ img2, float* __restrict__ dm_dmu1,
                                                                 float* __restrict__ dm_dsig1, float* __restrict__ dm_dsig12,
                                                                 float* __restrict__ sums) {
    __shared__ float4 rowbuf[2][SROW];
    __shared__ float red[SCOLS / 64];
    const size_t base = (size_t)blockIdx.z * H * W3;
    const float* a1 = img1 + base;
    const float* a2 = img2 + base;
    const StripGeom g = strip_geom(H, W3, R);
    const int t = threadIdx.x;
    float4 acc[11];
    float l1 = 0.f, ss = 0.f;
    float u = strip_load(a1, g, H, W3, 0, g.cs, g.ok_s), v = strip_load(a2, g, H, W3, 0, g.cs, g.ok_s);
    float uh = strip_load(a1, g, H, W3, 0, g.chalo, g.ok_h), vh = strip_load(a2, g, H, W3, 0, g.chalo, g.ok_h);
    for (int ib = 0; ib < g.n_in; ib += 11) {
#pragma unroll
        for (int r = 0; r < 11; r++) {
            const int i = ib + r;
            if (i < g.n_in) {
            float4* row = rowbuf[i & 1];
            row[t] = make_float4(u, v, __builtin_fmaf(u, u, v * v), u * v);
            if (t < 2 * SHALO) row[SCOLS + t] = make_float4(uh, vh, __builtin_fmaf(uh, uh, vh * vh), uh * vh);
            // the next row's values travel while this one is blurred
            u = strip_load(a1, g, H, W3, i + 1, g.cs, g.ok_s); v = strip_load(a2, g, H, W3, i + 1, g.cs, g.ok_s);
            uh = strip_load(a1, g, H, W3, i + 1, g.chalo, g.ok_h); vh = strip_load(a2, g, H, W3, i + 1, g.chalo, g.ok_h);
            __syncthreads();
            const float4 h = blur_row(row, t);
            const int oc = i - HALO;                           // the output row this input row is the centre of
            if (oc >= 0 && oc < g.n_out && g.ok_o) {
                const float4 ctr = row[t + SHALO];
                l1 += fabsf(ctr.x - ctr.y);
            }
#pragma unroll
            for (int d = 0; d < 11; d++) {
                float4& A = acc[(r - d + 11) % 11];
                const float w = GW[d];
                if (d == 0) A = make_float4(w * h.x, w * h.y, w * h.z, w * h.w);      // a new output row starts
                else { A.x = __builtin_fmaf(w, h.x, A.x); A.y = __builtin_fmaf(w, h.y, A.y); A.z = __builtin_fmaf(w, h.z, A.z); A.w = __builtin_fmaf(w, h.w, A.w); }
            }
            const int o = i - 2 * HALO;                        // the output row that just got its last tap
            if (o >= 0 && g.ok_o) {
                const float4 A4 = acc[(r + 1) % 11];
                const float mu1 = A4.x, mu2 = A4.y;
                const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
                const float sg_sum = (A4.z - mu1s) - mu2s, sg12 = A4.w - mu12;      // sigma1^2 + sigma2^2, sigma12
                const float A = mu1s + mu2s + C1, B = sg_sum + C2, Cc = 2.f * mu12 + C1, D = 2.f * sg12 + C2;
                // two reciprocals (hardware estimate + one Newton step: within an ulp of the quotient) instead of the four IEEE
                // division sequences of the formulas as written -- 16 of the ~250 instructions of a row were v_div_*
                const float rA = rcp_newton(A), rB = rcp_newton(B);
                const float rAB = rA * rB;
                ss += Cc * D * rAB;
                const size_t p = base + (size_t)(blockIdx.y * R + o) * W3 + g.c;
                dm_dmu1[p] = 2.f * rAB * (mu2 * (D - Cc) + mu1 * Cc * D * (rB - rA));
                dm_dsig1[p] = -Cc * D * rAB * rB;
                dm_dsig12[p] = 2.f * Cc * rAB;
            }
            }
        }
    }
    l1 = wave_sum_all(l1);
    ss = wave_sum_all(ss);
    if (lane_id() == 0) { red[t >> 6] = l1; }
    __syncthreads();
    if (t == 0) atomicAdd(&sums[0], red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    if (lane_id() == 0) { red[t >> 6] = ss; }
    __syncthreads();
    if (t == 0) atomicAdd(&sums[1], red[0] + red[1] + red[2] + red[3]);
}

// backward: v_img1 = w_l1*sign(a-b) + w_ssim*(conv(dmu1) + 2a*conv(dsig1) + b*conv(dsig12)), the three maps blurred the same way
__global__ __launch_bounds__(SCOLS) void loss_bwd_stream_kernel(int H, int W3, int R, const float* __restrict__ img1,
                                                                 const float* __restrict__ img2, const float* __restrict__ dm_dmu1,
                                                                 const float* __restrict__ dm_dsig1, const float* __restrict__ dm_dsig12,
                                                                 float w_l1, float w_ssim, float* __restrict__ v_img1) {
    __shared__ float4 rowbuf[2][SROW];
    const size_t base = (size_t)blockIdx.z * H * W3;
    const float* m1 = dm_dmu1 + base;
    const float* m2 = dm_dsig1 + base;
    const float* m3 = dm_dsig12 + base;
    const StripGeom g = strip_geom(H, W3, R);
    const int t = threadIdx.x;
    float4 acc[11];
    float x1 = strip_load(m1, g, H, W3, 0, g.cs, g.ok_s), x2 = strip_load(m2, g, H, W3, 0, g.cs, g.ok_s), x3 = strip_load(m3, g, H, W3, 0, g.cs, g.ok_s);
    float h1 = strip_load(m1, g, H, W3, 0, g.chalo, g.ok_h), h2 = strip_load(m2, g, H, W3, 0, g.chalo, g.ok_h), h3 = strip_load(m3, g, H, W3, 0, g.chalo, g.ok_h);
    for (int ib = 0; ib < g.n_in; ib += 11) {
#pragma unroll
        for (int r = 0; r < 11; r++) {
            const int i = ib + r;
            if (i < g.n_in) {
            float4* row = rowbuf[i & 1];
            row[t] = make_float4(x1, x2, x3, 0.f);
            if (t < 2 * SHALO) row[SCOLS + t] = make_float4(h1, h2, h3, 0.f);
            x1 = strip_load(m1, g, H, W3, i + 1, g.cs, g.ok_s); x2 = strip_load(m2, g, H, W3, i + 1, g.cs, g.ok_s); x3 = strip_load(m3, g, H, W3, i + 1, g.cs, g.ok_s);
            h1 = strip_load(m1, g, H, W3, i + 1, g.chalo, g.ok_h); h2 = strip_load(m2, g, H, W3, i + 1, g.chalo, g.ok_h); h3 = strip_load(m3, g, H, W3, i + 1, g.chalo, g.ok_h);
            // the two image values of the output row this iteration finishes
            const int o = i - 2 * HALO;
            float a = 0.f, b = 0.f;
            const size_t p = base + (size_t)(blockIdx.y * R + (o >= 0 ? o : 0)) * W3 + (g.ok_o ? g.c : 0);
            if (o >= 0) { a = img1[p]; b = img2[p]; }
            __syncthreads();
            const float4 h = blur_row(row, t);
#pragma unroll
            for (int d = 0; d < 11; d++) {
                float4& A = acc[(r - d + 11) % 11];
                const float w = GW[d];
                if (d == 0) A = make_float4(w * h.x, w * h.y, w * h.z, 0.f);
                else { A.x = __builtin_fmaf(w, h.x, A.x); A.y = __builtin_fmaf(w, h.y, A.y); A.z = __builtin_fmaf(w, h.z, A.z); }
            }
            if (o >= 0 && g.ok_o) {
                const float4 A4 = acc[(r + 1) % 11];
                const float dd = a - b;
                const float sgn = dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
                v_img1[p] = w_l1 * sgn + w_ssim * (A4.x + 2.f * a * A4.y + b * A4.z);
            }
            }
        }
    }
}

// rows per strip: enough blocks for every SIMD (256 CUs x 4) to hold about one and a half waves, at most 64 rows (10 extra rows
// are blurred horizontally per strip), at least 16
static int strip_rows(int C, int H, int W3) {
    const long long waves_per_row = (long long)C * mi_div_up(W3, SCOLS) * (SCOLS / 64);
    long long R = (long long)H * waves_per_row / 1536;
    R = (R + 7) / 8 * 8;
    return (int)(R < 16 ? 16 : R > 64 ? 64 : R);
}

static bool loss_stream() {
    static const bool v = [] { const char* e = MI_EXPERIMENT_ENV("MI3DGS_LOSS_STREAM"); return e && atoi(e) != 0; }();
    return v;
}
#endif  // MI3DGS_EXPERIMENTS

}  // namespace

// sums[2] must be zeroed by the caller (hipMemsetAsync on the same stream) before the call.
extern "C" int mi3dgs_loss_fwd(int C, int height, int width, const float* render, const float* target,
                               float* dm_dmu1, float* dm_dsigma1, float* dm_dsigma12, float* sums, void* stream) {
    MI_REQUIRE(C > 0 && height > 0 && width > 0, "loss_fwd: bad sizes");
#ifdef MI3DGS_EXPERIMENTS
    if (loss_stream()) {
        MI_REQUIRE((long long)height * width * 3 < (1ll << 31), "loss_fwd: image too large for 32-bit offsets");
        const int W3 = width * 3, R = strip_rows(C, height, W3);
        dim3 grid(mi_div_up(W3, SCOLS), mi_div_up(height, R), C);
        MI_LAUNCH("loss_fwd", loss_fwd_stream_kernel, grid, dim3(SCOLS), 0, (hipStream_t)stream, height, W3, R, render, target, dm_dmu1,
                  dm_dsigma1, dm_dsigma12, sums);
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    dim3 grid(mi_div_up(width, LT), mi_div_up(height, LT), C);
    MI_LAUNCH("loss_fwd", loss_fwd_kernel, grid, dim3(NT), 0, (hipStream_t)stream, height, width, render, target, dm_dmu1,
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
#ifdef MI3DGS_EXPERIMENTS
    if (loss_stream()) {
        MI_REQUIRE((long long)height * width * 3 < (1ll << 31), "loss_bwd: image too large for 32-bit offsets");
        const int W3 = width * 3, R = strip_rows(C, height, W3);
        dim3 grid(mi_div_up(W3, SCOLS), mi_div_up(height, R), C);
        MI_LAUNCH("loss_bwd", loss_bwd_stream_kernel, grid, dim3(SCOLS), 0, (hipStream_t)stream, height, W3, R, render, target, dm_dmu1,
                  dm_dsigma1, dm_dsigma12, w_l1, w_ssim, v_render);
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    dim3 grid(mi_div_up(width, LT), mi_div_up(height, LT), C);
    MI_LAUNCH("loss_bwd", loss_bwd_kernel, grid, dim3(NT), 0, (hipStream_t)stream, height, width, render, target, dm_dmu1,
              dm_dsigma1, dm_dsigma12, w_l1, w_ssim, v_render);
    MI_LAUNCH_CHECK();
    return 0;
}
