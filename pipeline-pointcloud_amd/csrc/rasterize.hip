// C-ABI entry points of the tile rasteriser (mi3dgs_rasterize_fwd / mi3dgs_rasterize_bwd), and -- in the EXPERIMENTS build
// only -- the round-1 VALU kernels they used to launch.  gfx950 only.
//
// Replaces gsplat rasterize_to_pixels_{fwd,bwd}, reached by the
// reference only through main.py:1312 / main.py:1343.
//
// The product kernels are rasterize_mfma.hip (forward) and rasterize_bwd_mm.hip (backward).  What follows under
// MI3DGS_EXPERIMENTS is the first correct path of round 1, kept for same-box A/B measurements
// (libmi3dgs_exp.so, mi3dgs_debug_set_raster_mode(0)):
//
// One 256-thread workgroup per 16x16 tile = 4 wave64s, each wave owning an 8x8 pixel
// quadrant (tighter screen footprint per wave => more wave-uniform skips than a 16x4 strip).
// The tile's depth-sorted splat records are gathered 256 at a time, ONE 64-byte line each,
// into LDS and then read back as broadcasts.
//
// Backward: per-pixel replay back-to-front; per-splat partials are reduced across the wave
// with DPP (no LDS traffic), combined across the 4 waves in an LDS [slot][16] table, and
// flushed with lanes mapped (record, dword) so that each (tile, Gaussian) costs ONE 64-byte
// float-atomic request into the packed gradient record (MI355X_MICROARCH "Global float
// atomics": requests, not bytes, are the unit that is rate-limited).
#include "common.h"

constexpr int TILE = 16;
[[maybe_unused]] constexpr int BLOCK = TILE * TILE;

#ifdef MI3DGS_EXPERIMENTS
#ifdef MI_RASTER_STATS
// debug build only (make STATS=1): [0] (wave,splat) visits, [1] visits with >=1 live lane,
// [2] live lanes, [3] slots flushed with atomics, [4] slots staged
__device__ unsigned long long g_raster_stats[8];
extern "C" int mi3dgs_debug_raster_stats(unsigned long long* host_out, int reset) {
    if (host_out) (void)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_raster_stats), sizeof(g_raster_stats));
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_raster_stats), z, sizeof(z)); }
    return 0;
}
#endif

namespace {

__device__ __forceinline__ void pixel_of_thread(int tid, int& lx, int& ly) {
    int w = tid >> 6, l = tid & 63;
    lx = ((w & 1) << 3) + (l & 7);
    ly = ((w >> 1) << 3) + (l >> 3);
}

// Two-wide float vectors: clang lowers their +, * and fma to v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32,
// one VALU issue for two lanes' worth of work.  Both rasterisers are 100 % VALU-issue bound
// (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs = the kernel duration), so the splats are staged in LDS as
// PAIRS, field by field, and everything that does not depend on the transmittance chain is
// evaluated for two splats per instruction.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr float LOG2E = 1.4426950408889634f;

// LDS record of a pair of splats (k even, k+1):
//   q0 = (x0, x1, y0, y1)            q1 = (A0', A1', B0', B1')     A' = -log2e A / 2, B' = -log2e B
//   q2 = (C0', C1', o0, o1)          q3 = (r0, g0, r1, g1)         C' = -log2e C / 2
//   q4 = (b0, b1)                    so that  log2(vis) = A' dx^2 + C' dy^2 + B' dx dy  directly
struct PairLds {
    float4 q0[BLOCK / 2], q1[BLOCK / 2], q2[BLOCK / 2], q3[BLOCK / 2];
    float2 q4[BLOCK / 2];
};

__device__ __forceinline__ void pair_store(PairLds& L, int slot, float4 a, float4 bb, float c) {
    // a = (x, y, conic A, conic B), bb = (conic C, opacity, r, g), c = b
    float* q0 = reinterpret_cast<float*>(&L.q0[slot >> 1]);
    float* q1 = reinterpret_cast<float*>(&L.q1[slot >> 1]);
    float* q2 = reinterpret_cast<float*>(&L.q2[slot >> 1]);
    float* q3 = reinterpret_cast<float*>(&L.q3[slot >> 1]);
    float* q4 = reinterpret_cast<float*>(&L.q4[slot >> 1]);
    int h = slot & 1;
    q0[h] = a.x; q0[2 + h] = a.y;
    q1[h] = -0.5f * LOG2E * a.z; q1[2 + h] = -LOG2E * a.w;
    q2[h] = -0.5f * LOG2E * bb.x; q2[2 + h] = bb.y;
    q3[2 * h] = bb.z; q3[2 * h + 1] = bb.w;
    q4[h] = c;
}

template <bool HAS_BG>
__global__ __launch_bounds__(BLOCK) void rasterize_fwd_kernel(
    int W, int H, int tw, int th, const float* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
    const float* __restrict__ backgrounds, float* __restrict__ render, float* __restrict__ alphas,
    int32_t* __restrict__ last_ids) {
    __shared__ PairLds L;
    int t = blockIdx.x;
    int cam = t / (tw * th);
    int tile_in = t - cam * (tw * th);
    int ty = tile_in / tw, tx = tile_in - ty * tw;
    int lx, ly;
    pixel_of_thread(threadIdx.x, lx, ly);
    int px_i = tx * TILE + lx, py_i = ty * TILE + ly;
    bool inside = px_i < W && py_i < H;
    const f2 PX = {(float)px_i + 0.5f, (float)px_i + 0.5f}, PY = {(float)py_i + 0.5f, (float)py_i + 0.5f};
    int start = tile_offsets[t];
    int end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;

    float T = 1.f, b = 0.f;
    f2 RG = {0.f, 0.f};
    int cur = 0;
    bool done = !inside;
    for (int bs = start; bs < end; bs += BLOCK) {
        if (__syncthreads_count(done) == BLOCK) break;
        int idx = bs + (int)threadIdx.x;
        if (idx < end) {
            const float4* rec = reinterpret_cast<const float4*>(splats + (size_t)flatten_ids[idx] * SPLAT_STRIDE);
            float4 a = rec[0], bb = rec[1];
            float c = reinterpret_cast<const float*>(rec)[SP_B];
            pair_store(L, (int)threadIdx.x, a, bb, c);
        } else {
            // pad the tail with zero-opacity records so the unrolled body needs no bounds test
            pair_store(L, (int)threadIdx.x, make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), 0.f);
        }
        __syncthreads();
        int bsz = min(BLOCK, end - bs);
        // The per-splat body is branch-free (predicated): divergent `continue`/`break` made this
        // loop issue more SALU (exec-mask bookkeeping) than VALU instructions.  Only two
        // wave-uniform branches remain: "whole wave finished" per group of 4 splats and
        // "no lane of the wave is touched by this splat".
        for (int k0 = 0; k0 < bsz; k0 += 4) {
            if (wave_ballot(!done) == 0ull) break;
            // independent part first, two splats per instruction; the dependent T chain after
            float alpha[4], cr[4], cg[4], cb[4];
            bool hit[4];
#pragma unroll
            for (int pr = 0; pr < 2; pr++) {
                const int j = (k0 >> 1) + pr;
                const float4 q0 = L.q0[j], q1 = L.q1[j], q2 = L.q2[j], q3 = L.q3[j];
                const float2 q4 = L.q4[j];
                const f2 X = {q0.x, q0.y}, Y = {q0.z, q0.w}, A = {q1.x, q1.y}, B = {q1.z, q1.w}, C = {q2.x, q2.y},
                         O = {q2.z, q2.w};
                const f2 DX = X - PX, DY = Y - PY;
                const f2 Tq = B * DY + A * DX;                  // log2(vis) = dx (A' dx + B' dy) + C' dy^2
                f2 S = DX * Tq;
                S = (C * DY) * DY + S;
                const f2 E = {__builtin_amdgcn_exp2f(S.x), __builtin_amdgcn_exp2f(S.y)};
                const f2 AL = O * E;
                alpha[2 * pr] = fminf(MAX_ALPHA, AL.x);
                alpha[2 * pr + 1] = fminf(MAX_ALPHA, AL.y);
                hit[2 * pr] = S.x <= 0.f && alpha[2 * pr] >= ALPHA_THRESHOLD;          // sigma >= 0
                hit[2 * pr + 1] = S.y <= 0.f && alpha[2 * pr + 1] >= ALPHA_THRESHOLD;
                cr[2 * pr] = q3.x; cg[2 * pr] = q3.y; cr[2 * pr + 1] = q3.z; cg[2 * pr + 1] = q3.w;
                cb[2 * pr] = q4.x; cb[2 * pr + 1] = q4.y;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                bool ok = !done && hit[u];
                if (wave_ballot(ok) == 0ull) continue;
                float nT = T * (1.f - alpha[u]);
                bool stop = ok && nT <= T_STOP;
                done = done || stop;
                ok = ok && !stop;
                float wgt = ok ? alpha[u] * T : 0.f;
                const f2 Wg = {wgt, wgt}, Cc = {cr[u], cg[u]};
                RG = Cc * Wg + RG;
                b += cb[u] * wgt;
                cur = ok ? bs + k0 + u : cur;
                T = ok ? nT : T;
            }
        }
    }
    if (inside) {
        size_t pix = ((size_t)cam * H + py_i) * W + px_i;
        float r = RG.x, g = RG.y;
        if (HAS_BG) {
            const float* bg = backgrounds + 3 * cam;
            r += T * bg[0]; g += T * bg[1]; b += T * bg[2];
        }
        render[3 * pix] = r; render[3 * pix + 1] = g; render[3 * pix + 2] = b;
        alphas[pix] = 1.f - T;
        last_ids[pix] = cur;
    }
}

template <bool HAS_BG, bool ABSGRAD>
__global__ __launch_bounds__(BLOCK) void rasterize_bwd_kernel(
    int W, int H, int tw, int th, const float* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
    const float* __restrict__ backgrounds, const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render, const float* __restrict__ v_alphas, float* __restrict__ v_splats) {
    __shared__ float4 sA[BLOCK], sB[BLOCK];
    __shared__ float sC[BLOCK];
    __shared__ int sId[BLOCK];
    __shared__ float acc[BLOCK][GRAD_STRIDE];
    __shared__ int touched[BLOCK];
    __shared__ int wave_max[4];

    int t = blockIdx.x;
    int cam = t / (tw * th);
    int tile_in = t - cam * (tw * th);
    int ty = tile_in / tw, tx = tile_in - ty * tw;
    int lx, ly;
    pixel_of_thread(threadIdx.x, lx, ly);
    int px_i = tx * TILE + lx, py_i = ty * TILE + ly;
    bool inside = px_i < W && py_i < H;
    float px = (float)px_i + 0.5f, py = (float)py_i + 0.5f;
    int start = tile_offsets[t];
    int end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;
    if (end <= start) return;
    int lane = lane_id(), wv = threadIdx.x >> 6;

    float T_final = 1.f, vr0 = 0.f, vr1 = 0.f, vr2 = 0.f, va = 0.f;
    int bin_final = -1;
    if (inside) {
        size_t pix = ((size_t)cam * H + py_i) * W + px_i;
        T_final = 1.f - alphas[pix];
        bin_final = last_ids[pix];
        vr0 = v_render[3 * pix]; vr1 = v_render[3 * pix + 1]; vr2 = v_render[3 * pix + 2];
        va = v_alphas[pix];
        // a pixel that composited nothing has last_id 0 and alpha 0: replaying splat `start`
        // is then harmless only if it is skipped, so mark it explicitly
        if (alphas[pix] == 0.f) bin_final = -1;
    }
    float bgdot = 0.f;
    if (HAS_BG) {
        const float* bg = backgrounds + 3 * cam;
        bgdot = bg[0] * vr0 + bg[1] * vr1 + bg[2] * vr2;
    }
    // wave / block maximum of bin_final
    int wmax = bin_final;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
    if (lane == 0) wave_max[wv] = wmax;
    __syncthreads();
    int bmax = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    if (bmax < start) return;

    float T = T_final;
    float buf0 = 0.f, buf1 = 0.f, buf2 = 0.f;
    for (int be = bmax; be >= start; be -= BLOCK) {
        // slot k <-> sorted index be - k
        int idx = be - (int)threadIdx.x;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GRAD_STRIDE; k++) acc[threadIdx.x][k] = 0.f;
        touched[threadIdx.x] = 0;
        if (idx >= start) {
            int id = flatten_ids[idx];
            const float4* rec = reinterpret_cast<const float4*>(splats + (size_t)id * SPLAT_STRIDE);
            sA[threadIdx.x] = rec[0]; sB[threadIdx.x] = rec[1];
            sC[threadIdx.x] = reinterpret_cast<const float*>(rec)[SP_B];
            sId[threadIdx.x] = id;
        }
        __syncthreads();
        int bsz = min(BLOCK, be - start + 1);
        int k0 = max(0, be - wmax);          // wave-uniform: nothing in this wave is live before k0
        for (int k = k0; k < bsz; k++) {
            int sidx = be - k;
            float4 a = sA[k], bb = sB[k];
            float cb = sC[k];
            float dx = a.x - px, dy = a.y - py;
            float sigma = 0.5f * (a.z * dx * dx + bb.x * dy * dy) + a.w * dx * dy;
            float vis = __expf(-sigma);
            float alpha = fminf(MAX_ALPHA, bb.y * vis);
            bool valid = (sidx <= bin_final) && !(sigma < 0.f || alpha < ALPHA_THRESHOLD);
#ifdef MI_RASTER_STATS
            {
                unsigned long long bm = wave_ballot(valid);
                if (lane == 0) {
                    atomicAdd(&g_raster_stats[0], 1ull);
                    if (bm) { atomicAdd(&g_raster_stats[1], 1ull); atomicAdd(&g_raster_stats[2], (unsigned long long)__popcll(bm)); }
                }
            }
#endif
            if (wave_ballot(valid) == 0ull) continue;
            // Branch-free live part: a dead lane runs it with alpha = 0 (ra = 1, T and buf
            // unchanged bit for bit, every partial 0), so no exec-mask region is needed.
            float a_eff = valid ? alpha : 0.f;
            float ra = __builtin_amdgcn_rcpf(1.f - a_eff);
            T *= ra;
            float fac = a_eff * T;
            float g_r = fac * vr0, g_g = fac * vr1, g_b = fac * vr2;
            float v_alpha = (bb.z * T - buf0 * ra) * vr0 + (bb.w * T - buf1 * ra) * vr1 + (cb * T - buf2 * ra) * vr2;
            v_alpha += T_final * ra * va;
            if (HAS_BG) v_alpha -= T_final * ra * bgdot;
            buf0 += bb.z * fac; buf1 += bb.w * fac; buf2 += cb * fac;
            float ov = bb.y * vis;
            bool grad_on = valid && ov <= MAX_ALPHA;
            float v_sigma = grad_on ? -ov * v_alpha : 0.f;
            float g_o = grad_on ? vis * v_alpha : 0.f;
            float g_ca = 0.5f * v_sigma * dx * dx;
            float g_cb = v_sigma * dx * dy;
            float g_cc = 0.5f * v_sigma * dy * dy;
            float g_x = v_sigma * (a.z * dx + a.w * dy);
            float g_y = v_sigma * (a.w * dx + bb.x * dy);
            float g_ax = 0.f, g_ay = 0.f;
            if (ABSGRAD) { g_ax = fabsf(g_x); g_ay = fabsf(g_y); }
            // reduce-scatter: lane l ends with the total of component (l >> 3) in g_x, lane 63
            // with the total of g_b  ->  ONE 9-lane LDS atomic per (quadrant, splat)
            wave_reduce_scatter8_plus1(g_x, g_y, g_ca, g_cb, g_cc, g_o, g_r, g_g, g_b);
            if (ABSGRAD) { g_ax = wave_sum_to_lane63(g_ax); g_ay = wave_sum_to_lane63(g_ay); }
            {
                bool last = lane == 63;
                if ((lane & 7) == 0 || last) atomicAdd(&acc[k][last ? GR_B : (lane >> 3)], last ? g_b : g_x);
                if (last) {
                    if (ABSGRAD) { atomicAdd(&acc[k][GR_ABSX], g_ax); atomicAdd(&acc[k][GR_ABSY], g_ay); }
                    touched[k] = 1;
                }
            }
        }
        __syncthreads();
        // flush: lane -> (record = lane>>4, dword = lane&15): 4 records = 4 x 64-B requests per instruction
        for (int s = wv * 64; s < wv * 64 + 64; s += 4) {
            int slot = s + (lane >> 4);
            int comp = lane & 15;
#ifdef MI_RASTER_STATS
            if (comp == 0 && slot < bsz) { atomicAdd(&g_raster_stats[4], 1ull); if (touched[slot]) atomicAdd(&g_raster_stats[3], 1ull); }
#endif
            if (slot < bsz && touched[slot] && comp < (ABSGRAD ? GR_DEPTH : GR_ABSX)) {
                float v = acc[slot][comp];
                atomicAdd(&v_splats[(size_t)sId[slot] * GRAD_STRIDE + comp], v);
            }
        }
    }
}

}  // namespace

#endif  // MI3DGS_EXPERIMENTS

int mi_rasterize_fwd_mfma(int n_tiles, int width, int height, int tile_width, int tile_height, const float* splats,
                          const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                          const float* backgrounds, float* render, float* alphas, int32_t* last_ids, void* seg_ws, size_t seg_ws_bytes,
                          hipStream_t st);
int mi_rasterize_bwd_mfma(int n_tiles, int width, int height, int tile_width, int tile_height, const float* splats,
                          const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                          const float* backgrounds, const float* alphas, const int32_t* last_ids, const float* v_render,
                          const float* v_alphas, int absgrad, float* v_splats, int variant, long long n_gauss, const float* render,
                          void* seg_ws, size_t seg_ws_bytes, hipStream_t st);
static int g_raster_mode = 1;
// Mode 1 = the product kernels, the only mode the product library has.  The experiments build (libmi3dgs_exp.so) adds:
// 0 = round-1 VALU kernels; 3 = MFMA forward + backward with the all-f32 cross-lane reduce-scatter instead of the bf16 MFMA
// contraction (correct; the f32 yardstick of tests/test_gpu_configs.py); 4 = backward with THREE-term bf16 pixel sums
// (24 significant bits; the A/B of VERDICT r2 #3); 14 = wave-flush backward (correct, slower); 21 / 22 = the product backward
// forced to its DEEP / WIDE shape; 11..13 = timing experiments with WRONG results (no group flush / constant colours).
// Forward and backward must run in the same mode.
extern "C" int mi3dgs_debug_set_raster_mode(int mode) {
#ifdef MI3DGS_EXPERIMENTS
    MI_REQUIRE(mode == 0 || mode == 1 || mode == 3 || mode == 4 || (mode >= 11 && mode <= 14) || mode == 21 || mode == 22,
               "set_raster_mode: unknown mode");
    g_raster_mode = mode;
    return 0;
#else
    MI_REQUIRE(mode == 1, "set_raster_mode: this is the product library, it has mode 1 only (experiments: libmi3dgs_exp.so)");
    return 0;
#endif
}

extern "C" int mi3dgs_rasterize_fwd(int C, int width, int height, int tile_size, int tile_width, int tile_height,
                                    const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                                    const int32_t* n_isect_dev, const float* backgrounds, float* render,
                                    float* alphas, int32_t* last_ids, void* seg_ws, size_t seg_ws_bytes, void* stream) {
    MI_REQUIRE(tile_size == TILE, "rasterize_fwd: tile_size must be 16");
    MI_REQUIRE(C > 0 && width > 0 && height > 0, "rasterize_fwd: bad sizes");
    MI_REQUIRE(tile_width == mi_div_up(width, TILE) && tile_height == mi_div_up(height, TILE),
               "rasterize_fwd: tile grid does not match image size");
    int n_tiles = C * tile_width * tile_height;
    hipStream_t st = (hipStream_t)stream;
#ifdef MI3DGS_EXPERIMENTS
    if (g_raster_mode == 0) {
        if (backgrounds)
            MI_LAUNCH("rasterize_fwd", rasterize_fwd_kernel<true>, dim3(n_tiles), dim3(BLOCK), 0, st, width, height, tile_width,
                               tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, render,
                               alphas, last_ids);
        else
            MI_LAUNCH("rasterize_fwd", rasterize_fwd_kernel<false>, dim3(n_tiles), dim3(BLOCK), 0, st, width, height, tile_width,
                               tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, render,
                               alphas, last_ids);
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    return mi_rasterize_fwd_mfma(n_tiles, width, height, tile_width, tile_height, splats, isect_offsets, flatten_ids,
                                 n_isect_dev, backgrounds, render, alphas, last_ids, seg_ws, seg_ws_bytes, st);
}

extern "C" int mi3dgs_rasterize_bwd(int C, int width, int height, int tile_size, int tile_width, int tile_height,
                                    const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                                    const int32_t* n_isect_dev, const float* backgrounds, const float* alphas,
                                    const int32_t* last_ids, const float* v_render, const float* v_alphas,
                                    int absgrad, float* v_splats, long long n_gaussians, const float* render, void* seg_ws,
                                    size_t seg_ws_bytes, void* stream) {
    MI_REQUIRE(tile_size == TILE, "rasterize_bwd: tile_size must be 16");
    MI_REQUIRE(tile_width == mi_div_up(width, TILE) && tile_height == mi_div_up(height, TILE),
               "rasterize_bwd: tile grid does not match image size");
    int n_tiles = C * tile_width * tile_height;
    hipStream_t st = (hipStream_t)stream;
#ifdef MI3DGS_EXPERIMENTS
    if (g_raster_mode == 0) {
#define LAUNCH_BWD(BG, AG)                                                                                            \
    MI_LAUNCH("rasterize_bwd", (rasterize_bwd_kernel<BG, AG>), dim3(n_tiles), dim3(BLOCK), 0, st, width, height, tile_width,   \
                       tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, alphas,     \
                       last_ids, v_render, v_alphas, v_splats)
        if (backgrounds) { if (absgrad) LAUNCH_BWD(true, true); else LAUNCH_BWD(true, false); }
        else { if (absgrad) LAUNCH_BWD(false, true); else LAUNCH_BWD(false, false); }
#undef LAUNCH_BWD
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    return mi_rasterize_bwd_mfma(n_tiles, width, height, tile_width, tile_height, splats, isect_offsets, flatten_ids,
                                 n_isect_dev, backgrounds, alphas, last_ids, v_render, v_alphas, absgrad, v_splats, g_raster_mode,
                                 n_gaussians, render, seg_ws, seg_ws_bytes, st);
}
