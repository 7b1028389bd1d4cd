// Tile rasteriser: the forward kernel and the reduce-scatter backward kernel (the A/B partner of rasterize_bwd_mm.hip, which is
// the product backward), both with the pixel x splat quadratic forms on the MATRIX pipe (rasterize_mfma.h).
// gfx950 only.  Replaces gsplat rasterize_to_pixels_{fwd,bwd}, reached by the
// reference only through main.py:1312 / main.py:1343.
//
// What is a matrix product here.  For a pixel p of a tile and a splat s of its list
//     log2 alpha(s, p) = log2 o_s - log2e * sigma(s, p)
//                      = c0 u^2 + c1 uv + c2 v^2 + c3 u + c4 v + c5,     (u, v) = p - tile centre,
// a [32 splats x 6] . [6 x 64 pixels] product.  Round 2 first ran it on v_mfma_f32_32x32x2_f32; an f32-input MFMA turned
// out to occupy the vector ALUs for its whole duration (DESIGN.md finding 25), so it now runs as two
// v_mfma_f32_32x32x16_bf16 per 32 pixels with every coefficient in three bf16 terms and a basis that is exact in bf16
// (rasterize_mfma.h: the same sums in f32, 24 significant bits per coefficient).
// The six coefficients are computed once per (tile, splat) when the record is staged into LDS
// (quad_coefs, tile-centre coordinates keep |u|, |v| <= 7.5 so the f32 chain cancels to ~4e-4 in log2
// units at worst); the per-pixel basis is ten registers per lane for the whole kernel.
//
// Layout.  A = coefficients (row = splat), B = basis (column = pixel).  The C/D map of 32x32 puts one pixel
// COLUMN on a lane and 16 of the 32 splat ROWS in its registers, the other 16 rows of the same pixel on
// lane + 32.  A wave owns an 8x8 quadrant = two 32-pixel column blocks X and Y; 16 v_permlane32_swap
// (half a VALU per visit) exchange X's upper-half rows against Y's lower-half rows, after which EVERY lane
// owns one pixel and holds all 32 splats of the sub-batch in 32 registers, statically indexed.  The
// front-to-back chain then runs out of registers.
//
// Forward and backward evaluate log2 alpha through the SAME instruction sequence (quad_coefs -> split3 -> the
// same four MFMAs -> v_min -> v_exp), so the backward's membership test alpha >= 1/255 is the forward's
// bit for bit (round 1 used exp2 of a pre-scaled conic forward and __expf backward).
//
// Backward (this file's kernel): what is reduced over the pixels of a quadrant per splat is q = -dL/dsigma and its five
// moments q u, q v, q u^2, q uv, q v^2 plus the three colour sums (9 values, one cross-lane reduce-scatter per live visit);
// the conic / mean / opacity gradients follow per (tile, splat) from the moments at flush time, so the per-pair
// products dx^2, dx dy, A dx + B dy ... are gone from the inner loop.

#include "rasterize_mfma.h"

namespace mfma_raster {

template <bool HAS_BG>
__global__ __launch_bounds__(BLOCK) void rasterize_fwd_kernel(
    int W, int H, int tw, int th, const float* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
    const float* __restrict__ backgrounds, float* __restrict__ render, float* __restrict__ alphas,
    int32_t* __restrict__ last_ids, int bands, SegWs seg) {
    __shared__ Staged L;
    __shared__ uint32_t s_slot;
    const int t = tile_of_block((int)blockIdx.x, n_tiles_total, bands, tw);
    if (t < 0) return;
    const int cam = t / (tw * th);
    const int tile_in = t - cam * (tw * th);
    const int ty = tile_in / tw, tx = tile_in - ty * tw;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    int lx, ly;
    pixel_of_lane(wv, lane, lx, ly);
    const int px_i = tx * TILE + lx, py_i = ty * TILE + ly;
    const bool inside = px_i < W && py_i < H;
    const float xc = (float)(tx * TILE) + 8.f, yc = (float)(ty * TILE) + 8.f;     // pixel centre p + 0.5 = centre + (u, v)
    const Basis basis = make_basis(wv, lane);
    const int start = tile_offsets[t];
    const int end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;

    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
    int cur = 0;
    unsigned long long live = wave_ballot(inside);            // lanes still compositing
    uint32_t nb = 0;                                          // segment boundaries this block walked past (backward segments)
    bool seg_on = seg.ckpt != nullptr;
    for (int bs = start; bs < end; bs += BLOCK) {
        if (!__syncthreads_or(live != 0ull)) break;
        if (seg_on && bs > start && ((bs - start) & (int)(seg.seg - 1)) == 0) {
            // somebody composites on beyond this boundary: leave the state for the backward's segment in front of it
            if (threadIdx.x == 0) {
                const uint32_t slot = atomicAdd(&seg.ctl[0], 1u);
                s_slot = slot;
                if (slot < seg.cap) seg.work[slot] = make_uint4((uint32_t)t, (uint32_t)bs - seg.seg, slot, seg.seg);
            }
            __syncthreads();
            const uint32_t slot = s_slot;
            if (slot < seg.cap) {
                seg.ckpt[(size_t)slot * BLOCK + threadIdx.x] = make_float4(T, cr, cg, cb);
                nb++;
            } else {
                seg_on = false;           // (can not happen: cap >= I / 256; the tile's own block then keeps the rest)
            }
        }
        {
            // (fetching the list one batch ahead was measured: forward 136.6 -> 149.5 us on the same box; the other
            // blocks of the CU already hide this gather)
            const int i1 = bs + (int)threadIdx.x;
            stage_splat(L, (int)threadIdx.x, load_rec(splats, i1 < end ? flatten_ids[i1] : -1), xc, yc);
        }
        __syncthreads();
        const int bsz = min(BLOCK, end - bs);
        for (int sb = 0; sb * SUB < bsz; sb++) {
            if (live == 0ull) break;
            float s[SUB];
            eval_sub_batch(L, sb, lane, basis, s);
            const lds_f4_ptr uni = opaque_lds_base(&L.uni[sb * SUB]);
            Rgb col_next = lds_rgb(uni, 0);
#pragma unroll
            for (int i = 0; i < SUB; i++) {
                if ((i & 7) == 0 && i > 0 && live == 0ull) break;
                const Rgb col = col_next;
                col_next = lds_rgb(uni, i + 1);                 // one visit ahead: its latency hides behind this visit
                const unsigned long long hit = mask_ge(s[i], LOG2_ALPHA_THRESHOLD) & live;
                if (hit == 0ull) continue;
                const float alpha = alpha_of(s[i]);
                const float wgt = alpha * T;
                const float nT = T - wgt;                       // T (1 - alpha)
                const unsigned long long comp = mask_gt(nT, T_STOP) & hit;      // lanes that composite this splat
                live &= ~(hit & ~comp);                         // the others that hit it are finished
                if (lane_of(comp)) {
                    cr = __builtin_fmaf(col.x, wgt, cr);
                    cg = __builtin_fmaf(col.y, wgt, cg);
                    cb = __builtin_fmaf(col.z, wgt, cb);
                    cur = bs + sb * SUB + i;
                    T = nT;
                }
            }
        }
    }
    if (inside) {
        const size_t pix = ((size_t)cam * H + py_i) * W + px_i;
        if (HAS_BG) {
            const float* bg = backgrounds + 3 * cam;
            cr += T * bg[0]; cg += T * bg[1]; cb += T * bg[2];
        }
        render[3 * pix] = cr; render[3 * pix + 1] = cg; render[3 * pix + 2] = cb;
        alphas[pix] = 1.f - T;
        last_ids[pix] = cur;
    }
    if (seg.ckpt != nullptr && threadIdx.x == 0) seg.tile_skip[t] = nb * seg.seg;
}

#ifdef MI3DGS_EXPERIMENTS
// ---------------------------------------------------------------------------------------- backward (cross-lane reduce-scatter)
struct StagedBwd {
    Staged f;
    float4 geo[BLOCK];        // mx, my (relative to the tile centre), A, B
    float2 geo2[BLOCK];       // C, 1 / o
    int id[BLOCK];
    float acc[BLOCK][AC_STRIDE];
    int wave_max[4];
};


// One sub-batch of the backward walk, rows i = 0..31 <-> sorted indices be - 32 sb - i (back to front).
// FAST: every pixel of the wave that composited anything is already in range (index <= its last contributor).
template <bool ABSGRAD, bool FAST>
__device__ __forceinline__ void bwd_sub_batch(StagedBwd& L, const float (&s)[SUB], int sb, int be, int lane, int bin_final,
                                              unsigned long long has, const PixelBasis& px, const float (&vrgb)[3], float tail,
                                              float& T, float& bufdot) {
    const lds_f4_ptr uni = opaque_lds_base(&L.f.uni[sb * SUB]);
    Rgb col_next = lds_rgb(uni, 0);
    const bool adds = (lane & 7) == 0 || lane == 63;
    float* acc_lane = &L.acc[sb * SUB][lane == 63 ? AC_B : (lane >> 3)];
#pragma unroll
    for (int i = 0; i < SUB; i++) {
        const int k = sb * SUB + i;
        const Rgb col = col_next;
        col_next = lds_rgb(uni, i + 1);
        // the forward's own membership test (same MFMA result, same compare), for the splats this pixel reached
        unsigned long long valid = mask_ge(s[i], LOG2_ALPHA_THRESHOLD) & has;
        if (!FAST) valid &= mask_ge_i(bin_final, be - k);
        if (valid == 0ull) continue;
        const float alpha = alpha_of(s[i]);
        // branch-free live part: a lane that does not take part runs it with alpha = 0 (ra = 1, T and bufdot
        // unchanged bit for bit, every partial 0)
        const float a_eff = lane_of(valid) ? alpha : 0.f;
        const float ra = __builtin_amdgcn_rcpf(1.f - a_eff);
        T *= ra;
        const float fac = a_eff * T;
        float g_r = fac * vrgb[0], g_g = fac * vrgb[1], g_b = fac * vrgb[2];
        const float cv = col.x * vrgb[0] + col.y * vrgb[1] + col.z * vrgb[2];            // c . v_rgb
        const float v_alpha = T * cv - ra * (bufdot - tail);
        bufdot = __builtin_fmaf(cv, fac, bufdot);
        // q = o vis dL/dalpha = -dL/dsigma; zero where the 0.999 clamp is active (alpha == o vis otherwise)
        const unsigned long long gon = valid & mask_le(s[i], LOG2_MAX_ALPHA);
        float q = lane_of(gon) ? alpha * v_alpha : 0.f;
        float qu = q * px.u, qv = q * px.v, quu = q * px.uu, quv = q * px.uv, qvv = q * px.vv;
        float g_ax = 0.f, g_ay = 0.f;
        if (ABSGRAD) {
            const float4 ge = L.geo[k];
            const float cC = L.geo2[k].x;
            const float dx = ge.x - px.u, dy = ge.y - px.v;
            g_ax = fabsf(q * (ge.z * dx + ge.w * dy));
            g_ay = fabsf(q * (ge.w * dx + cC * dy));
        }
        // reduce-scatter: lane l ends with the total of value number (l >> 3) in qu, lane 63 with g_b's
        wave_reduce_scatter8_plus1(qu, qv, quu, quv, qvv, q, g_r, g_g, g_b);
        if (ABSGRAD) { g_ax = wave_sum_to_lane63(g_ax); g_ay = wave_sum_to_lane63(g_ay); }
        // one 9-lane LDS atomic per (quadrant, splat); `acc_lane` already points at this lane's component of the
        // sub-batch's first row, so the row is an immediate offset
        if (adds) atomicAdd(acc_lane + i * AC_STRIDE, lane == 63 ? g_b : qu);
        if (ABSGRAD && lane == 63) { atomicAdd(&L.acc[k][AC_ABSX], g_ax); atomicAdd(&L.acc[k][AC_ABSY], g_ay); }
    }
}

template <bool HAS_BG, bool ABSGRAD, bool PREFETCH>
__global__ __launch_bounds__(BLOCK) void rasterize_bwd_kernel(
    int W, int H, int tw, int th, const float* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
    const float* __restrict__ backgrounds, const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render, const float* __restrict__ v_alphas, float* __restrict__ v_splats, int bands) {
    __shared__ StagedBwd L;
    const int t = tile_of_block((int)blockIdx.x, n_tiles_total, bands, tw);
    if (t < 0) return;
    const int cam = t / (tw * th);
    const int tile_in = t - cam * (tw * th);
    const int ty = tile_in / tw, tx = tile_in - ty * tw;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    int lx, ly;
    pixel_of_lane(wv, lane, lx, ly);
    const int px_i = tx * TILE + lx, py_i = ty * TILE + ly;
    const bool inside = px_i < W && py_i < H;
    const float xc = (float)(tx * TILE) + 8.f, yc = (float)(ty * TILE) + 8.f;
    PixelBasis px;
    px.u = (float)lx - 7.5f; px.v = (float)ly - 7.5f;
    px.uu = px.u * px.u; px.uv = px.u * px.v; px.vv = px.v * px.v;
    const int start = tile_offsets[t];
    const int end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;
    if (end <= start) return;

    float T_final = 1.f, vr0 = 0.f, vr1 = 0.f, vr2 = 0.f, va = 0.f;
    int bin_final = -1;
    if (inside) {
        const size_t pix = ((size_t)cam * H + py_i) * W + px_i;
        const float al = alphas[pix];
        T_final = 1.f - al;
        bin_final = last_ids[pix];
        vr0 = v_render[3 * pix]; vr1 = v_render[3 * pix + 1]; vr2 = v_render[3 * pix + 2];
        va = v_alphas[pix];
        // a pixel that composited nothing has last_id 0 and alpha 0: mark it so that slot `start` is skipped
        if (al == 0.f) bin_final = -1;
    }
    const float vrgb[3] = {vr0, vr1, vr2};
    float tail = T_final * va;                      // T_final (v_alpha - bg . v_rgb)
    if (HAS_BG) {
        const float* bg = backgrounds + 3 * cam;
        tail -= T_final * (bg[0] * vr0 + bg[1] * vr1 + bg[2] * vr2);
    }
    int wmax = bin_final;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);            // scalar: the loop bounds and list indices stay on the SALU
    // smallest last-contributor index among the wave's pixels that composited anything: once the walk is at or
    // below it, every such pixel is in range and the per-visit index compare drops out (FAST)
    int wmin = bin_final >= 0 ? bin_final : 0x7fffffff;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmin = min(wmin, __shfl_xor(wmin, o, 64));
    wmin = __builtin_amdgcn_readfirstlane(wmin);
    if (lane == 0) L.wave_max[wv] = wmax;
    __syncthreads();
    const int bmax = max(max(L.wave_max[0], L.wave_max[1]), max(L.wave_max[2], L.wave_max[3]));
    if (bmax < start) return;
    const Basis basis = make_basis(wv, lane);

    float T = T_final;
    float bufdot = 0.f;                         // (colour accumulated behind the current splat) . v_rgb
    const unsigned long long has = wave_ballot(bin_final >= 0);
    // ids two batches ahead, records one batch ahead (slot k of a batch <-> sorted index be - k)
    const int j0 = bmax - (int)threadIdx.x;
    int id_cur = j0 >= start ? flatten_ids[j0] : -1;
    RecRegs rec_next = load_rec(splats, id_cur);
    int id_next = j0 - BLOCK >= start ? flatten_ids[j0 - BLOCK] : -1;
    for (int be = bmax; be >= start; be -= BLOCK) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < AC_STRIDE; k++) L.acc[threadIdx.x][k] = 0.f;
        if (!PREFETCH) {          // A/B: the gather between the two barriers, as before
            const int j1 = be - (int)threadIdx.x;
            id_cur = j1 >= start ? flatten_ids[j1] : -1;
            rec_next = load_rec(splats, id_cur);
        }
        {
            stage_splat(L.f, (int)threadIdx.x, rec_next, xc, yc);
            if (id_cur >= 0) {
                const float4 a = rec_next.a, bb = rec_next.bb;
                L.geo[threadIdx.x] = make_float4(a.x - xc, a.y - yc, a.z, a.w);
                L.geo2[threadIdx.x] = make_float2(bb.x, bb.y > 0.f ? 1.f / bb.y : 0.f);
                L.id[threadIdx.x] = id_cur;
            }
        }
        __syncthreads();
        if (PREFETCH) {
            id_cur = id_next;
            rec_next = load_rec(splats, id_next);
            const int j2 = be - 2 * BLOCK - (int)threadIdx.x;
            id_next = j2 >= start ? flatten_ids[j2] : -1;
        }
        const int bsz = min(BLOCK, be - start + 1);
        const int k0 = max(0, be - wmax);            // wave-uniform: nothing in this wave is live before slot k0
        for (int sb = k0 / SUB; sb * SUB < bsz; sb++) {
            float s[SUB];
            eval_sub_batch(L.f, sb, lane, basis, s);
            if (be - sb * SUB <= wmin)
                bwd_sub_batch<ABSGRAD, true>(L, s, sb, be, lane, bin_final, has, px, vrgb, tail, T, bufdot);
            else
                bwd_sub_batch<ABSGRAD, false>(L, s, sb, be, lane, bin_final, has, px, vrgb, tail, T, bufdot);
        }
        __syncthreads();
        // flush: lane -> (record = lane >> 4, dword = lane & 15): 4 records = 4 x 64-B requests per instruction.
        // The gradients of (x, y, conic A, B, C, opacity) follow from the moments about the tile centre:
        //   sum q dx = mx M - Mu, sum q dx^2 = mx^2 M - 2 mx Mu + Muu, ...   (dx = mx - u, dy = my - v)
        for (int sl = wv * 64; sl < wv * 64 + 64; sl += 4) {
            const int slot = sl + (lane >> 4);
            const int comp = lane & 15;
            // a slot no pixel touched (or whose sums all cancelled to exactly zero) has nothing to add: its 16 lanes
            // find that out together from the raw sums, no per-visit "touched" flag
            const bool any_raw = slot < bsz && comp < AC_STRIDE && L.acc[slot][comp < AC_STRIDE ? comp : 0] != 0.f;
            const unsigned long long nz = wave_ballot(any_raw);
            const bool touched = ((nz >> (lane & 48)) & 0xFFFFull) != 0ull;
            if (slot < bsz && touched && comp < (ABSGRAD ? GR_DEPTH : GR_ABSX)) {
                const float* ac = L.acc[slot];
                const float M = ac[AC_Q], Mu = ac[AC_QU], Mv = ac[AC_QV];
                const float4 ge = L.geo[slot];
                const float2 g2 = L.geo2[slot];
                const float mx = ge.x, my = ge.y;
                const float sdx = mx * M - Mu, sdy = my * M - Mv;                 // sum q dx, sum q dy
                float val;
                switch (comp) {
                    case GR_X: val = -(ge.z * sdx + ge.w * sdy); break;
                    case GR_Y: val = -(ge.w * sdx + g2.x * sdy); break;
                    case GR_CA: val = -0.5f * (mx * (mx * M - 2.f * Mu) + ac[AC_QUU]); break;
                    case GR_CB: val = -(mx * (my * M - Mv) - my * Mu + ac[AC_QUV]); break;
                    case GR_CC: val = -0.5f * (my * (my * M - 2.f * Mv) + ac[AC_QVV]); break;
                    case GR_OPA: val = M * g2.y; break;
                    case GR_R: val = ac[AC_R]; break;
                    case GR_G: val = ac[AC_G]; break;
                    case GR_B: val = ac[AC_B]; break;
                    case GR_ABSX: val = ac[AC_ABSX]; break;
                    default: val = ac[AC_ABSY]; break;
                }
                atomicAdd(&v_splats[(size_t)L.id[slot] * GRAD_STRIDE + comp], val);
            }
        }
    }
}


#endif  // MI3DGS_EXPERIMENTS

}  // namespace mfma_raster

int mi_rasterize_fwd_mfma(int n_tiles, int width, int height, int tile_width, int tile_height, const float* splats,
                          const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                          const float* backgrounds, float* render, float* alphas, int32_t* last_ids, void* seg_ws, size_t seg_ws_bytes,
                          hipStream_t st) {
    using namespace mfma_raster;
    SegWs seg = {nullptr, nullptr, nullptr, nullptr, 0u, 0u};
    if (seg_ws) {
        MI_REQUIRE(seg_ws_layout(n_tiles, seg_ws, seg_ws_bytes, &seg), "rasterize_fwd: segment workspace too small (mi3dgs_raster_seg_workspace_bytes)");
        // (the work counter is clear: mi3dgs_raster_seg_workspace_init once, and every backward leaves it clear again)
        if (!seg_ws_in_use(n_tiles, seg_ws_bytes)) seg = SegWs{nullptr, nullptr, nullptr, nullptr, 0u, 0u};
    }
#define LAUNCH_FWD(BG)                                                                                                    \
    MI_LAUNCH("rasterize_fwd", (rasterize_fwd_kernel<BG>), dim3(raster_grid(n_tiles, tile_width)), dim3(BLOCK), 0, st, width, height, tile_width,     \
              tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, render, alphas, last_ids, raster_bands(), seg)
    if (backgrounds) LAUNCH_FWD(true); else LAUNCH_FWD(false);
#undef LAUNCH_FWD
    MI_LAUNCH_CHECK();
    return 0;
}

int mi_rasterize_bwd_mm(int n_tiles, int width, int height, int tile_width, int tile_height, long long n_gauss, const float* splats,
                        const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                        const float* backgrounds, const float* alphas, const int32_t* last_ids, const float* v_render,
                        const float* v_alphas, int absgrad, float* v_splats, int experiment, const float* render, void* seg_ws,
                        size_t seg_ws_bytes, hipStream_t st);

int mi_rasterize_bwd_mfma(int n_tiles, int width, int height, int tile_width, int tile_height, const float* splats,
                          const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                          const float* backgrounds, const float* alphas, const int32_t* last_ids, const float* v_render,
                          const float* v_alphas, int absgrad, float* v_splats, int mode, long long n_gauss, const float* render,
                          void* seg_ws, size_t seg_ws_bytes, hipStream_t st) {
    using namespace mfma_raster;
    // mode 1: the contraction on the matrix pipe (rasterize_bwd_mm.hip), the product path and the only one of the product
    // library.  Experiments build: 3 = cross-lane reduce-scatter in f32 (this file), 4 and 11..14 = variants of the product
    // kernel (rasterize_bwd_mm.hip: 4 three-term sums, 14 wave flush, 11..13 timing experiments with wrong results)
#ifdef MI3DGS_EXPERIMENTS
    if (mode == 3) {
#define LAUNCH_BWD(BG, AG)                                                                                                 \
    MI_LAUNCH("rasterize_bwd", (rasterize_bwd_kernel<BG, AG, false>), dim3(raster_grid(n_tiles, tile_width)), dim3(BLOCK), 0, st, width, height, tile_width, \
              tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, alphas, last_ids,        \
              v_render, v_alphas, v_splats, raster_bands())
        if (backgrounds) { if (absgrad) LAUNCH_BWD(true, true); else LAUNCH_BWD(true, false); }
        else { if (absgrad) LAUNCH_BWD(false, true); else LAUNCH_BWD(false, false); }
#undef LAUNCH_BWD
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    return mi_rasterize_bwd_mm(n_tiles, width, height, tile_width, tile_height, n_gauss, splats, isect_offsets, flatten_ids, n_isect_dev,
                               backgrounds, alphas, last_ids, v_render, v_alphas, absgrad, v_splats, mode == 1 ? 0 : mode, render, seg_ws,
                               seg_ws_bytes, st);
}

extern "C" size_t mi3dgs_raster_seg_workspace_bytes(int n_tiles, long long max_isect) {
    if (n_tiles <= 0 || max_isect < 0) return 0;
    return mfma_raster::seg_ws_bytes_for(n_tiles, max_isect);
}

extern "C" int mi3dgs_raster_seg_workspace_init(void* seg_workspace, size_t seg_workspace_bytes, void* stream) {
    MI_REQUIRE(seg_workspace && seg_workspace_bytes >= 256, "raster_seg_workspace_init: no workspace");
    MI_HIP(hipMemsetAsync(seg_workspace, 0, 256, (hipStream_t)stream));
    return 0;
}
