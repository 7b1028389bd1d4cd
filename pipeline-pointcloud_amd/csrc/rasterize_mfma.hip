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
    int32_t* __restrict__ last_ids, SegWs seg) {
    __shared__ Staged L;
    __shared__ uint32_t s_slot;
    const int t = (int)blockIdx.x;
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
                const uint32_t slot = atomicAdd(&seg.ctl[SEG_CTL_ITEMS], 1u);
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
                // one visit ahead: its latency hides behind this visit.  (Round 4 tried the backward's scheme here -- one read per
                // sub-batch, v_readlane in the visits that composite: S2 137.5 -> 141.6 us, S1 56.0 -> 56.8, wolf 87.1 -> 81.6;
                // the forward has no stores for the read to queue behind, and it stays as it is.)
                col_next = lds_rgb(uni, i + 1);
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

// ---------------------------------------------------------------------------------------- forward in segments
// One block per tile walks the tile's list serially: a launch lasts as long as its heaviest tile.  On a dense object-centric scene
// (gsplat's MCMC strategy at its cap of 1 M Gaussians, 960 x 720: 2.3 M intersections, a tenth of the tiles with 2 500 - 5 200 entries,
// every one of them reached) rasterize_fwd took 555 us of a 1.9 ms step for work that fills the device for a fifth of that time.
// With the segment workspace the heavy tiles are walked as segments of 256 entries side by side:
//   fwd_plan_kernel             a thread per tile: a tile of more than 256 entries books one work item per 256 entries;
//   rasterize_fwd_seg_kernel    the first blocks of the grid are segment workers: every segment from T = 1, C = 0: its transmittance
//                               product P, its colour C, its last contributor, and whether a pixel ran into the transmittance stop
//                               inside it; the other blocks are the tiles of at most 256 entries, walked as always;
//   fwd_combine_kernel          per heavy tile and pixel, segment after segment: T_in P > 1e-4 and no stop inside -> the segment is
//                               taken whole (C += T_in C_s, T_in *= P); otherwise the pixel stops inside THAT segment: its state
//                               at the segment's start is left in the segment's slot.  Also leaves the backward's checkpoints
//                               at the boundaries pixels walked past;
//   fwd_finish_kernel           the segments pixels stop in, walked again for those pixels from their true state, entry by entry
//                               as the serial forward does; all of them side by side (strung behind one another in the combine
//                               pass they took 298 us of a 487 us forward).
// Same result as the serial walk up to the rounding of T_in * (local product) against the running product.

// the serial forward's walk over entries [lo, hi) for the block's pixels
__device__ __forceinline__ void fwd_walk_range(Staged& L, const float* __restrict__ splats, const int32_t* __restrict__ flatten_ids, int lo,
                                               int hi, float xc, float yc, int lane, const Basis& basis, float& T, float& cr, float& cg,
                                               float& cb, int& cur, unsigned long long& live) {
    for (int bs = lo; bs < hi; bs += BLOCK) {
        if (!__syncthreads_or(live != 0ull)) break;
        {
            const int i1 = bs + (int)threadIdx.x;
            stage_splat(L, (int)threadIdx.x, load_rec(splats, i1 < hi ? flatten_ids[i1] : -1), xc, yc);
        }
        __syncthreads();
        const int bsz = min(BLOCK, hi - bs);
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
                col_next = lds_rgb(uni, i + 1);
                const unsigned long long hit = mask_ge(s[i], LOG2_ALPHA_THRESHOLD) & live;
                if (hit == 0ull) continue;
                const float alpha = alpha_of(s[i]);
                const float wgt = alpha * T;
                const float nT = T - wgt;
                const unsigned long long comp = mask_gt(nT, T_STOP) & hit;
                live &= ~(hit & ~comp);
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
}

struct TileGeom { int cam, tx, ty, px_i, py_i; bool inside; float xc, yc; };
__device__ __forceinline__ TileGeom tile_geom(int t, int tw, int th, int W, int H, int wv, int lane) {
    TileGeom g;
    g.cam = t / (tw * th);
    const int tile_in = t - g.cam * (tw * th);
    g.ty = tile_in / tw; g.tx = tile_in - g.ty * tw;
    int lx, ly;
    pixel_of_lane(wv, lane, lx, ly);
    g.px_i = g.tx * TILE + lx; g.py_i = g.ty * TILE + ly;
    g.inside = g.px_i < W && g.py_i < H;
    g.xc = (float)(g.tx * TILE) + 8.f; g.yc = (float)(g.ty * TILE) + 8.f;
    return g;
}

#ifdef MI3DGS_OS_STAMPS
// Probe build only (tools/fwd_seg_probe.py): per block of rasterize_fwd_seg_kernel start / end wall-clock stamps (100 MHz), entries walked.
__device__ unsigned long long g_rf_stamps[8192][3];
#define RF_STAMP(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_rf_stamps[blockIdx.x][i] = (v); } while (0)
}  // namespace mfma_raster
extern "C" int mi3dgs_debug_read_rf_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mfma_raster::g_rf_stamps), bytes < sizeof(mfma_raster::g_rf_stamps) ? bytes : sizeof(mfma_raster::g_rf_stamps));
}
namespace mfma_raster {
#else
#define RF_STAMP(i, v) do { } while (0)
#endif

constexpr int FWD_SEG_WORKERS = 2048;
constexpr int SEG_NONE = 0x3fffffff, SEG_STOPPED = 0x40000000;

// One block: slots in tile order by a prefix sum, the two counters WRITTEN (a forward with no backward behind it -- an evaluation
// render -- leaves nothing for the next forward to trip over).
__global__ __launch_bounds__(1024) void fwd_plan_kernel(const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ n_isect_ptr,
                                                        int n_tiles_total, SegWs seg) {
    __shared__ unsigned long long s_wave[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long run = 0ull;                                   // low word: segments booked so far, high word: heavy tiles
    for (int t0 = 0; t0 < n_tiles_total; t0 += 1024) {
        const int t = t0 + (int)threadIdx.x;
        uint32_t n_seg = 0u;
        int start = 0;
        if (t < n_tiles_total) {
            start = tile_offsets[t];
            const int end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;
            if (end - start > SEG_MIN) n_seg = (uint32_t)((end - start + SEG_MIN - 1) / SEG_MIN);
        }
        const unsigned long long mine = n_seg ? ((1ull << 32) | n_seg) : 0ull;
        unsigned long long incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        unsigned long long before = run, total = 0ull;
        for (int w = 0; w < 16; w++) {
            const unsigned long long v = s_wave[w];
            if (w < wv) before += v;
            total += v;
        }
        if (n_seg) {
            const unsigned long long at = before + incl - mine;
            const uint32_t base = (uint32_t)at, hv = (uint32_t)(at >> 32);
            // a segment's item carries 0 entries until the combine pass has seen a pixel walk past its end boundary (the backward
            // skips those); base + n_seg <= cap: the capacity counts every segment of every tile
            for (uint32_t k = 0; k < n_seg; k++) seg.work[base + k] = make_uint4((uint32_t)t, (uint32_t)start + k * SEG_MIN, base + k, 0u);
            seg.tile_items[t] = make_uint2(base, n_seg);
            seg.heavy[hv] = (uint32_t)t;
        }
        run += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { seg.ctl[SEG_CTL_ITEMS] = (uint32_t)run; seg.ctl[SEG_CTL_HEAVY] = (uint32_t)(run >> 32); }
}

template <bool HAS_BG>
__global__ __launch_bounds__(BLOCK) void rasterize_fwd_seg_kernel(int W, int H, int tw, int th, const float* __restrict__ splats,
                                                                  const int32_t* __restrict__ tile_offsets,
                                                                  const int32_t* __restrict__ flatten_ids,
                                                                  const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
                                                                  const float* __restrict__ backgrounds, float* __restrict__ render,
                                                                  float* __restrict__ alphas, int32_t* __restrict__ last_ids,
                                                                  int n_workers, SegWs seg) {
    __shared__ Staged L;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const Basis basis = make_basis(wv, lane);
    RF_STAMP(0, wall_clock64()); RF_STAMP(1, 0ull); RF_STAMP(2, 0ull);
    if ((int)blockIdx.x < n_workers) {
        const int n_items = (int)min(seg.ctl[SEG_CTL_ITEMS], seg.cap);
        for (int item = blockIdx.x; item < n_items; item += n_workers) {
            const uint4 wk = seg.work[item];
            const int t = (int)wk.x, lo = (int)wk.y;
            const int end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;
            const int hi = min(lo + SEG_MIN, end);
            const TileGeom g = tile_geom(t, tw, th, W, H, wv, lane);
            float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
            int cur = -1;
            const unsigned long long live0 = wave_ballot(g.inside);
            unsigned long long live = live0;
            __syncthreads();                                   // the walk before may still be reading L
            fwd_walk_range(L, splats, flatten_ids, lo, hi, g.xc, g.yc, lane, basis, T, cr, cg, cb, cur, live);
            const bool stopped = ((live0 & ~live) >> lane) & 1ull;
            seg.ckpt[(size_t)wk.z * BLOCK + threadIdx.x] = make_float4(T, cr, cg, cb);
            seg.local_last[(size_t)wk.z * BLOCK + threadIdx.x] = (cur < 0 ? SEG_NONE : cur) | (stopped ? SEG_STOPPED : 0);
            RF_STAMP(2, (unsigned long long)(hi - lo));
        }
        RF_STAMP(1, wall_clock64());
        return;
    }
    const int t = (int)blockIdx.x - n_workers;
    const int start = tile_offsets[t];
    const int end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;
    if (end - start > SEG_MIN) return;                         // in segments: the workers and the combine pass
    const TileGeom g = tile_geom(t, tw, th, W, H, wv, lane);
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
    int cur = 0;
    unsigned long long live = wave_ballot(g.inside);
    fwd_walk_range(L, splats, flatten_ids, start, end, g.xc, g.yc, lane, basis, T, cr, cg, cb, cur, live);
    if (g.inside) {
        const size_t pix = ((size_t)g.cam * H + g.py_i) * W + g.px_i;
        if (HAS_BG) {
            const float* bg = backgrounds + 3 * g.cam;
            cr += T * bg[0]; cg += T * bg[1]; cb += T * bg[2];
        }
        render[3 * pix] = cr; render[3 * pix + 1] = cg; render[3 * pix + 2] = cb;
        alphas[pix] = 1.f - T;
        last_ids[pix] = cur;
    }
    if (threadIdx.x == 0) seg.tile_skip[t] = 0u;
    RF_STAMP(1, wall_clock64()); RF_STAMP(2, (unsigned long long)(end - start));
}

constexpr int SEG_PENDING = (int)0x80000000;

// No walking here: a thread per pixel strings the segments' results together.  A pixel that can not take a segment whole (it stops
// inside it) leaves its state at the segment's start in the segment's own slot and is finished here: fwd_finish_kernel walks
// that segment for it.
template <bool HAS_BG>
__global__ __launch_bounds__(BLOCK) void fwd_combine_kernel(int W, int H, int tw, int th, const float* __restrict__ backgrounds,
                                                            float* __restrict__ render, float* __restrict__ alphas,
                                                            int32_t* __restrict__ last_ids, SegWs seg) {
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int n_heavy = (int)seg.ctl[SEG_CTL_HEAVY];
    for (int h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        const int t = (int)seg.heavy[h];
        const uint2 ti = seg.tile_items[t];
        const TileGeom g = tile_geom(t, tw, th, W, H, wv, lane);
        float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
        int cur = 0;
        bool mine = g.inside, pending = false;
        uint32_t nb = 0;
        // The per-segment results are fetched PF segments ahead (a dependent chain of global loads, one per segment, was 2.5 us a
        // segment); per segment one barrier carries one block-wide fact: somebody is still compositing here.
        constexpr int PF = 8;
        bool go = true;
        for (uint32_t k = 0; go && k < ti.y; k += PF) {
            float4 lc[PF];
            int enc[PF];
#pragma unroll
            for (int j = 0; j < PF; j++)
                if (k + j < ti.y) {
                    lc[j] = seg.ckpt[(size_t)(ti.x + k + j) * BLOCK + threadIdx.x];
                    enc[j] = seg.local_last[(size_t)(ti.x + k + j) * BLOCK + threadIdx.x];
                }
#pragma unroll
            for (int j = 0; j < PF; j++) {
                if (go && k + j < ti.y) {
                    const uint32_t kk = k + j, slot = ti.x + kk;
                    const int f = __syncthreads_or(mine);
                    if (!f) {
                        go = false;
                    } else {
                        if (kk > 0) {
                            // somebody composites on beyond the boundary in front of this segment: the backward's checkpoint (for
                            // the pixels that do: the backward reads no other) and work item for the segment before
                            if (mine) seg.ckpt[(size_t)(slot - 1) * BLOCK + threadIdx.x] = make_float4(T, cr, cg, cb);
                            if (threadIdx.x == 0) seg.work[slot - 1].w = (uint32_t)SEG_MIN;
                            nb++;
                        }
                        if (mine) {
                            // take the segment whole: the pixel does not stop inside it (the running product only falls: its value
                            // behind the segment bounds every value inside)
                            // (the tile's first segment was walked from the pixel's true state, T = 1: taken as it is, stop or no stop)
                            const bool stopped = (enc[j] & SEG_STOPPED) != 0;
                            if (kk == 0 || (!stopped && T * lc[j].x > T_STOP)) {
                                cr = __builtin_fmaf(T, lc[j].y, cr); cg = __builtin_fmaf(T, lc[j].z, cg); cb = __builtin_fmaf(T, lc[j].w, cb);
                                T *= lc[j].x;
                                if ((enc[j] & SEG_NONE) != SEG_NONE) cur = enc[j] & SEG_NONE;
                                if (stopped) mine = false;
                            } else {
                                seg.ckpt[(size_t)slot * BLOCK + threadIdx.x] = make_float4(T, cr, cg, cb);
                                seg.local_last[(size_t)slot * BLOCK + threadIdx.x] = cur | SEG_PENDING;
                                mine = false; pending = true;
                            }
                        }
                    }
                }
            }
        }
        if (g.inside && !pending) {
            const size_t pix = ((size_t)g.cam * H + g.py_i) * W + g.px_i;
            if (HAS_BG) {
                const float* bg = backgrounds + 3 * g.cam;
                cr += T * bg[0]; cg += T * bg[1]; cb += T * bg[2];
            }
            render[3 * pix] = cr; render[3 * pix + 1] = cg; render[3 * pix + 2] = cb;
            alphas[pix] = 1.f - T;
            last_ids[pix] = cur;
        }
        if (threadIdx.x == 0) seg.tile_skip[t] = nb * SEG_MIN;
    }
}

// The segments some pixel stops in, walked for those pixels from their true state at the segment's start, entry by entry as the
// serial forward does -- all such segments side by side.  (A pixel that by rounding does NOT stop in this walk -- its T_in P was
// within an ulp of 1e-4 -- ends here all the same: what it leaves out is bounded by T = 1e-4.)
template <bool HAS_BG>
__global__ __launch_bounds__(BLOCK) void fwd_finish_kernel(int W, int H, int tw, int th, const float* __restrict__ splats,
                                                           const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
                                                           const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
                                                           const float* __restrict__ backgrounds, float* __restrict__ render,
                                                           float* __restrict__ alphas, int32_t* __restrict__ last_ids, SegWs seg) {
    __shared__ Staged L;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const Basis basis = make_basis(wv, lane);
    const int n_items = (int)min(seg.ctl[SEG_CTL_ITEMS], seg.cap);
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const uint4 wk = seg.work[item];
        const int enc = seg.local_last[(size_t)wk.z * BLOCK + threadIdx.x];
        const bool pending = enc < 0;
        if (!__syncthreads_or(pending)) continue;          // (also: the walk before has finished reading L)
        const int t = (int)wk.x, lo = (int)wk.y;
        const int end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;
        const int hi = min(lo + SEG_MIN, end);
        const TileGeom g = tile_geom(t, tw, th, W, H, wv, lane);
        const float4 ck = seg.ckpt[(size_t)wk.z * BLOCK + threadIdx.x];
        float T = ck.x, cr = ck.y, cg = ck.z, cb = ck.w;
        int cur = enc & SEG_NONE;
        unsigned long long live = wave_ballot(pending);
        fwd_walk_range(L, splats, flatten_ids, lo, hi, g.xc, g.yc, lane, basis, T, cr, cg, cb, cur, live);
        if (pending) {
            const size_t pix = ((size_t)g.cam * H + g.py_i) * W + g.px_i;
            if (HAS_BG) {
                const float* bg = backgrounds + 3 * g.cam;
                cr += T * bg[0]; cg += T * bg[1]; cb += T * bg[2];
            }
            render[3 * pix] = cr; render[3 * pix + 1] = cg; render[3 * pix + 2] = cb;
            alphas[pix] = 1.f - T;
            last_ids[pix] = cur;
        }
    }
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
    const float* __restrict__ v_render, const float* __restrict__ v_alphas, float* __restrict__ v_splats) {
    __shared__ StagedBwd L;
    const int t = (int)blockIdx.x;
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

// 1: with a segment workspace the forward walks tiles of more than 256 entries as segments side by side.  Off unless asked for:
// the serial walk stops where a tile's pixels are saturated, the segments are all walked before anybody knows where that is.
// Measured (profiles/r03_fwd_segments.txt): lists walked to their ends (MCMC at its cap after refinement has stopped, 1.9 M
// intersections on 2 700 tiles) 555 -> 283 us; the same model while it trains -4 .. -10 % of the step rate, default strategy -11 %.
int g_fwd_segments = 0;
extern "C" int mi3dgs_debug_set_raster_fwd_segments(int on) {
    g_fwd_segments = on ? 1 : 0;
    return 0;
}
int mi_rasterize_fwd_mfma(int n_tiles, int width, int height, int tile_width, int tile_height, const float* splats,
                          const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                          const float* backgrounds, float* render, float* alphas, int32_t* last_ids, void* seg_ws, size_t seg_ws_bytes,
                          hipStream_t st) {
    using namespace mfma_raster;
    SegWs seg = {};
    if (seg_ws) {
        MI_REQUIRE(seg_ws_layout(n_tiles, seg_ws, seg_ws_bytes, &seg), "rasterize_fwd: segment workspace too small (mi3dgs_raster_seg_workspace_bytes)");
        // (the work counter is clear: mi3dgs_raster_seg_workspace_init once, and every backward leaves it clear again)
        if (!seg_ws_in_use(n_tiles, seg_ws_bytes)) seg = SegWs{};
    }
    if (seg.ckpt != nullptr && g_fwd_segments) {
        const int n_workers = (int)min((size_t)FWD_SEG_WORKERS, (size_t)seg.cap);
        MI_LAUNCH("rasterize_fwd_plan", fwd_plan_kernel, dim3(1), dim3(1024), 0, st, isect_offsets, n_isect_dev, n_tiles, seg);
#define LAUNCH_SEG(BG)                                                                                                    \
    MI_LAUNCH("rasterize_fwd", (rasterize_fwd_seg_kernel<BG>), dim3(n_workers + n_tiles), dim3(BLOCK), 0, st, width,        \
              height, tile_width, tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, render, alphas, last_ids, \
              n_workers, seg);                                                                               \
    MI_LAUNCH("rasterize_fwd_combine", fwd_combine_kernel<BG>, dim3(512), dim3(BLOCK), 0, st, width, height, tile_width, tile_height,           \
              backgrounds, render, alphas, last_ids, seg);                                                                   \
    MI_LAUNCH("rasterize_fwd_finish", fwd_finish_kernel<BG>, dim3(n_workers), dim3(BLOCK), 0, st, width, height, tile_width, tile_height, splats, \
              isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, render, alphas, last_ids, seg)
        if (backgrounds) { LAUNCH_SEG(true); } else { LAUNCH_SEG(false); }
#undef LAUNCH_SEG
        MI_LAUNCH_CHECK();
        return 0;
    }
#define LAUNCH_FWD(BG)                                                                                                    \
    MI_LAUNCH("rasterize_fwd", (rasterize_fwd_kernel<BG>), dim3(n_tiles), dim3(BLOCK), 0, st, width, height, tile_width,     \
              tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, render, alphas, last_ids, seg)
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
    // library.  Experiments build: 3 = cross-lane reduce-scatter in f32 (this file), 4 = the product kernel with three-term
    // sums (rasterize_bwd_mm.hip)
#ifdef MI3DGS_EXPERIMENTS
    if (mode == 3) {
#define LAUNCH_BWD(BG, AG)                                                                                                 \
    MI_LAUNCH("rasterize_bwd", (rasterize_bwd_kernel<BG, AG, false>), dim3(n_tiles), dim3(BLOCK), 0, st, width, height, tile_width, \
              tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, alphas, last_ids,        \
              v_render, v_alphas, v_splats)
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
