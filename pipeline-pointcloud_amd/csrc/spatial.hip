// Input-side kernels of the training path, gfx950 only.
//
//  1. Exact k nearest neighbours of the SfM points (k <= 4): the initial Gaussian scales are
//     log(mean distance to the 3 nearest neighbours) (splatfacto / simple_trainer initialisation
//     reached through source/container/src/main.py:1271 and :1328).  Upstream runs sklearn on the
//     CPU; here: uniform grid over the robust bounding box, points radix-sorted by cell (the
//     library's own sort), one thread per point walking cubic shells of cells until the k-th
//     best distance is provably final.  O(N) memory, no host synchronisation.
//  2. Area-average image downscale (what `ensure_downscaled_images`,
//     source/container/src/main.py:419-481, asks cv2.INTER_AREA for) and u8 -> f32 target
//     conversion for the device image cache.
#include "common.h"

#include <math.h>

extern "C" size_t mi3dgs_sort_workspace_bytes(long long n);
extern "C" int mi3dgs_sort_pairs_u32(uint32_t* keys, uint32_t* vals, long long n, int nbits, void* workspace,
                                     size_t workspace_bytes, void* stream);

namespace {

// ------------------------------------------------------------------------------- k-NN
constexpr uint32_t KNN_MAX_CELLS = 1u << 24;      // 64 MB of cell starts
constexpr int KNN_CELL_BITS = 24;
constexpr int KNN_MAX_DIM = 1024;

struct KnnGrid {
    float origin[3];
    float h, inv_h;
    int dims[3];
    uint32_t ncells;
    float ext[3];          // extent the grid has to span
    uint32_t occupied;     // number of non-empty cells, counted after the first sort
    uint32_t n_todo;       // queries the shell walk gave up on (finished by knn_brute_kernel)
};

// word offsets into the statistics block: sums, sums of squares, count, min / max (ordered ints),
// then the inclusion window [lo, hi] per axis of the current trimming round
enum { ST_SUM = 0, ST_SQ = 3, ST_CNT = 6, ST_MIN = 7, ST_MAX = 10, ST_WLO = 13, ST_WHI = 16, ST_WORDS = 20 };
constexpr int KNN_TRIM_ROUNDS = 3;
constexpr int KNN_RMAX = 4;
constexpr int KNN_BRUTE_BLOCKS = 2048;

// order-preserving float <-> uint map so that min / max can be integer atomics
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ void knn_reset_stats(uint32_t* st) {
    for (int t = 0; t < 7; t++) st[t] = 0;
    for (int a = 0; a < 3; a++) { st[ST_MIN + a] = 0xffffffffu; st[ST_MAX + a] = 0; }
}

__global__ void knn_init_kernel(uint32_t* __restrict__ st) {
    knn_reset_stats(st);
    float* sf = reinterpret_cast<float*>(st);
    for (int a = 0; a < 3; a++) { sf[ST_WLO + a] = -INFINITY; sf[ST_WHI + a] = INFINITY; }
}

// moments and bounds of the points inside the current window (all three coordinates finite and
// within it)
__global__ __launch_bounds__(256) void knn_stats_kernel(uint32_t n, const float* __restrict__ pts,
                                                        uint32_t* __restrict__ st) {
    const float* sf = reinterpret_cast<const float*>(st);
    float wlo[3], whi[3];
    for (int a = 0; a < 3; a++) { wlo[a] = sf[ST_WLO + a]; whi[a] = sf[ST_WHI + a]; }
    float s[3] = {0, 0, 0}, q[3] = {0, 0, 0}, lo[3] = {INFINITY, INFINITY, INFINITY},
          hi[3] = {-INFINITY, -INFINITY, -INFINITY}, cnt = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v[3];
        bool in = true;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            v[a] = pts[(size_t)i * 3 + a];
            in = in && isfinite(v[a]) && v[a] >= wlo[a] && v[a] <= whi[a];
        }
        if (!in) continue;
        cnt += 1.f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            s[a] += v[a];
            q[a] += v[a] * v[a];
            lo[a] = fminf(lo[a], v[a]);
            hi[a] = fmaxf(hi[a], v[a]);
        }
    }
    cnt = wave_sum_all(cnt);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        s[a] = wave_sum_all(s[a]);
        q[a] = wave_sum_all(q[a]);
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    }
    // one set of atomics per block (13 same-address atomics per wave serialise at ~10 ns each)
    __shared__ float red[4][13];
    int wv = threadIdx.x >> 6;
    if (lane_id() == 0) {
        red[wv][0] = cnt;
        for (int a = 0; a < 3; a++) {
            red[wv][1 + a] = s[a]; red[wv][4 + a] = q[a]; red[wv][7 + a] = lo[a]; red[wv][10 + a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 13) {
        int t = threadIdx.x;
        float v = red[0][t];
        for (int k = 1; k < 4; k++)
            v = (t < 7) ? v + red[k][t] : (t < 10 ? fminf(v, red[k][t]) : fmaxf(v, red[k][t]));
        if (t == 0) atomicAdd(reinterpret_cast<float*>(st) + ST_CNT, v);
        else if (t < 4) atomicAdd(reinterpret_cast<float*>(st) + ST_SUM + (t - 1), v);
        else if (t < 7) atomicAdd(reinterpret_cast<float*>(st) + ST_SQ + (t - 4), v);
        else if (t < 10) atomicMin(st + ST_MIN + (t - 7), f2ord(v));
        else atomicMax(st + ST_MAX + (t - 10), f2ord(v));
    }
}

// dims for cell size h (grown by 26 % a step until the grid fits the caps and still spans ext);
// false = degenerate, the caller falls back to one cell
__device__ bool knn_fit_dims(const float (&ext)[3], float& h, int (&d)[3]) {
    for (int it = 0; it < 96; it++) {
        unsigned long long prod = 1;
        bool covered = true;
        for (int a = 0; a < 3; a++) {
            float want = ceilf(ext[a] / h);
            d[a] = (int)fminf(fmaxf(want, 1.f), (float)KNN_MAX_DIM);
            prod *= (unsigned long long)d[a];
            covered = covered && (float)d[a] * h >= ext[a];
        }
        if (prod <= KNN_MAX_CELLS && covered) return h > 0.f && isfinite(h);
        h *= 1.26f;
    }
    return false;
}

__device__ void knn_store_grid(KnnGrid* g, bool ok, float h, const int (&d)[3]) {
    g->dims[0] = ok ? d[0] : 1;
    g->dims[1] = ok ? d[1] : 1;
    g->dims[2] = ok ? d[2] : 1;
    g->h = ok ? h : 1.f;
    g->inv_h = ok ? 1.f / h : 1.f;
    g->ncells = (uint32_t)g->dims[0] * (uint32_t)g->dims[1] * (uint32_t)g->dims[2];
    g->occupied = 0;
    g->n_todo = 0;
}

// One thread.  Trimming round: the next window is [max(min, mean - 3 sd), min(max, mean + 3 sd)] of
// the points in the current one (SfM clouds carry outliers hundreds of scene radii away; three
// rounds home in on the bulk).  Last round (`final`): the grid over that window; whatever lies
// outside is clamped into the border cells, which the search treats as unbounded.
__global__ void knn_grid_kernel(uint32_t n, uint32_t* __restrict__ st, KnnGrid* __restrict__ g, int final) {
    float* sf = reinterpret_cast<float*>(st);
    float cnt = fmaxf(sf[ST_CNT], 1.f);
    float lo[3], ext[3], emax = 0.f, amax = 0.f;
    for (int a = 0; a < 3; a++) {
        float mean = sf[ST_SUM + a] / cnt;
        float var = fmaxf(sf[ST_SQ + a] / cnt - mean * mean, 0.f);
        float sd = sqrtf(var);
        float mn = ord2f(st[ST_MIN + a]), mx = ord2f(st[ST_MAX + a]);
        if (!(mn <= mx)) { mn = 0.f; mx = 0.f; }          // nothing finite inside the window
        float l = fmaxf(mn, mean - 3.f * sd), h = fminf(mx, mean + 3.f * sd);
        if (!(l <= h)) { l = mn; h = mx; }
        lo[a] = l;
        ext[a] = h - l;
        emax = fmaxf(emax, ext[a]);
        amax = fmaxf(amax, fmaxf(fabsf(l), fabsf(h)));
        sf[ST_WLO + a] = l;
        sf[ST_WHI + a] = h;
    }
    knn_reset_stats(st);
    if (!final) return;
    // cells must stay resolvable in float32 around the coordinates they sit at
    emax = fmaxf(fmaxf(emax, 1e-5f * amax), 1e-30f);
    float target = fminf(fmaxf((float)n * 0.5f, 1.f), (float)KNN_MAX_CELLS);
    float h = 1.f / cbrtf(target);
    for (int a = 0; a < 3; a++) {
        ext[a] = fmaxf(ext[a], 1e-3f * emax);             // flat clouds: at least one thin layer of cells
        h *= cbrtf(ext[a]);                                // cbrt(volume / target) without forming the volume
    }
    h = fmaxf(h, 1e-6f * emax);
    int d[3] = {1, 1, 1};
    bool ok = knn_fit_dims(ext, h, d);
    for (int a = 0; a < 3; a++) {
        g->origin[a] = lo[a];
        g->ext[a] = ext[a];
    }
    knn_store_grid(g, ok, h, d);
}

// The first grid assumes the points fill their volume.  SfM points lie on surfaces, so most cells
// stay empty and the occupied ones are crowded: after the first sort the occupancy is known, and
// the cell size is cut so that an occupied cell holds ~3 points (a surface's cell count grows
// with 1/h^2, hence the square root).  One thread.
__global__ void knn_refine_kernel(uint32_t n, KnnGrid* __restrict__ g) {
    float occ = (float)n / fmaxf((float)g->occupied, 1.f);
    if (occ <= 6.f || g->ncells <= 1) { g->occupied = 0; return; }
    float ext[3] = {g->ext[0], g->ext[1], g->ext[2]};
    float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
    float h = fmaxf(g->h / fminf(sqrtf(occ / 3.f), 16.f), 1e-6f * emax);
    int d[3];
    bool ok = knn_fit_dims(ext, h, d);
    knn_store_grid(g, ok, h, d);
}

__global__ __launch_bounds__(256) void knn_occupancy_kernel(uint32_t n, const uint32_t* __restrict__ keys,
                                                            KnnGrid* __restrict__ g) {
    float heads = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        heads += (i == 0 || keys[i] != keys[i - 1]) ? 1.f : 0.f;
    heads = wave_sum_all(heads);                           // exact below 2^24 per wave
    if (lane_id() == 0 && heads > 0.f) atomicAdd(&g->occupied, (uint32_t)heads);
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int dim) {
    float c = floorf((v - o) * inv_h);
    return (int)fminf(fmaxf(c, 0.f), (float)(dim - 1));      // NaN -> 0
}

__global__ __launch_bounds__(256) void knn_cell_kernel(uint32_t n, const float* __restrict__ pts,
                                                       const KnnGrid* __restrict__ gp, uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    KnnGrid g = *gp;
    int cx = cell_coord(pts[(size_t)i * 3 + 0], g.origin[0], g.inv_h, g.dims[0]);
    int cy = cell_coord(pts[(size_t)i * 3 + 1], g.origin[1], g.inv_h, g.dims[1]);
    int cz = cell_coord(pts[(size_t)i * 3 + 2], g.origin[2], g.inv_h, g.dims[2]);
    keys[i] = ((uint32_t)cz * g.dims[1] + cy) * g.dims[0] + cx;
    vals[i] = i;
}

// start[c] = first sorted position whose cell >= c, for c in [0, ncells]; also gathers the points
// into sorted order as float4 (xyz, original index) so that the search reads 16-byte records.
__global__ __launch_bounds__(256) void knn_starts_kernel(uint32_t n, const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals,
                                                         const float* __restrict__ pts, const KnnGrid* __restrict__ gp,
                                                         uint32_t* __restrict__ start, float4* __restrict__ spts) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t ncells = gp->ncells;
    uint32_t k = keys[i], v = vals[i];
    spts[i] = make_float4(pts[(size_t)v * 3], pts[(size_t)v * 3 + 1], pts[(size_t)v * 3 + 2], __uint_as_float(v));
    uint32_t first = (i == 0) ? 0u : keys[i - 1] + 1u;
    for (uint32_t c = first; c <= k; c++) start[c] = i;
    if (i == n - 1)
        for (uint32_t c = k + 1; c <= ncells; c++) start[c] = n;
}

template <int K>
__device__ __forceinline__ void knn_insert(float (&bd)[K], uint32_t (&bi)[K], float d, uint32_t id) {
    if (d >= bd[K - 1]) return;
    bd[K - 1] = d;
    bi[K - 1] = id;
#pragma unroll
    for (int j = K - 1; j > 0; j--) {
        if (bd[j] < bd[j - 1]) {
            float td = bd[j]; bd[j] = bd[j - 1]; bd[j - 1] = td;
            uint32_t ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void knn_query_kernel(uint32_t n, const float4* __restrict__ spts,
                                                        const uint32_t* __restrict__ start,
                                                        KnnGrid* __restrict__ gp, uint32_t* __restrict__ todo,
                                                        float* __restrict__ out_d2, int32_t* __restrict__ out_idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KnnGrid g = *gp;
    const float4 me = spts[i];
    const uint32_t self = __float_as_uint(me.w);
    const float q[3] = {me.x, me.y, me.z};
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) c[a] = cell_coord(q[a], g.origin[a], g.inv_h, g.dims[a]);
    float bd[K];
    uint32_t bi[K];
#pragma unroll
    for (int j = 0; j < K; j++) { bd[j] = INFINITY; bi[j] = 0xffffffffu; }
    // Shells up to radius KNN_RMAX; a query still open after that (an outlier, a point in a sparse
    // fringe) would walk O(r^2) mostly empty rows per further shell on ONE lane: it is handed to
    // knn_brute_kernel, where a whole block scans the cloud for it.
    bool done = false;
    for (int r = 0; r <= KNN_RMAX; r++) {
        // shell at Chebyshev radius r: rows (z, y) with |dz| == r or |dy| == r take the whole x run,
        // which is contiguous in the sorted order; inner rows only the two end cells.
        int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dims[2] - 1);
        int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dims[1] - 1);
        int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dims[0] - 1);
        for (int z = z0; z <= z1; z++) {
            bool zface = (z == c[2] - r) || (z == c[2] + r);
            for (int y = y0; y <= y1; y++) {
                bool full = zface || (y == c[1] - r) || (y == c[1] + r);
                uint32_t row = ((uint32_t)z * g.dims[1] + y) * g.dims[0];
                int nseg = full ? 1 : 2;
                for (int sgm = 0; sgm < nseg; sgm++) {
                    int xa, xb;
                    if (full) { xa = x0; xb = x1; }
                    else if (sgm == 0) { xa = xb = c[0] - r; if (xa < 0) continue; }
                    else { xa = xb = c[0] + r; if (xa >= g.dims[0] || r == 0) continue; }
                    uint32_t s = start[row + xa], e = start[row + xb + 1];
                    for (uint32_t j = s; j < e; j++) {
                        float4 p = spts[j];
                        float dx = p.x - q[0], dy = p.y - q[1], dz = p.z - q[2];
                        float d = dx * dx + dy * dy + dz * dz;
                        uint32_t id = __float_as_uint(p.w);
                        if (id != self) knn_insert<K>(bd, bi, d, id);
                    }
                }
            }
        }
        // every point not visited yet lies beyond one of the open faces of the visited box; distances
        // are taken relative to the grid origin so that the rounding slack scales with the grid, not
        // with how far from zero the cloud sits
        float bound = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float qo = q[a] - g.origin[a];
            float far = (float)(c[a] + r + 1) * g.h;
            float slack = 1e-6f * (fabsf(qo) + far);
            if (c[a] - r > 0) bound = fminf(bound, qo - (float)(c[a] - r) * g.h - slack);
            if (c[a] + r < g.dims[a] - 1) bound = fminf(bound, far - qo - slack);
        }
        bound = fmaxf(bound, 0.f);
        done = (bound == INFINITY) || (bd[K - 1] <= bound * bound);     // INFINITY: the box is the whole grid
        if (done) break;
    }
    if (done) {
#pragma unroll
        for (int j = 0; j < K; j++) {
            out_d2[(size_t)self * K + j] = bd[j];
            if (out_idx) out_idx[(size_t)self * K + j] = (int32_t)bi[j];
        }
    }
    // wave-aggregated append of the open queries
    unsigned long long open = wave_ballot(!done);
    if (open) {
        uint32_t base = 0;
        int leader = __ffsll((long long)open) - 1;
        if ((int)lane_id() == leader) base = atomicAdd(&gp->n_todo, (uint32_t)__popcll(open));
        base = __shfl(base, leader);
        if (!done) todo[base + __popcll(open & ((1ull << lane_id()) - 1ull))] = i;
    }
}

// One block per open query: every thread keeps the K best of its stride of the cloud, then K
// rounds of block-wide arg-min pop the overall best.  16 bytes per (query, point), streamed.
template <int K>
__global__ __launch_bounds__(256) void knn_brute_kernel(uint32_t n, const float4* __restrict__ spts,
                                                        const KnnGrid* __restrict__ gp,
                                                        const uint32_t* __restrict__ todo,
                                                        float* __restrict__ out_d2, int32_t* __restrict__ out_idx) {
    __shared__ unsigned long long cand[4];
    const uint32_t n_todo = gp->n_todo;
    for (uint32_t u = blockIdx.x; u < n_todo; u += gridDim.x) {
        const float4 me = spts[todo[u]];
        const uint32_t self = __float_as_uint(me.w);
        float bd[K];
        uint32_t bi[K];
#pragma unroll
        for (int j = 0; j < K; j++) { bd[j] = INFINITY; bi[j] = 0xffffffffu; }
        for (uint32_t j = threadIdx.x; j < n; j += 256) {
            float4 p = spts[j];
            float dx = p.x - me.x, dy = p.y - me.y, dz = p.z - me.z;
            float d = dx * dx + dy * dy + dz * dz;
            uint32_t id = __float_as_uint(p.w);
            if (id != self) knn_insert<K>(bd, bi, d, id);
        }
        for (int round = 0; round < K; round++) {
            // distances are >= 0 (or NaN, mapped to the end), so their bit patterns order like the values
            uint32_t db = (bd[0] == bd[0]) ? __float_as_uint(bd[0]) : 0x7fffffffu;
            unsigned long long key = ((unsigned long long)db << 32) | threadIdx.x, best = key;
            for (int o = 32; o > 0; o >>= 1) {
                unsigned long long other = __shfl_xor(best, o);
                best = other < best ? other : best;
            }
            if (lane_id() == 0) cand[threadIdx.x >> 6] = best;
            __syncthreads();
            best = cand[0];
            for (int wv = 1; wv < 4; wv++) best = cand[wv] < best ? cand[wv] : best;
            __syncthreads();
            if ((uint32_t)(best & 0xffffffffu) == threadIdx.x) {
                out_d2[(size_t)self * K + round] = bd[0];
                if (out_idx) out_idx[(size_t)self * K + round] = (int32_t)bi[0];
#pragma unroll
                for (int j = 0; j + 1 < K; j++) { bd[j] = bd[j + 1]; bi[j] = bi[j + 1]; }
                bd[K - 1] = INFINITY;
                bi[K - 1] = 0xffffffffu;
            }
        }
    }
}

inline size_t al64(size_t words) { return (words + 63) & ~(size_t)63; }

struct KnnWs {
    uint32_t *stats, *keys, *vals, *start;
    KnnGrid* grid;
    float4* spts;
    void* sort_ws;
    size_t sort_bytes;
};

size_t knn_ws_layout(long long n, uint32_t* base, KnnWs* w) {
    size_t off = 0;
    auto take = [&](size_t words) { size_t o = off; off += al64(words); return base ? base + o : nullptr; };
    uint32_t* stats = take(ST_WORDS);
    uint32_t* grid = take((sizeof(KnnGrid) + 3) / 4);
    uint32_t* keys = take((size_t)n);
    uint32_t* vals = take((size_t)n);
    uint32_t* start = take((size_t)KNN_MAX_CELLS + 1);
    uint32_t* spts = take((size_t)n * 4);
    size_t sort_bytes = mi3dgs_sort_workspace_bytes(n);
    uint32_t* sws = take((sort_bytes + 3) / 4);
    if (w) {
        w->stats = stats; w->grid = reinterpret_cast<KnnGrid*>(grid); w->keys = keys; w->vals = vals;
        w->start = start; w->spts = reinterpret_cast<float4*>(spts); w->sort_ws = sws; w->sort_bytes = sort_bytes;
    }
    return off * 4;
}

// --------------------------------------------------------------------------- images
// Area average: output pixel (x, y) covers source [x sx, (x+1) sx) x [y sy, (y+1) sy) with
// fractional end weights (INTER_AREA's definition), result rounded to nearest, ties to even.
template <typename OutT>
__global__ __launch_bounds__(256) void area_down_kernel(const uint8_t* __restrict__ src, int H, int W, int Cn,
                                                        OutT* __restrict__ dst, int h, int w, float out_scale) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    double sx = (double)W / w, sy = (double)H / h;
    double fx0 = x * sx, fx1 = fmin((x + 1) * sx, (double)W), fy0 = y * sy, fy1 = fmin((y + 1) * sy, (double)H);
    int ix0 = (int)floor(fx0), ix1 = min((int)ceil(fx1), W), iy0 = (int)floor(fy0), iy1 = min((int)ceil(fy1), H);
    float acc[4] = {0, 0, 0, 0};
    for (int yy = iy0; yy < iy1; yy++) {
        float wy = (float)(fmin((double)(yy + 1), fy1) - fmax((double)yy, fy0));
        for (int xx = ix0; xx < ix1; xx++) {
            float wgt = wy * (float)(fmin((double)(xx + 1), fx1) - fmax((double)xx, fx0));
            const uint8_t* p = src + ((size_t)yy * W + xx) * Cn;
            for (int ch = 0; ch < Cn; ch++) acc[ch] += wgt * (float)p[ch];
        }
    }
    float inv = (float)(1.0 / ((fx1 - fx0) * (fy1 - fy0)));
    for (int ch = 0; ch < Cn; ch++) {
        float v = acc[ch] * inv;
        if constexpr (sizeof(OutT) == 1) dst[((size_t)y * w + x) * Cn + ch] = (OutT)fminf(fmaxf(rintf(v), 0.f), 255.f);
        else dst[((size_t)y * w + x) * Cn + ch] = (OutT)(v * out_scale);
    }
}

// Lens undistortion (what nerfstudio's datamanager does with cv2.undistort before training,
// reached through reference main.py:1303-1306; the multi-GPU branch runs COLMAP's undistorter
// instead, main.py:1157-1180).  Output pixel (u, v) of the pinhole camera K_dst looks up the source
// image at distort((u - cx') / fx', (v - cy') / fy') through K_src, bilinear, zero outside (cv2
// remap INTER_LINEAR / BORDER_CONSTANT; pixel index = coordinate, the caller shifts principal
// points as it wishes).  model 0: OpenCV rational + tangential (k1 k2 p1 p2 k3 k4 k5 k6; covers
// COLMAP SIMPLE_RADIAL / RADIAL / OPENCV / FULL_OPENCV), 1: OpenCV fisheye (k1..k4).
struct UndistortArgs {
    float fx, fy, cx, cy;          // source (distorted) camera
    float nfx, nfy, ncx, ncy;      // destination pinhole
    float d[8];
    int model;
};

template <typename OutT>
__global__ __launch_bounds__(256) void undistort_kernel(const uint8_t* __restrict__ src, int H, int W, int Cn,
                                                        OutT* __restrict__ dst, int h, int w, UndistortArgs A,
                                                        float out_scale) {
    int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y;
    if (u >= w || v >= h) return;
    float x = ((float)u - A.ncx) / A.nfx, y = ((float)v - A.ncy) / A.nfy;
    float xd, yd;
    if (A.model == 0) {
        float r2 = x * x + y * y;
        float num = 1.f + r2 * (A.d[0] + r2 * (A.d[1] + r2 * A.d[4]));
        float den = 1.f + r2 * (A.d[5] + r2 * (A.d[6] + r2 * A.d[7]));
        float rad = num / den;
        xd = x * rad + 2.f * A.d[2] * x * y + A.d[3] * (r2 + 2.f * x * x);
        yd = y * rad + A.d[2] * (r2 + 2.f * y * y) + 2.f * A.d[3] * x * y;
    } else {
        float r = sqrtf(x * x + y * y);
        float th = atanf(r), t2 = th * th;
        float thd = th * (1.f + t2 * (A.d[0] + t2 * (A.d[1] + t2 * (A.d[2] + t2 * A.d[3]))));
        float sc = r > 1e-8f ? thd / r : 1.f;
        xd = x * sc; yd = y * sc;
    }
    float us = A.fx * xd + A.cx, vs = A.fy * yd + A.cy;
    float fu = floorf(us), fv = floorf(vs);
    int x0 = (int)fu, y0 = (int)fv;
    float ax = us - fu, ay = vs - fv;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (us > -1.f && vs > -1.f && us < (float)W && vs < (float)H) {
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                int xx = x0 + dx, yy = y0 + dy;
                float wgt = (dx ? ax : 1.f - ax) * (dy ? ay : 1.f - ay);
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    const uint8_t* p = src + ((size_t)yy * W + xx) * Cn;
                    for (int ch = 0; ch < Cn; ch++) acc[ch] += wgt * (float)p[ch];
                }
            }
    }
    for (int ch = 0; ch < Cn; ch++) {
        if constexpr (sizeof(OutT) == 1) dst[((size_t)v * w + u) * Cn + ch] = (OutT)fminf(fmaxf(rintf(acc[ch]), 0.f), 255.f);
        else dst[((size_t)v * w + u) * Cn + ch] = (OutT)(acc[ch] * out_scale);
    }
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ src, size_t n4, size_t n,
                                                        float* __restrict__ dst, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        uint32_t v = reinterpret_cast<const uint32_t*>(src)[i];
        float4 o = make_float4((float)(v & 255u) * scale, (float)((v >> 8) & 255u) * scale,
                               (float)((v >> 16) & 255u) * scale, (float)(v >> 24) * scale);
        reinterpret_cast<float4*>(dst)[i] = o;
    }
    if (i == 0)
        for (size_t j = n4 * 4; j < n; j++) dst[j] = (float)src[j] * scale;
}

}  // namespace

extern "C" size_t mi3dgs_knn_workspace_bytes(long long n) {
    if (n <= 0) return 0;
    return knn_ws_layout(n, nullptr, nullptr);
}

extern "C" int mi3dgs_knn(long long n, const float* points, int k, float* out_d2, int32_t* out_idx_opt,
                          void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(n >= 0 && n < (1ll << 31), "knn: bad n");
    MI_REQUIRE(k >= 1 && k <= 4, "knn: k must be in [1,4]");
    if (n == 0) return 0;
    MI_REQUIRE(points && out_d2, "knn: null pointer");
    MI_REQUIRE(workspace && workspace_bytes >= mi3dgs_knn_workspace_bytes(n), "knn: workspace too small");
    MI_REQUIRE(((uintptr_t)workspace & 15) == 0, "knn: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    KnnWs w;
    knn_ws_layout(n, (uint32_t*)workspace, &w);
    uint32_t nn = (uint32_t)n;
    int blocks = mi_div_up(n, 256);
    MI_LAUNCH("knn_init", knn_init_kernel, dim3(1), dim3(1), 0, st, w.stats);
    for (int round = 0; round < KNN_TRIM_ROUNDS; round++) {
        MI_LAUNCH("knn_stats", knn_stats_kernel, dim3(blocks < 256 ? blocks : 256), dim3(256), 0, st, nn, points,
                  w.stats);
        MI_LAUNCH("knn_grid", knn_grid_kernel, dim3(1), dim3(1), 0, st, nn, w.stats, w.grid,
                  (int)(round == KNN_TRIM_ROUNDS - 1));
    }
    for (int pass = 0; pass < 2; pass++) {
        MI_LAUNCH("knn_cell", knn_cell_kernel, dim3(blocks), dim3(256), 0, st, nn, points, w.grid, w.keys, w.vals);
        MI_LAUNCH_CHECK();
        int rc = mi3dgs_sort_pairs_u32(w.keys, w.vals, n, KNN_CELL_BITS, w.sort_ws, w.sort_bytes, stream);
        if (rc) return rc;
        if (pass == 0) {
            MI_LAUNCH("knn_occupancy", knn_occupancy_kernel, dim3(blocks < 512 ? blocks : 512), dim3(256), 0, st, nn, w.keys, w.grid);
            MI_LAUNCH("knn_refine", knn_refine_kernel, dim3(1), dim3(1), 0, st, nn, w.grid);
        }
    }
    MI_LAUNCH("knn_starts", knn_starts_kernel, dim3(blocks), dim3(256), 0, st, nn, w.keys, w.vals, points, w.grid,
              w.start, w.spts);
    uint32_t* todo = w.keys;                               // the cell keys are dead once the starts exist
#define KNN_RUN(KK)                                                                                              \
    MI_LAUNCH("knn_query", knn_query_kernel<KK>, dim3(blocks), dim3(256), 0, st, nn, w.spts, w.start, w.grid, todo, \
              out_d2, out_idx_opt);                                                                              \
    MI_LAUNCH("knn_brute", knn_brute_kernel<KK>, dim3(KNN_BRUTE_BLOCKS), dim3(256), 0, st, nn, w.spts, w.grid, todo, \
              out_d2, out_idx_opt)
    switch (k) {
        case 1: KNN_RUN(1); break;
        case 2: KNN_RUN(2); break;
        case 3: KNN_RUN(3); break;
        default: KNN_RUN(4); break;
    }
#undef KNN_RUN
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_image_downscale_area(const uint8_t* src, int height, int width, int channels, void* dst,
                                           int out_height, int out_width, int dst_is_f32, void* stream) {
    MI_REQUIRE(src && dst, "image_downscale_area: null pointer");
    MI_REQUIRE(channels >= 1 && channels <= 4, "image_downscale_area: channels must be in [1,4]");
    MI_REQUIRE(height >= 1 && width >= 1 && out_height >= 1 && out_width >= 1, "image_downscale_area: empty image");
    MI_REQUIRE(out_height <= height && out_width <= width, "image_downscale_area: output larger than input");
    MI_REQUIRE(out_height <= 65535, "image_downscale_area: output too tall");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(mi_div_up(out_width, 256), out_height);
    if (dst_is_f32)
        MI_LAUNCH("area_down", area_down_kernel<float>, grid, dim3(256), 0, st, src, height, width, channels,
                  (float*)dst, out_height, out_width, 1.0f / 255.0f);
    else
        MI_LAUNCH("area_down", area_down_kernel<uint8_t>, grid, dim3(256), 0, st, src, height, width, channels,
                  (uint8_t*)dst, out_height, out_width, 1.0f);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_image_u8_to_f32(const uint8_t* src, long long n, float* dst, float scale, void* stream) {
    MI_REQUIRE(n >= 0, "image_u8_to_f32: bad n");
    if (n == 0) return 0;
    MI_REQUIRE(src && dst, "image_u8_to_f32: null pointer");
    MI_REQUIRE(((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 15) == 0, "image_u8_to_f32: src 4-byte / dst 16-byte aligned");
    size_t n4 = (size_t)n / 4;
    size_t threads = n4 > 0 ? n4 : 1;
    MI_LAUNCH("u8_to_f32", u8_to_f32_kernel, dim3(mi_div_up((long long)threads, 256)), dim3(256), 0,
              (hipStream_t)stream, src, n4, (size_t)n, dst, scale);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_image_undistort(const uint8_t* src, int height, int width, int channels, void* dst, int out_height,
                                      int out_width, const float* k_src, const float* k_dst, int model,
                                      const float* dist, int n_dist, int dst_is_f32, void* stream) {
    MI_REQUIRE(src && dst && k_src && k_dst, "image_undistort: null pointer");
    MI_REQUIRE(channels >= 1 && channels <= 4, "image_undistort: channels must be in [1,4]");
    MI_REQUIRE(height >= 1 && width >= 1 && out_height >= 1 && out_width >= 1 && out_height <= 65535,
               "image_undistort: bad image size");
    MI_REQUIRE(model == 0 || model == 1, "image_undistort: model must be 0 (OpenCV) or 1 (OpenCV fisheye)");
    MI_REQUIRE(n_dist >= 0 && n_dist <= (model == 0 ? 8 : 4) && (n_dist == 0 || dist), "image_undistort: bad coefficients");
    MI_REQUIRE(k_src[0] != 0.f && k_src[1] != 0.f && k_dst[0] != 0.f && k_dst[1] != 0.f, "image_undistort: zero focal length");
    UndistortArgs A;
    A.fx = k_src[0]; A.fy = k_src[1]; A.cx = k_src[2]; A.cy = k_src[3];      // HOST arrays: 4 floats each
    A.nfx = k_dst[0]; A.nfy = k_dst[1]; A.ncx = k_dst[2]; A.ncy = k_dst[3];
    for (int i = 0; i < 8; i++) A.d[i] = i < n_dist ? dist[i] : 0.f;
    A.model = model;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(mi_div_up(out_width, 256), out_height);
    if (dst_is_f32)
        MI_LAUNCH("undistort", undistort_kernel<float>, grid, dim3(256), 0, st, src, height, width, channels, (float*)dst,
                  out_height, out_width, A, 1.0f / 255.0f);
    else
        MI_LAUNCH("undistort", undistort_kernel<uint8_t>, grid, dim3(256), 0, st, src, height, width, channels,
                  (uint8_t*)dst, out_height, out_width, A, 1.0f);
    MI_LAUNCH_CHECK();
    return 0;
}
