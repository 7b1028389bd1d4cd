// Shared pieces of the MFMA rasterisers (rasterize_mfma.hip: forward and the reduce-scatter backward;
// rasterize_bwd_mm.hip: the backward with the pixel contraction on the matrix pipe).  gfx950 only.
#pragma once
#include "common.h"

namespace mfma_raster {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;
constexpr int SUB = 32;                       // splats per MFMA sub-batch
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LOG2_MAX_ALPHA = -0.0014434169f;   // log2(0.999)
constexpr float LOG2_ALPHA_THRESHOLD = -7.99435344f;   // log2(1 / 255)

// 64-bit lane masks straight from a compare (no bool round trip), and back: the wave-uniform bookkeeping of the
// chain (who is still compositing, who hits this splat, who stops here) lives in SGPR pairs on the scalar unit.
__device__ __forceinline__ unsigned long long mask_ge(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 3 /* oge */); }
__device__ __forceinline__ unsigned long long mask_gt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 2 /* ogt */); }
__device__ __forceinline__ unsigned long long mask_le(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 5 /* ole */); }
__device__ __forceinline__ unsigned long long mask_ge_i(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 39 /* sge */); }
__device__ __forceinline__ bool lane_of(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

typedef float f16v __attribute__((ext_vector_type(16)));

// ---- the six coefficients of log2 alpha over a tile, shared by both kernels.  contract(off) + explicit
// fma: the two kernels must round identically whatever their surrounding code looks like.
__device__ __forceinline__ void quad_coefs(float x, float y, float A, float B, float C, float opac, float xc, float yc,
                                           float c[6]) {
#pragma clang fp contract(off)
    const float a = (-0.5f * LOG2E) * A, b = (-LOG2E) * B, cc = (-0.5f * LOG2E) * C;
    const float mx = x - xc, my = y - yc;          // splat centre relative to the tile centre; d = (mx - u, my - v)
    c[0] = a;
    c[1] = b;
    c[2] = cc;
    c[3] = -__builtin_fmaf(2.f * a, mx, b * my);
    c[4] = -__builtin_fmaf(2.f * cc, my, b * mx);
    const float tq = __builtin_fmaf(b, my, a * mx);
    c[5] = __builtin_fmaf(mx, tq, (cc * my) * my) + __builtin_amdgcn_logf(opac);     // v_log_f32 = log2
}

// LDS image of one staged batch of 256 splats
// The quadratic form goes through the matrix pipe as bf16 with every coefficient in THREE bf16 terms (hi and mid are exact
// truncations, lo is rounded: 24 significant bits) against basis values that are exact in bf16 (half-integers up to 7.5, their
// products up to 56.25): every product is exact in f32 and the 18 of them are summed in f32, the arithmetic of the f32-input
// MFMA in another order.  Why not v_mfma_f32_32x32x2_f32, which this file used first: an f32-input MFMA occupies the vector
// ALUs for its whole duration (tools/micro/mfma_valu_overlap.hip: one wave alternating a 16x16x4 f32 MFMA with six v_fma
// takes the SUM of both, and a second wave's v_fma gain nothing either), i.e. six of them cost 384 vector cycles per
// sub-batch; four v_mfma_f32_32x32x16_bf16 take 128 cycles of the matrix pipe, part of them beside vector work.
// K slots (18 of 32): c0 {hi mid lo} c1 {..} c2 {hi mid | lo} c3 {..} c4 {..} c5 {hi | mid lo}; lanes 0..31 supply slots 0..7 of
// the first instruction and 16, 17 of the second, lanes 32..63 slots 8..15.
typedef unsigned qu4 __attribute__((ext_vector_type(4)));
typedef __bf16 qbf8 __attribute__((ext_vector_type(8)));
// SLOTS = splats per staged batch: 256 in the forward (one record per thread), 128 in the backward (its LDS has to leave
// room for four blocks per CU: rasterize_bwd_mm.hip)
template <int SLOTS>
struct StagedN {
    qu4 coefA[SLOTS / SUB][2][SUB];      // A operand of the first instruction: [sub-batch][lane half][row] = 8 bf16
    unsigned coefB[SLOTS / SUB][SUB];    // A operand of the second: slots 16, 17 (lanes 0..31; the other 14 slots are zero)
    float4 uni[SLOTS + 2];               // per splat, wave-uniform in the chain: r, g, b (read one or two visits ahead)
};
typedef StagedN<BLOCK> Staged;

// A splat record as the rasterisers use it: fetched one batch AHEAD of its use (the loads of batch b + 1 are issued
// before batch b is composited; a tile's list is walked in batches of 256 and the gather used to sit, fully exposed,
// between two barriers at the head of every batch: 38 - 40 % of both kernels' wave cycles were SQ_WAIT_ANY).
struct RecRegs { float4 a, bb; float cb; };      // x y A B | C o r g | b;  o = 0: padding

__device__ __forceinline__ RecRegs load_rec(const float* __restrict__ splats, int id) {
    RecRegs r;
    r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.bb = r.a; r.cb = 0.f;
    if (id >= 0) {
        const float* rec_f = splats + (size_t)id * SPLAT_STRIDE;
        const float4* rec = reinterpret_cast<const float4*>(rec_f);
        r.a = rec[0]; r.bb = rec[1]; r.cb = rec_f[SP_B];
    }
    return r;
}

// v = hi + mid + lo in bf16: two exact truncations and a rounded remainder
__device__ __forceinline__ void split3(float v, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
#pragma clang fp contract(off)
    const unsigned b0 = __builtin_bit_cast(unsigned, v) & 0xFFFF0000u;
    const float r1 = v - __builtin_bit_cast(float, b0);
    const unsigned b1 = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
    const float r2 = r1 - __builtin_bit_cast(float, b1);
    const __bf16 l = (__bf16)r2;
    hi = (unsigned short)(b0 >> 16);
    mid = (unsigned short)(b1 >> 16);
    lo = __builtin_bit_cast(unsigned short, l);
}

template <typename STAGED>
__device__ __forceinline__ void stage_splat(STAGED& L, int slot, const RecRegs& r, float xc, float yc) {
    float c[6] = {0.f, 0.f, 0.f, 0.f, 0.f, -INFINITY};      // padding: alpha = 2^-inf = 0
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r.bb.y > 0.f) {
        quad_coefs(r.a.x, r.a.y, r.a.z, r.a.w, r.bb.x, r.bb.y, xc, yc, c);
        u = make_float4(r.bb.z, r.bb.w, r.cb, 0.f);
    }
    const int sb = slot >> 5, row = slot & 31;
    unsigned short t[18];
#pragma unroll
    for (int k = 0; k < 6; k++) split3(c[k], t[3 * k], t[3 * k + 1], t[3 * k + 2]);
    if (!(r.bb.y > 0.f)) { t[15] = 0xFF80u; t[16] = 0; t[17] = 0; }        // -inf, not the NaN its residuals would give
    auto pk = [&](int i) { return (unsigned)t[i] | ((unsigned)t[i + 1] << 16); };
    L.coefA[sb][0][row] = qu4{pk(0), pk(2), pk(4), pk(6)};
    L.coefA[sb][1][row] = qu4{pk(8), pk(10), pk(12), pk(14)};
    L.coefB[sb][row] = pk(16);
    L.uni[slot] = u;
}

// per-lane basis operands: for the X block lane l supplies basis_{2q + (l >> 5)} of the pixel that lane (l & 31)
// owns after the swap, for the Y block of the pixel lane 32 + (l & 31) owns.  Basis k: u^2, uv, v^2, u, v, 1.
__device__ __forceinline__ float basis_of(int k, float u, float v) {
    switch (k) {
        case 0: return u * u;
        case 1: return u * v;
        case 2: return v * v;
        case 3: return u;
        case 4: return v;
        default: return 1.f;
    }
}

__device__ __forceinline__ void pixel_of_lane(int wave, int lane, int& lx, int& ly) {
    lx = ((wave & 1) << 3) + (lane & 7);
    ly = ((wave >> 1) << 3) + (lane >> 3);
}

struct Basis { qu4 x1, y1; unsigned x2, y2; };      // B operands of the X and Y column blocks: first instruction, second

// slot -> which basis function it multiplies (see Staged): three slots per coefficient
__device__ __forceinline__ unsigned basis_slot_bits(int slot, float u, float v) {
    if (slot >= 18) return 0u;
    return __builtin_bit_cast(unsigned, basis_of(slot / 3, u, v)) >> 16;          // exact in bf16
}

__device__ __forceinline__ Basis make_basis(int wave, int lane) {
    Basis b;
    const bool h1 = (lane >> 5) != 0;
    int lx, ly;
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        pixel_of_lane(wave, 32 * blk + (lane & 31), lx, ly);
        const float u = (float)lx - 7.5f, v = (float)ly - 7.5f;
        // the six basis values once (bf16 bits, exact), then the words of slots 8 h + 2 e, 8 h + 2 e + 1 (slot s multiplies basis
        // s / 3) for both halves of the wave and ONE select per word -- basis_slot_bits with a run-time slot was a six-way select
        // chain per half word, twenty of them per call, at the head of every forward and backward block
        unsigned k[6];
#pragma unroll
        for (int i = 0; i < 6; i++) k[i] = basis_slot_bits(3 * i, u, v);
        // h = 0: slots (0,1) (2,3) (4,5) (6,7) -> bases (0,0) (0,1) (1,1) (2,2); h = 1: slots (8,9) .. (14,15) -> (2,3) (3,3) (4,4) (4,5)
        const unsigned w0[4] = {k[0] | (k[0] << 16), k[0] | (k[1] << 16), k[1] | (k[1] << 16), k[2] | (k[2] << 16)};
        const unsigned w1[4] = {k[2] | (k[3] << 16), k[3] | (k[3] << 16), k[4] | (k[4] << 16), k[4] | (k[5] << 16)};
        qu4 w;
#pragma unroll
        for (int e = 0; e < 4; e++) w[e] = h1 ? w1[e] : w0[e];
        const unsigned w2 = h1 ? 0u : (k[5] | (k[5] << 16));           // slots 16, 17
        if (blk == 0) { b.x1 = w; b.x2 = w2; } else { b.y1 = w; b.y2 = w2; }
    }
    return b;
}

// log2 alpha (opacity folded in, NOT yet clamped) of this lane's pixel against the 32 splats of sub-batch
// `sb`: s[i] for row i in depth order.
template <typename STAGED>
__device__ __forceinline__ void eval_sub_batch(const STAGED& L, int sb, int lane, const Basis& b, float s[SUB]) {
    const int h = lane >> 5, row = lane & 31;
    const qu4 a1 = L.coefA[sb][h][row];
    const unsigned a2w = L.coefB[sb][row];
    const qu4 a2 = {h == 0 ? a2w : 0u, 0u, 0u, 0u};
    const qu4 bx2 = {b.x2, 0u, 0u, 0u}, by2 = {b.y2, 0u, 0u, 0u};
    f16v X = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16v Y = X;
    X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(qbf8, a1), __builtin_bit_cast(qbf8, b.x1), X, 0, 0, 0);
    Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(qbf8, a1), __builtin_bit_cast(qbf8, b.y1), Y, 0, 0, 0);
    X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(qbf8, a2), __builtin_bit_cast(qbf8, bx2), X, 0, 0, 0);
    Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(qbf8, a2), __builtin_bit_cast(qbf8, by2), Y, 0, 0, 0);
    // register r of a 32x32 accumulator = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  swap(X[r], Y[r]) hands X's
    // upper-half rows to the lower lanes and Y's lower-half rows to the upper lanes: afterwards, on every lane,
    // X[r] = row (r & 3) + 8 (r >> 2) and Y[r] = that + 4 of the lane's OWN pixel.
#pragma unroll
    for (int r = 0; r < 16; r++) {
        // (copy the elements out first: __builtin_bit_cast applied directly to `X[r]` reads element 0 for every r
        // with this clang)
        const float xv = X[r], yv = Y[r];
        auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xv), __builtin_bit_cast(unsigned, yv),
                                                   false, false);
        const unsigned x_new = sw[0], y_new = sw[1];
        const float xr = __builtin_bit_cast(float, x_new), yr = __builtin_bit_cast(float, y_new);
        const int row0 = (r & 3) + 8 * (r >> 2);
        s[row0] = xr;
        s[row0 + 4] = yr;
    }
}

// The chain reads one wave-uniform float4 per visit (the splat's colour).  hipcc treats the LDS address as a
// scalar and re-materialises it into a VGPR for every read (a v_mov per visit in VALU-issue-bound loops); an
// opaque per-sub-batch VGPR base keeps the row index in the instruction's immediate offset instead.
struct Rgb { float x, y, z; };
struct alignas(16) F4pod { float x, y, z, w; };
typedef __attribute__((address_space(3))) const F4pod* lds_f4_ptr;
__device__ __forceinline__ lds_f4_ptr opaque_lds_base(const float4* p) {
    lds_f4_ptr q = (lds_f4_ptr)reinterpret_cast<const F4pod*>(p);
    asm volatile("" : "+v"(q));
    return q;
}
__device__ __forceinline__ Rgb lds_rgb(lds_f4_ptr p, int i) { return Rgb{p[i].x, p[i].y, p[i].z}; }

// alpha of a pair from its log2: min(0.999, o vis).  (sigma is not clamped at 0: a PSD form evaluated through the
// six-term chain can come out up to ~3e-4 positive in log2 units at the splat centre, i.e. alpha up to 1.0002 o.)
// The membership test alpha >= 1/255 is taken in the log domain, on the MFMA result itself: a visit that no lane
// hits costs ONE vector instruction, and forward and backward can not disagree.
__device__ __forceinline__ float alpha_of(float s) {
    return __builtin_amdgcn_exp2f(__builtin_fminf(s, LOG2_MAX_ALPHA));
}


// acc row of a staged slot: moments of q = -dL/dsigma about the tile centre, then the colour sums
constexpr int AC_QU = 0, AC_QV = 1, AC_QUU = 2, AC_QUV = 3, AC_QVV = 4, AC_Q = 5, AC_R = 6, AC_G = 7, AC_B = 8,
              AC_ABSX = 9, AC_ABSY = 10, AC_STRIDE = 12;

// Block -> tile: tile = block.  Workgroups go to the eight XCDs round-robin by workgroup id and every XCD has its own L2, so XCD k
// owns the tile COLUMNS x = k mod 8 (120 tiles per row at 1080p: t and t + 120 meet in one L2 and are resident together) and
// every XCD gets the same mix of heavy and light tiles.  Two other mappings were measured in round 2 and removed in round 4
// (docs/FINDINGS_r01_r02.md): one contiguous eighth of the image per XCD (rasterize_fwd 129 -> 204 us, rasterize_bwd 487 -> 750:
// the heavy middle of the picture lands on three or four XCDs) and clusters of 4 x 2 tiles per XCD (slower too).  Balance
// across XCDs beats locality within one.

struct PixelBasis { float u, v, uu, uv, vv; };

// ---- backward in segments.  One block per tile walks its list serially, so a launch is as long as its heaviest tile: on the
// reference's wolf.spz (100 k Gaussians, 960 x 720) a few hundred tiles hold thousands of splats each, 84 % of them reached, and
// rasterize_bwd was 37 % of a real training step.  The forward therefore leaves its per-pixel state (T, r, g, b) at every
// seg-entry boundary a tile's block walks past (one work item + one 4 KB checkpoint per boundary, handed out by an atomic
// counter), and the backward processes the entries in front of each boundary as a tile of their own: transmittance from the
// checkpoint, "colour behind" = (final colour - checkpoint colour) . v_rgb.  The tile's own block keeps the entries behind its
// last boundary.  Tiles that stop before the first boundary leave nothing and cost one branch per batch.
// Segment length: 256 entries (the forward's batch).  Shorter segments balance better (S1 146 -> 118 us, wolf 1280 x 720
// 199 -> 153 at 256 against 143 / 174 at 512), but every item has a fixed cost, which shows where lists are long and evenly long,
// i.e. where nothing needed balancing: S2 pays 20 us for 2 373 items at 256, and even its 577 items at 512 only cost (forward
// +4 us, backward +4 us).  The capacity per tile is the host-side proxy for that regime: above 1 024 entries per tile, on a grid
// of at least 4 096 tiles, both rasterisers ignore the workspace (seg_ws_in_use) and run as if none had been given.
constexpr int SEG_MIN = 256;              // = BLOCK: the forward can only stop at its batch boundaries
// control words (the first bytes of the workspace)
constexpr int SEG_CTL_ITEMS = 0;      // work items the forward booked
constexpr int SEG_CTL_OUT = 1;        // backward: dedicated workers that have read SEG_CTL_ITEMS
constexpr int SEG_CTL_HEAVY = 2;      // forward in segments: tiles walked as segments
struct SegWs {
    uint32_t* ctl;        // SEG_CTL_*
    uint32_t* tile_skip;  // [n_tiles] entries at the head of each tile's list that belong to work items (its own block starts behind them)
    uint32_t* heavy;      // [n_tiles] forward in segments: the tiles whose lists are walked as segments
    uint2* tile_items;    // [n_tiles] forward in segments: {first item, segments} of a heavy tile
    uint4* work;          // [cap] {tile, first list entry of the segment, checkpoint slot of its END boundary, entries in the segment}
                          //       (entries = 0: a segment of the FORWARD whose end boundary no pixel walked past -- no backward item)
    float4* ckpt;         // [cap][256] T, r, g, b per pixel (thread order of the forward block) at the boundary
    int32_t* local_last;  // [cap][256] forward in segments: last contributor inside the segment (0x3fffffff: none) | bit 30: stopped inside
    uint32_t cap;
    uint32_t seg;         // entries per segment of this call (forward only)
};
constexpr size_t SEG_ITEM_BYTES = 16 + BLOCK * 16 + BLOCK * 4;
inline size_t seg_ws_fixed(int n_tiles) {
    const size_t a4 = ((size_t)n_tiles * 4 + 255) & ~(size_t)255, a8 = ((size_t)n_tiles * 8 + 255) & ~(size_t)255;
    return 256 + 2 * a4 + a8 + 512;
}
inline size_t seg_ws_cap(int n_tiles, size_t bytes) {
    const size_t fixed = seg_ws_fixed(n_tiles);
    return bytes < fixed + SEG_ITEM_BYTES ? 0 : (bytes - fixed) / SEG_ITEM_BYTES;
}
inline size_t seg_ws_bytes_for(int n_tiles, long long max_isect) {
    // a backward item has >= SEG_MIN entries of its tile in front of its boundary; a heavy tile's last forward segment may be short
    const size_t cap = (size_t)(max_isect / SEG_MIN) + (size_t)n_tiles + 16;
    return seg_ws_fixed(n_tiles) + cap * SEG_ITEM_BYTES;
}
// The workspace is not used where lists are long everywhere AND there are tiles enough to fill the device many times over
// (S2, S3: nothing to balance).  Few tiles with long lists are the opposite case: the first 3 000 steps of an ns-train run render
// at a quarter of the resolution -- 180 tiles for 256 CUs, thousands of entries each -- and the serial walk took 253 us there.
constexpr size_t SEG_OFF_ENTRIES_PER_TILE = 1024;
constexpr int SEG_OFF_MIN_TILES = 4096;
inline bool seg_ws_in_use(int n_tiles, size_t bytes) {
    const size_t cap = seg_ws_cap(n_tiles, bytes);
    if (cap == 0) return false;
    const size_t nt = (size_t)(n_tiles > 0 ? n_tiles : 1);
    const size_t est_isect = cap > nt + 16 ? (cap - nt - 16) * SEG_MIN : 0;
    const bool long_everywhere = est_isect / nt > SEG_OFF_ENTRIES_PER_TILE;
    return !(long_everywhere && n_tiles >= SEG_OFF_MIN_TILES);
}
// both rasterisers derive the same views from (n_tiles, bytes)
inline bool seg_ws_layout(int n_tiles, void* base, size_t bytes, SegWs* out) {
    const size_t cap = seg_ws_cap(n_tiles, bytes);
    if (cap == 0) return false;
    const size_t a4 = ((size_t)n_tiles * 4 + 255) & ~(size_t)255, a8 = ((size_t)n_tiles * 8 + 255) & ~(size_t)255;
    char* b = (char*)base;
    out->ctl = (uint32_t*)b;
    out->tile_skip = (uint32_t*)(b + 256);
    out->heavy = (uint32_t*)(b + 256 + a4);
    out->tile_items = (uint2*)(b + 256 + 2 * a4);
    char* w = b + 256 + 2 * a4 + a8;
    out->work = (uint4*)w;
    char* c = w + ((cap * 16 + 255) & ~(size_t)255);
    out->ckpt = (float4*)c;
    out->local_last = (int32_t*)(c + cap * BLOCK * 16);
    out->cap = (uint32_t)cap;
    out->seg = SEG_MIN;
    return true;
}

}  // namespace mfma_raster
