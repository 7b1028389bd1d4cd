// Backward tile rasteriser with the per-splat pixel sums contracted on the matrix pipe (bf16 MFMA, two-term split).
// gfx950 only.  Replaces gsplat rasterize_to_pixels_bwd, reached by the reference only through
// main.py:1312 / main.py:1343.  Built with -fno-slp-vectorize (see the Makefile).
#include "rasterize_mfma.h"
#include <stdlib.h>

namespace mfma_raster {

// ---------------------------------------------------------------------------------------- backward, contraction on MFMA
// The per-splat sums over the pixels of a quadrant ARE a matrix product,
//     [Q | W] (splats x 64 pixels)  .  [u v u^2 uv v^2 1 | v_r v_g v_b] (64 pixels x 9),
// with Q = q(s, p) = -dL/dsigma and W = alpha T (the colour weight): moments in columns 0..5, colour sums in 6..8.
// The chain produces Q and W with lane = pixel, the A operand of an MFMA wants lane = row (splat) with K (pixels)
// inside the lane, so a wave transposes through a private LDS region: each visit stores its two values as rows
// (one 32-bit word per pixel), after CH splats the wave reads the 16 rows back as A operands (4 x ds_read_b128 per lane).
//
// Which MFMA.  The f32-input forms do NOT run beside the vector pipe: tools/micro/mfma_valu_overlap.hip, one wave of
// v_mfma_f32_16x16x4_f32 and one wave of v_fma_f32 on the same SIMD take the SUM of their times (5 929 us against 4 372 +
// 1 797), and the f32 contraction (64 MFMAs = 2 048 cycles per 32 visits) made this kernel slower, 500 -> 750 us.  The bf16
// forms do overlap and are 16 x cheaper, so the values go through the matrix pipe as TWO bf16 terms each: a word holds
// hi = the top 16 bits of the f32 (an exact split) and lo = bf16(value - hi) (round to nearest), and the K index of the
// MFMA is (pixel, part): both parts of a pixel multiply the same basis value, so sum_k A[k] B[k] = sum_p (hi_p + lo_p) b(p)
// with no unpacking.  16 significant bits per term (relative error <= 2^-16, unbiased), f32 accumulation.  The basis values
// (half-integers up to 7.5, their products up to 56.25) are exact in bf16; v_rgb gets the same two-term split as two column
// groups (6..8 hi, 9..11 lo) whose sums land on the same three slots.  4 x v_mfma_f32_16x16x32_bf16 (64 matrix-pipe cycles)
// per chunk replace the five moment multiplies and the 24-instruction cross-lane reduce-scatter of every live visit.
// With ABSGRAD the two |.| sums ride along as two more rows per splat (chunks of 4).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int TR_STRIDE = 68;
constexpr int TR_ROWS = 16;
constexpr int GRP = 64;                   // slots per flush group: the per-wave sums of 64 splats wait in LDS, then the block flushes them
// ---- two shapes of the kernel, chosen per launch by the number of Gaussians (mi_rasterize_bwd_mm):
//   STG        splats per staged batch: 256 (one per thread) or 128 (LDS 50 -> 40 KB: room for four blocks per CU)
//   PIPELINE   true: a chunk's contraction is issued one MFMA per visit among the NEXT chunk's visits (16 + 4 more live
//              registers); false: operands are read and the four MFMAs issued when the chunk ends
//   COL_AHEAD  the wave-uniform colour read runs this many visits ahead (3 registers each)
//   WAVES      waves per SIMD the register allocation is held to (3: 168 VGPRs; 4: 126)
// DEEP = the round-2 kernel: fastest per wave, three waves per SIMD.  WIDE: 126 registers, 40 KB, four waves per SIMD.
// Same-box A/B (profiles/r03_raster_bwd_shape_ab.txt, tools/raster_ab.py): S2 (2 M Gaussians, every tile a few hundred
// reached splats) 459 -> 398 us with WIDE; S1 148 -> 144; the reference's wolf.spz at 960 x 720 (100 k Gaussians, the time is the
// serial walk of a few hundred heavy tiles) 201 -> 205, with absgrad 271 -> 288: there a wave's own speed counts, not how many
// waves wait beside it.  WIDE by the compiler's spiller instead (pipelining kept, 39 registers in scratch) LOST 8 - 15 %
// (r03_raster_bwd_occupancy_ab.txt); WIDE's ingredients at three waves (no pipelining, 256-slot batches) lose 3 - 5 %.
struct ShapeDeep { static constexpr int STG = 256, COL_AHEAD = 2, WAVES = 3; static constexpr bool PIPELINE = true; };
struct ShapeWide { static constexpr int STG = 128, COL_AHEAD = 1, WAVES = 4; static constexpr bool PIPELINE = false; };
// columns of a per-wave sum row: the nine of the reduce-scatter kernel, then the lo part of the colour sums (9..11; the flush adds
// them), then |x|, |y|.  MERGE (ABSGRAD at four waves per SIMD, where the LDS has to stay under 40 KB): the lo colour sums
// are added to the hi ones before they are stored (three lanes of the accumulator, one cross-lane read each) and |x|, |y| take
// columns 9, 10: eleven columns instead of fourteen.
template <typename SH> struct AbsCols {
    static constexpr bool MERGE = SH::WAVES >= 4;
    static constexpr int ABSX = MERGE ? 9 : 12, ABSY = MERGE ? 10 : 13;
};

// LDS float atomics are lane-serial on this part (tools/micro/lds_ops.hip: ds_add_f32 takes ~3 LDS cycles per ACTIVE lane,
// 55 for the 18 lanes that would add a chunk's sums, against 3 for a plain ds_write_b32), so every wave keeps its own sums
// (plain stores, each (wave, slot, column) written at most once per group) and the flush adds the four quadrants.
template <bool ABSGRAD, typename SH>
struct StagedBwdMM {
    static constexpr int STG = SH::STG;
    static constexpr int AW = ABSGRAD ? (AbsCols<SH>::MERGE ? 11 : 14) : 12;
    StagedN<STG> f;
    float4 geo[STG];          // mx, my (relative to the tile centre), A, B
    float2 geo2[STG];         // C, 1 / o
    int id[STG];
    float accw[4][GRP][AW];
    alignas(16) unsigned tr[4][TR_ROWS][TR_STRIDE];      // per wave: rows = (value, splat of the chunk), columns = the wave's 64 pixels
    unsigned long long gmask[4];        // per wave: slots of the current group whose visit was live (its sums are meaningful)
    int wave_max[4];
};

// what a lane does with the accumulator of a chunk: D row 4 (lane >> 4) + r, column lane & 15
struct MMLane {
    u4v bop[4];               // B operand of MFMA m: column (lane & 15) at the pixels 16 m + 4 (lane >> 4) + 0..3, both parts
    int acc_off;              // float offset of this lane's (first row, column) inside a chunk of accw, or -1: nothing to store
};

__device__ __forceinline__ unsigned f32_hi(float v) { return __builtin_bit_cast(unsigned, v) & 0xFFFF0000u; }

// two values -> two words (hi | lo)
__device__ __forceinline__ void split2(float a, float b, unsigned& wa, unsigned& wb) {
    const unsigned ha = f32_hi(a), hb = f32_hi(b);
    const f2v r = {a - __builtin_bit_cast(float, ha), b - __builtin_bit_cast(float, hb)};
    const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf2v));      // v_cvt_pk_bf16_f32
    wa = ha | (l & 0xFFFFu);
    wb = hb | (l >> 16);
}

// three terms (experiments build, A/B of VERDICT r2 #3): hi and mid are exact truncations, lo is rounded -- 24 significant bits.
// Word A = hi | mid, word B = lo | 0; both multiply the same basis, so the two words are two more ROWS of the same product
// whose results add up.
__device__ __forceinline__ void split3w(float a, unsigned& wA, unsigned& wB) {
#pragma clang fp contract(off)
    const unsigned h = f32_hi(a);
    const float r1 = a - __builtin_bit_cast(float, h);
    const unsigned m = f32_hi(r1);
    const float r2 = r1 - __builtin_bit_cast(float, m);
    const f2v r = {r2, 0.f};
    const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf2v)) & 0xFFFFu;
    wA = h | (m >> 16);
    wB = l;
}

// The contraction of a chunk is software-pipelined against the visits of the NEXT chunk: its A operands are read when its last
// visit has stored (LDS executes a wave's instructions in order, so the next chunk's stores to the same rows stay behind these
// reads), its 4 MFMAs are issued one per visit between the vector instructions of the following visits, and its
// sums are stored when that chunk ends.  At a sub-batch boundary a pending chunk has its operands and none of its MFMAs.
struct MMPend {
    u4v a[4];
    f4v d;
    int slot0;                // first slot of the chunk inside its group
    int slot_abs;             // ... inside the staged batch (wave-flush variant)
    unsigned live;            // ... which of its visits were live (wave-flush variant)
    bool on;                  // wave-uniform
};

__device__ __forceinline__ void mm_issue(MMPend& P, const MMLane& mm, int m) {
    P.d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, P.a[m]), __builtin_bit_cast(bf8v, mm.bop[m]), P.d, 0, 0, 0);
}

// WF (variant, MI3DGS_RASTER_MODE=14): the wave turns its own chunk's sums into gradient records and adds them to global
// memory itself -- no per-wave sums kept for a block flush, no group barriers, but one 64-byte float-atomic request per
// (quadrant, splat) instead of one per (tile, splat).
template <bool ABSGRAD, bool WF, bool T3, typename SH>
__device__ __forceinline__ void mm_finish(StagedBwdMM<ABSGRAD, SH>& L, MMPend& P, const MMLane& mm, int wv, int lane,
                                          float* __restrict__ v_splats) {
    constexpr int AW = StagedBwdMM<ABSGRAD, SH>::AW;
    constexpr int CH = (ABSGRAD || T3) ? 4 : 8;
    if (T3) {          // rows 8..15 (lanes 32..63) hold the sums of the third term: add them to rows 0..7
#pragma unroll
        for (int r = 0; r < 4; r++) P.d[r] += __shfl_down(P.d[r], 32, 64);
    }
    if (ABSGRAD && AbsCols<SH>::MERGE) {     // colour sums: column j + 3 (from the lo part of v_rgb) onto column j = 6..8 (row_shl:3 inside the 16-lane row)
        const bool col = (lane & 15) >= 6 && (lane & 15) < 9;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float up = __shfl_down(P.d[r], 3, 64);          // lane j + 3 of the same 16-lane row (j <= 8)
            P.d[r] += col ? up : 0.f;
        }
    }
    if (mm.acc_off >= 0) {
        float* a = &L.accw[wv][WF ? 0 : P.slot0][0] + mm.acc_off;
        a[0] = P.d[0]; a[AW] = P.d[1]; a[2 * AW] = P.d[2]; a[3 * AW] = P.d[3];
    }
    P.on = false;
    if (WF) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // same-wave LDS hand-off
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < CH / 4; q++) {
            const int sc = 4 * q + (lane >> 4), comp = lane & 15;
            const bool lv = (P.live >> sc) & 1u;
            if (lv && comp < (ABSGRAD ? GR_DEPTH : GR_ABSX)) {
                const float* ac = L.accw[wv][sc];
                const int slot = P.slot_abs + sc;
                const float M = ac[AC_Q], Mu = ac[AC_QU], Mv = ac[AC_QV];
                const float4 ge = L.geo[slot];
                const float2 g2 = L.geo2[slot];
                const float mx = ge.x, my = ge.y;
                const float sdx = mx * M - Mu, sdy = my * M - Mv;
                float val;
                switch (comp) {
                    case GR_X: val = -(ge.z * sdx + ge.w * sdy); break;
                    case GR_Y: val = -(ge.w * sdx + g2.x * sdy); break;
                    case GR_CA: val = -0.5f * (mx * (mx * M - 2.f * Mu) + ac[AC_QUU]); break;
                    case GR_CB: val = -(mx * (my * M - Mv) - my * Mu + ac[AC_QUV]); break;
                    case GR_CC: val = -0.5f * (my * (my * M - 2.f * Mv) + ac[AC_QVV]); break;
                    case GR_OPA: val = M * g2.y; break;
                    case GR_R: case GR_G: case GR_B: val = (ABSGRAD && AbsCols<SH>::MERGE) ? ac[comp] : ac[comp] + ac[comp + 3]; break;
                    case GR_ABSX: val = ac[AbsCols<SH>::ABSX < AW ? AbsCols<SH>::ABSX : 0]; break;
                    default: val = ac[AbsCols<SH>::ABSY < AW ? AbsCols<SH>::ABSY : 0]; break;
                }
                atomicAdd(&v_splats[(size_t)L.id[slot] * GRAD_STRIDE + comp], val);
            }
        }
    }
}

template <bool ABSGRAD, typename SH>
__device__ __forceinline__ void mm_read(StagedBwdMM<ABSGRAD, SH>& L, MMPend& P, int wv, int lane, int slot0) {
    asm volatile("" ::: "memory");      // the rows were written by other lanes of this wave: LDS is in order per wave
    const u4v* ap = reinterpret_cast<const u4v*>(&L.tr[wv][lane & 15][4 * (lane >> 4)]);
    P.a[0] = ap[0]; P.a[1] = ap[4]; P.a[2] = ap[8]; P.a[3] = ap[12];
    asm volatile("" ::: "memory");      // the next chunk's stores stay behind these reads
    P.d = f4v{0.f, 0.f, 0.f, 0.f};
    P.slot0 = slot0;
    P.on = true;
}

template <bool ABSGRAD, bool WF, bool T3, typename SH>
__device__ __forceinline__ void mm_drain(StagedBwdMM<ABSGRAD, SH>& L, MMPend& P, const MMLane& mm, int wv, int lane,
                                         float* __restrict__ v_splats) {
    if (!P.on) return;
#pragma unroll
    for (int m = 0; m < 4; m++) mm_issue(P, mm, m);
    mm_finish<ABSGRAD, WF, T3, SH>(L, P, mm, wv, lane, v_splats);
}

// One sub-batch of the backward walk, rows i = 0..31 <-> sorted indices be - 32 sb - i (back to front).
// FAST: every pixel of the wave that composited anything is already in range (index <= its last contributor).
// gbit0: bit of this sub-batch's first chunk in the group's mask.
template <bool ABSGRAD, bool FAST, int EXP, typename SH>
__device__ __forceinline__ void bwd_sub_batch_mm(StagedBwdMM<ABSGRAD, SH>& L, const float (&s)[SUB], int sb, int be, int lane, int wv,
                                                 int bin_final, unsigned long long has, const PixelBasis& px, const MMLane& mm,
                                                 MMPend& P, unsigned long long& gmask, const float (&vrgb)[3], float tail,
                                                 float& T, float& bufdot, float* __restrict__ v_splats) {
    constexpr bool WF = (EXP & 4) != 0;
    constexpr bool T3 = (EXP & 8) != 0 && !ABSGRAD;      // three-term transport (experiments build)
    constexpr int CH = (ABSGRAD || T3) ? 4 : 8;  // splats per chunk: CH x (2 or 4 values) = 16 rows
    const int gs0 = (sb * SUB) & (GRP - 1);      // slot of row 0 inside its group
    const lds_f4_ptr uni = opaque_lds_base(&L.f.uni[sb * SUB]);
    // colours two visits ahead: LDS serves a wave in order, so a read queues behind the two stores of the visit before it
    Rgb col_next = lds_rgb(uni, 0), col_next2 = lds_rgb(uni, SH::COL_AHEAD == 2 ? 1 : 0);
    unsigned* trw = &L.tr[wv][0][lane];
    // This loop is bound by the instructions ONE wave can issue (about one per four cycles, whatever their kind), so scalar
    // bookkeeping counts like vector work.  A dead visit is a compare and a branch: its rows keep whatever they held, the
    // contraction is row-wise (garbage in row r only reaches sums of row r), and the flush takes a (wave, slot) sum only if
    // the visit was live (`gmask`, one bit per slot of the group).
    unsigned live = 0u;                          // bits of this sub-batch
#pragma unroll
    for (int i = 0; i < SUB; i++) {
        const int k = sb * SUB + i;
        const int ci = i % CH;
        const Rgb col = (EXP & 2) ? Rgb{0.3f, 0.4f, 0.5f} : col_next;
        if (!(EXP & 2)) {
            if (SH::COL_AHEAD == 2) { col_next = col_next2; col_next2 = lds_rgb(uni, i + 2); }
            else col_next = lds_rgb(uni, i + 1);
        }
        if (SH::PIPELINE && ci < 4) mm_issue(P, mm, ci);         // of the chunk before (results unused if there was none)
        // the forward's own membership test (same MFMA result, same compare), for the splats this pixel reached
        unsigned long long valid = mask_ge(s[i], LOG2_ALPHA_THRESHOLD) & has;
        if (!FAST) valid &= mask_ge_i(bin_final, be - k);
        if (valid != 0ull) {
            live |= 1u << i;
            const float alpha = alpha_of(s[i]);
            // branch-free live part: a lane that does not take part runs it with alpha = 0 (ra = 1, T and bufdot
            // unchanged bit for bit, both rows 0)
            const float a_eff = lane_of(valid) ? alpha : 0.f;
            const float ra = __builtin_amdgcn_rcpf(1.f - a_eff);
            T *= ra;
            const float fac = a_eff * T;
            const float cv = col.x * vrgb[0] + col.y * vrgb[1] + col.z * vrgb[2];            // c . v_rgb
            const float v_alpha = T * cv - ra * (bufdot - tail);
            bufdot = __builtin_fmaf(cv, fac, bufdot);
            // q = o vis dL/dalpha = -dL/dsigma; zero where the 0.999 clamp is active (alpha == o vis otherwise)
            const unsigned long long gon = valid & mask_le(s[i], LOG2_MAX_ALPHA);
            const float q = lane_of(gon) ? alpha * v_alpha : 0.f;
            unsigned wq, wf;
            if (T3) {
                unsigned wq2, wf2;
                split3w(q, wq, wq2);
                split3w(fac, wf, wf2);
                trw[(2 * CH + ci) * TR_STRIDE] = wq2;
                trw[(3 * CH + ci) * TR_STRIDE] = wf2;
            } else {
                split2(q, fac, wq, wf);
            }
            trw[ci * TR_STRIDE] = wq;
            trw[(CH + ci) * TR_STRIDE] = wf;
            if (ABSGRAD) {
                const float4 ge = L.geo[k];
                const float cC = L.geo2[k].x;
                const float dx = ge.x - px.u, dy = ge.y - px.v;
                unsigned wx, wy;
                split2(fabsf(q * (ge.z * dx + ge.w * dy)), fabsf(q * (ge.w * dx + cC * dy)), wx, wy);
                trw[(2 * CH + ci) * TR_STRIDE] = wx;
                trw[(3 * CH + ci) * TR_STRIDE] = wy;
            }
        }
        if (ci == CH - 1) {
            if (SH::PIPELINE && P.on) mm_finish<ABSGRAD, WF, T3, SH>(L, P, mm, wv, lane, v_splats);
            const unsigned cl = (live >> (i - (CH - 1))) & ((1u << CH) - 1u);
            if (cl) {
                mm_read<ABSGRAD, SH>(L, P, wv, lane, gs0 + i - (CH - 1));
                P.slot_abs = k - (CH - 1);
                P.live = cl;
                if (!SH::PIPELINE) mm_drain<ABSGRAD, WF, T3, SH>(L, P, mm, wv, lane, v_splats);      // read, four MFMAs, store: nothing stays live
            }
        }
    }
    gmask |= (unsigned long long)live << gs0;
}

#ifdef MI3DGS_OS_STAMPS
// Probe build only (tools/raster_probe.py): start / end wall-clock stamps (100 MHz) and list length of every tile's block.
__device__ unsigned long long g_rb_stamps[16384][8];      // start, end, walked entries; prologue done, first records staged, first group walked, pixel values in and reduced
#define RB_STAMP(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_rb_stamps[blockIdx.x][i] = (v); } while (0)
}  // namespace mfma_raster
extern "C" int mi3dgs_debug_read_rb_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mfma_raster::g_rb_stamps), bytes < sizeof(mfma_raster::g_rb_stamps) ? bytes : sizeof(mfma_raster::g_rb_stamps));
}
namespace mfma_raster {
#else
#define RB_STAMP(i, v) do { } while (0)
#endif

// One walk of rasterize_bwd: the entries [lo, ...] of tile t's list, back to front, for the block's 256 pixels.
//   item < 0: the tile's own block -- everything behind the last boundary the forward left (the whole list if it left none);
//   item >= 0: segment `item` of the work list -- the entries in front of a boundary, state from the forward's checkpoint.
template <bool HAS_BG, bool ABSGRAD, int EXP, typename SH>
__device__ __forceinline__ void bwd_walk(StagedBwdMM<ABSGRAD, SH>& L, int t, int item, const SegWs& seg, const float* __restrict__ render,
    int W, int H, int tw, int th, const float* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
    const float* __restrict__ backgrounds, const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render, const float* __restrict__ v_alphas, float* __restrict__ v_splats) {
    constexpr int AW = StagedBwdMM<ABSGRAD, SH>::AW;
    constexpr int STG = SH::STG;
    const bool worker = item >= 0;
    int seg_lo = 0, seg_len = 0;
    uint32_t slot_ck = 0;
    if (worker) {
        const uint4 wk = seg.work[item];
        t = (int)wk.x; seg_lo = (int)wk.y; slot_ck = wk.z; seg_len = (int)wk.w;
        if (seg_len == 0) return;          // a forward segment whose end boundary no pixel walked past: nothing for the backward
    }
    const int cam = t / (tw * th);
    const int tile_in = t - cam * (tw * th);
    const int ty = tile_in / tw, tx = tile_in - ty * tw;
    int lane = lane_id(), wv = threadIdx.x >> 6;  
    // a segment worker runs this walk in a loop: what depends on the thread alone must be recomputed per walk, not kept in
    // registers across it (hoisted, those values cost 33 - 38 spilled VGPRs)
    asm volatile("" : "+v"(lane), "+v"(wv));
    const int tid = wv * 64 + lane;
    int lx, ly;
    pixel_of_lane(wv, lane, lx, ly);
    const int px_i = tx * TILE + lx, py_i = ty * TILE + ly;
    const bool inside = px_i < W && py_i < H;
    const float xc = (float)(tx * TILE) + 8.f, yc = (float)(ty * TILE) + 8.f;
    PixelBasis px;
    px.u = (float)lx - 7.5f; px.v = (float)ly - 7.5f;
    px.uu = px.u * px.u; px.uv = px.u * px.v; px.vv = px.v * px.v;
    int start, end;
    if (worker) {
        start = seg_lo; end = seg_lo + seg_len;
    } else {
        start = tile_offsets[t];
        end = (t + 1 < n_tiles_total) ? tile_offsets[t + 1] : *n_isect_ptr;
        if (seg.ckpt) start += (int)seg.tile_skip[t];
    }
    RB_STAMP(0, wall_clock64()); RB_STAMP(1, 0ull); RB_STAMP(2, 0ull);
    // the pixel's values do not depend on the tile's range: requested before the range is looked at, so that they travel
    // beside the two offsets instead of behind them (a block's life starts with four dependent round trips otherwise:
    // offsets -> pixels -> list -> records; measured fixed cost per block ~15 us of a median 41 us at three blocks per CU)
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
    if (end <= start) return;
    const float vrgb[3] = {vr0, vr1, vr2};
    float tail = T_final * va;                      // T_final (v_alpha - bg . v_rgb)
    float bgdot = 0.f;
    if (HAS_BG) {
        const float* bg = backgrounds + 3 * cam;
        bgdot = bg[0] * vr0 + bg[1] * vr1 + bg[2] * vr2;
        tail -= T_final * bgdot;
    }
    float T = T_final;
    float bufdot = 0.f;                         // (colour accumulated behind the current splat) . v_rgb
    if (bin_final < start) bin_final = -1;      // its contributors all lie in front of this walk's range
    if (worker && bin_final >= end) {
        // the pixel went on past the boundary: enter the segment with the forward's own state there
        const size_t pix = ((size_t)cam * H + py_i) * W + px_i;
        const float4 ck = seg.ckpt[(size_t)slot_ck * BLOCK + tid];
        const float r0 = render[3 * pix], r1 = render[3 * pix + 1], r2 = render[3 * pix + 2];
        T = ck.x;
        // colour composited behind the boundary = final colour - background share - colour at the boundary
        bufdot = (r0 - ck.y) * vr0 + (r1 - ck.z) * vr1 + (r2 - ck.w) * vr2 - T_final * bgdot;
        bin_final = end - 1;
    }
    int wmax = bin_final;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    int wmin = bin_final >= 0 ? bin_final : 0x7fffffff;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wmin = min(wmin, __shfl_xor(wmin, o, 64));
    wmin = __builtin_amdgcn_readfirstlane(wmin);
    if (lane == 0) L.wave_max[wv] = wmax;
    __syncthreads();
    const int bmax = max(max(L.wave_max[0], L.wave_max[1]), max(L.wave_max[2], L.wave_max[3]));
    if (bmax < start) return;
    RB_STAMP(2, (unsigned long long)(bmax - start + 1)); RB_STAMP(6, wall_clock64());
    const Basis basis = make_basis(wv, lane);

    // the B operands of the contraction and this lane's place in its result.
    // The v_rgb part of B (columns 6..11) needs, per lane, one colour channel of SIXTEEN of the wave's pixels -- values the
    // pixel-owner lanes already hold (vr0..2: lane l owns pixel l of the quadrant).  They go through the wave's transposition
    // rows (free until the first visit): three stores, sixteen reads.  Until round 3 every lane fetched them from v_render
    // itself, and the bounds test around each fetch compiled to load -> s_waitcnt vmcnt(0) sixteen times in a row: sixteen
    // dependent memory round trips at the head of every block's life, most of the ~15 us fixed cost per block (and the reason
    // rasterize_bwd was 37 % of a real training step whose tiles hold ~70 splats each).
    MMLane mm;
    {
        const int j = lane & 15, g = lane >> 4;
        // Every lane writes the twelve column values of ITS pixel (bf16, in both halves of a word) into twelve transposition
        // rows; a lane's operand for MFMA m is then four consecutive words of row j: one 128-bit read.  (Until late in round 3
        // each lane worked the sixteen values out itself -- pixel coordinates, selects by column, conversions, sixteen times over:
        // ~640 VALU instructions, 7.1 us of a short block's 16.2 at four waves per SIMD; tools/raster_ab.py probe.)
        {
            unsigned* const trw = &L.tr[wv][0][lane];
            const float bas[6] = {px.u, px.v, px.uu, px.uv, px.vv, 1.f};
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const unsigned hb = __builtin_bit_cast(unsigned, bas[c]) >> 16;          // exact
                trw[c * TR_STRIDE] = hb | (hb << 16);
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float val = vrgb[c];                                               // 0 outside the image
                const unsigned hi = f32_hi(val);
                const f2v r = {val - __builtin_bit_cast(float, hi), 0.f};
                const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf2v)) & 0xFFFFu;
                trw[(6 + c) * TR_STRIDE] = (hi >> 16) | hi;
                trw[(9 + c) * TR_STRIDE] = lo | (lo << 16);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // same-wave LDS hand-off
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const u4v w = *reinterpret_cast<const u4v*>(&L.tr[wv][j < 12 ? j : 0][16 * m + 4 * g]);      // pixels 16 m + 4 g + 0..3
            mm.bop[m] = j < 12 ? w : u4v{0u, 0u, 0u, 0u};
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the rows are rewritten by the first chunk
        __builtin_amdgcn_wave_barrier();
        // columns of the product: 0..5 moments (AC_QU .. AC_Q), 6..8 colour sums from the hi part of v_rgb, 9..11 from its lo part
        if (ABSGRAD) {
            // rows: Q 0..3, W 4..7, |x| 8..11, |y| 12..15  ->  lane group g holds value g of splats r = 0..3
            mm.acc_off = g == 0 ? (j < 6 ? j : -1) : g == 1 ? (j >= 6 && j < (AbsCols<SH>::MERGE ? 9 : 12) ? j : -1) : (j == 5 ? (g == 2 ? AbsCols<SH>::ABSX : AbsCols<SH>::ABSY) : -1);
        } else if ((EXP & 8) != 0) {
            // three terms: rows Q 0..3, W 4..7 (words hi | mid), Q 8..11, W 12..15 (words lo | 0); groups 2, 3 are added to 0, 1 in mm_finish
            mm.acc_off = g == 0 ? (j < 6 ? j : -1) : g == 1 ? (j >= 6 && j < 12 ? j : -1) : -1;
        } else {
            // rows: Q 0..7, W 8..15  ->  groups 0, 1 hold Q of splats 4 g + r, groups 2, 3 hold W of splats 4 (g - 2) + r
            const int col = g < 2 ? (j < 6 ? j : -1) : (j >= 6 && j < 12 ? j : -1);
            mm.acc_off = col < 0 ? -1 : 4 * (g & 1) * AW + col;
        }
    }

    const unsigned long long has = wave_ballot(bin_final >= 0);
    MMPend P;
    P.on = false;
    RB_STAMP(3, wall_clock64()); RB_STAMP(4, 0ull); RB_STAMP(5, 0ull);
    for (int be = bmax; be >= start; be -= STG) {
        __syncthreads();
        if (tid < STG) {
            const int j1 = be - tid;
            const int id_cur = j1 >= start ? flatten_ids[j1] : -1;
            const RecRegs rec = load_rec(splats, id_cur);
            stage_splat(L.f, tid, rec, xc, yc);
            if (id_cur >= 0) {
                L.geo[tid] = make_float4(rec.a.x - xc, rec.a.y - yc, rec.a.z, rec.a.w);
                L.geo2[tid] = make_float2(rec.bb.x, rec.bb.y > 0.f ? 1.f / rec.bb.y : 0.f);
                L.id[tid] = id_cur;
            }
        }
        __syncthreads();
        if (be == bmax) RB_STAMP(4, wall_clock64());
        const int bsz = min(STG, be - start + 1);
        const int k0 = max(0, be - wmax);            // wave-uniform: nothing in this wave is live before slot k0
        for (int g0 = 0; g0 < bsz; g0 += GRP) {      // block-uniform: groups of 64 slots
            unsigned long long gmask = 0ull;
#pragma unroll 1
            for (int sb = g0 / SUB; sb < g0 / SUB + GRP / SUB; sb++) {
                if (sb * SUB >= bsz || sb < k0 / SUB) continue;
                float s[SUB];
                eval_sub_batch(L.f, sb, lane, basis, s);
                if (be - sb * SUB <= wmin)
                    bwd_sub_batch_mm<ABSGRAD, true, EXP, SH>(L, s, sb, be, lane, wv, bin_final, has, px, mm, P, gmask, vrgb, tail, T, bufdot, v_splats);
                else
                    bwd_sub_batch_mm<ABSGRAD, false, EXP, SH>(L, s, sb, be, lane, wv, bin_final, has, px, mm, P, gmask, vrgb, tail, T, bufdot, v_splats);
            }
            mm_drain<ABSGRAD, (EXP & 4) != 0, (EXP & 8) != 0 && !ABSGRAD, SH>(L, P, mm, wv, lane, v_splats);
            if (EXP & 5) continue;          // 1: timing experiment, no group barriers, no flush; 4: the waves have flushed themselves
            if (lane == 0) L.gmask[wv] = gmask;
            __syncthreads();
            if (be == bmax && g0 == 0) RB_STAMP(5, wall_clock64());
            // flush: lane -> (slot = lane >> 4, column = lane & 15), 16 slots per round over the block.  A wave's 4 slots lie in
            // one chunk, so "which quadrants have sums for it" is wave-uniform.  The gradients of (x, y, conic A, B, C, opacity)
            // follow from the moments about the tile centre:
            //   sum q dx = mx M - Mu, sum q dx^2 = mx^2 M - 2 mx Mu + Muu, ...   (dx = mx - u, dy = my - v)
            const unsigned long long m0 = L.gmask[0], m1 = L.gmask[1], m2 = L.gmask[2], m3 = L.gmask[3];
            if ((m0 | m1 | m2 | m3) != 0ull) {
#pragma unroll 1
                for (int rd = 0; rd < GRP / 16; rd++) {
                    // this wave's four slots of the round: skip it if no quadrant has anything for them
                    const int sg0 = rd * 16 + 4 * wv;
                    if ((((m0 | m1 | m2 | m3) >> sg0) & 15ull) == 0ull) continue;
                    const int sg = sg0 + (lane >> 4);
                    const int comp = lane & 15;
                    const int cc = comp < AW ? comp : 0;
                    float sum = 0.f;
                    if ((m0 >> sg) & 1ull) sum += L.accw[0][sg][cc];
                    if ((m1 >> sg) & 1ull) sum += L.accw[1][sg][cc];
                    if ((m2 >> sg) & 1ull) sum += L.accw[2][sg][cc];
                    if ((m3 >> sg) & 1ull) sum += L.accw[3][sg][cc];
                    const int slot = g0 + sg;
                    const unsigned long long nz = wave_ballot(slot < bsz && comp < AW && sum != 0.f);
                    const bool touched = ((nz >> (lane & 48)) & 0xFFFFull) != 0ull;
                    // the other columns of this slot, from the 16 lanes that hold them
                    const int rb = lane & 48;
                    const float M = __shfl(sum, rb + AC_Q, 64), Mu = __shfl(sum, rb + AC_QU, 64), Mv = __shfl(sum, rb + AC_QV, 64);
                    // second operand by output component: x, y none; conic A, B, C their second moments; r, g, b the lo sums
                    const int xsrc = comp == GR_CA ? AC_QUU : comp == GR_CB ? AC_QUV : comp == GR_CC ? AC_QVV
                                     : (!(ABSGRAD && AbsCols<SH>::MERGE) && comp >= GR_R && comp <= GR_B) ? comp + 3 : comp == GR_ABSX ? AbsCols<SH>::ABSX : comp == GR_ABSY ? AbsCols<SH>::ABSY : 0;
                    const float X = __shfl(sum, rb + xsrc, 64);
                    if (slot < bsz && touched && comp < (ABSGRAD ? GR_DEPTH : GR_ABSX)) {
                        const float4 ge = L.geo[slot];
                        const float2 g2 = L.geo2[slot];
                        const float mx = ge.x, my = ge.y;
                        const float sdx = mx * M - Mu, sdy = my * M - Mv;                 // sum q dx, sum q dy
                        float val;
                        switch (comp) {
                            case GR_X: val = -(ge.z * sdx + ge.w * sdy); break;
                            case GR_Y: val = -(ge.w * sdx + g2.x * sdy); break;
                            case GR_CA: val = -0.5f * (mx * (mx * M - 2.f * Mu) + X); break;
                            case GR_CB: val = -(mx * (my * M - Mv) - my * Mu + X); break;
                            case GR_CC: val = -0.5f * (my * (my * M - 2.f * Mv) + X); break;
                            case GR_OPA: val = M * g2.y; break;
                            case GR_R: case GR_G: case GR_B: val = (ABSGRAD && AbsCols<SH>::MERGE) ? sum : sum + X; break;
                            default: val = X; break;          // |x|, |y|
                        }
                        atomicAdd(&v_splats[(size_t)L.id[slot] * GRAD_STRIDE + comp], val);
                    }
                }
            }
            __syncthreads();
        }
    }
    RB_STAMP(1, wall_clock64());
}

// The first `n_workers` blocks loop over the forward's work list (they are resident from the start of the launch and run beside
// the tiles' own blocks; behind them they were a tail: measured 425 -> 462 us on S2 with 577 items); the blocks behind them
// take one tile each.  Idle workers cost ~7 ns each.
// Dedicated workers, each with a static share of the items.  Few items (wolf 558, S1 970): 512 workers; 1 024 cost wolf 6 us of
// 119.  Many (MCMC at its cap: ~7 000 items, the tiles' own blocks done after two batches): 512 workers are half the device's
// slots and the other half stands empty -- rasterize_bwd 700 us at 512, 477 at 1 024, 480 at 2 048, 495 at 4 096, 567 at 8 192;
// the whole 30 000-step run 48.3 -> 44.3 s (profiles/r03_bwd_workers.txt).  Which case: the workspace's capacity (the launcher).
// Handing the items out by ticket instead (tiles' blocks joining in when done) was built and measured: same-address atomics from
// 3 000 blocks drain at ~30 ns each -- wolf 120 -> 208 us with a ticket and a count-out per block, 134 us with compare-and-swap
// tickets and no count-out (MCMC: 1 455 us of retries).
constexpr int SEG_WORKERS = 512, SEG_WORKERS_MANY = 1024;
constexpr size_t SEG_MANY_ITEMS = 4096;      // workspace sized for more than a million intersections

template <bool HAS_BG, bool ABSGRAD, int EXP, typename SH>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(SH::WAVES, SH::WAVES))) void rasterize_bwd_mm_kernel(
    int W, int H, int tw, int th, const float* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
    const float* __restrict__ backgrounds, const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render, const float* __restrict__ v_alphas, float* __restrict__ v_splats, int bands, int n_workers,
    SegWs seg, const float* __restrict__ render) {
    __shared__ StagedBwdMM<ABSGRAD, SH> L;
    __shared__ int s_items;
    int item = (int)blockIdx.x < n_workers ? (int)blockIdx.x : -1;            // < 0: a tile's own block
    int t = -1, n_items = 0;
    if (item < 0) {
        t = tile_of_block((int)blockIdx.x - n_workers, n_tiles_total, bands, tw);
        if (t < 0) return;
    } else {
        // The workers leave the counter clear for the next forward: one thread per worker reads it and counts itself in; the
        // last one to do so resets both words (every reader has read by then).  A clear per forward was a launch per step.
        if (threadIdx.x == 0) {
            s_items = (int)min(__hip_atomic_load(&seg.ctl[SEG_CTL_ITEMS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), seg.cap);
            if (__hip_atomic_fetch_add(&seg.ctl[SEG_CTL_OUT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)n_workers - 1u) {
                __hip_atomic_store(&seg.ctl[SEG_CTL_ITEMS], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&seg.ctl[SEG_CTL_OUT], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&seg.ctl[SEG_CTL_HEAVY], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        n_items = s_items;
    }
    for (;;) {
        if (item >= 0) {
            if (item >= n_items) return;
            __syncthreads();                             // the walk before may still be reading L
        }
        bwd_walk<HAS_BG, ABSGRAD, EXP, SH>(L, t, item, seg, render, W, H, tw, th, splats, tile_offsets, flatten_ids, n_isect_ptr,
                                           n_tiles_total, backgrounds, alphas, last_ids, v_render, v_alphas, v_splats);
        if (item < 0) return;
        item += n_workers;
    }
}

}  // namespace mfma_raster

// Which shape: WIDE (four waves per SIMD) where many blocks of similar weight keep every CU busy, DEEP (fastest single wave)
// where a few heavy tiles decide -- which, with the segment workspace, no longer happens (see the launcher).  The host knows neither list lengths nor reached fractions without a sync; it knows the number
// of Gaussians in the call, and the regimes measured separate on it (2 M and 300 k: WIDE wins by 13 % and 3 %; 100 k real
// splats: DEEP wins by 2 - 6 %).  MI3DGS_BWD_WIDE_MIN moves the switch-over (documented tuning knob, include/mi3dgs.h).
static long long bwd_wide_min() {
    static const long long v = [] { const char* e = getenv("MI3DGS_BWD_WIDE_MIN"); return e ? atoll(e) : 200000ll; }();
    return v;
}

// experiment: 0 = the product kernel, the only one the product library holds.  Experiments build (libmi3dgs_exp.so):
// 4 = three-term transport of the pixel sums (correct), 14 = the wave-flush variant (correct); 11..13 = timing experiments with
// WRONG results, kept for the measurements quoted in docs/FINDINGS_r01_r02.md (bit 0 no group barriers / flush, bit 1 no colour
// reads; only without background and absgrad); 21 / 22 = force the DEEP / WIDE shape
int mi_rasterize_bwd_mm(int n_tiles, int width, int height, int tile_width, int tile_height, long long n_gauss, const float* splats,
                        const int32_t* isect_offsets, const int32_t* flatten_ids, const int32_t* n_isect_dev,
                        const float* backgrounds, const float* alphas, const int32_t* last_ids, const float* v_render,
                        const float* v_alphas, int absgrad, float* v_splats, int experiment, const float* render, void* seg_ws,
                        size_t seg_ws_bytes, hipStream_t st) {
    using namespace mfma_raster;
    SegWs seg = {};
    if (seg_ws) {
        MI_REQUIRE(render != nullptr, "rasterize_bwd: the segment workspace needs the forward's render too");
        MI_REQUIRE(seg_ws_layout(n_tiles, seg_ws, seg_ws_bytes, &seg), "rasterize_bwd: segment workspace too small");
        if (!seg_ws_in_use(n_tiles, seg_ws_bytes)) seg = SegWs{};      // as the forward decided
    }
    // (the workspace was sized for max_isect / 256 + n_tiles + 16 items: what is beyond the tiles' share is the host's only
    //  estimate of how many intersections the caller expects)
    const bool many_items = seg.ckpt && (size_t)seg.cap > (size_t)n_tiles + 16 + SEG_MANY_ITEMS;
    const int n_workers = seg.ckpt ? (many_items ? SEG_WORKERS_MANY : SEG_WORKERS) : 0;
    const int grid = raster_grid(n_tiles, tile_width) + n_workers;
#define LAUNCH_MM(BG, AG, E, SH)                                                                                                 \
    MI_LAUNCH("rasterize_bwd", (rasterize_bwd_mm_kernel<BG, AG, E, SH>), dim3(grid), dim3(BLOCK), 0, st, width, height, tile_width, \
              tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, alphas, last_ids,        \
              v_render, v_alphas, v_splats, raster_bands(), n_workers, seg, render)
    // with the segment workspace no walk is longer than 512 entries and a launch is many short blocks whose fixed cost (three
    // dependent round trips, 11 - 16 us) is what the fourth resident block hides: WIDE then wins on the real-splat regime too
    // (wolf 960x720 125 -> 114 us, 1920x1080 256 -> 215; 640x480 92 -> 96), profiles/r03_raster_bwd_shape_ab.txt
    bool wide = n_gauss >= bwd_wide_min() || seg.ckpt != nullptr;
#ifdef MI3DGS_EXPERIMENTS
    if (experiment == 21 || experiment == 22) { wide = experiment == 22; experiment = 0; }
    if (experiment == 4 && !absgrad) {           // three-term transport of the pixel sums (correct results, 24 significant bits)
        if (backgrounds) LAUNCH_MM(true, false, 8, ShapeDeep); else LAUNCH_MM(false, false, 8, ShapeDeep);
        MI_LAUNCH_CHECK();
        return 0;
    }
    if (experiment == 14) {           // wave-flush variant (correct results)
        if (backgrounds) { if (absgrad) LAUNCH_MM(true, true, 4, ShapeDeep); else LAUNCH_MM(true, false, 4, ShapeDeep); }
        else { if (absgrad) LAUNCH_MM(false, true, 4, ShapeDeep); else LAUNCH_MM(false, false, 4, ShapeDeep); }
        MI_LAUNCH_CHECK();
        return 0;
    }
    if (experiment >= 11 && experiment <= 13 && !backgrounds && !absgrad) {
        if (experiment == 11) LAUNCH_MM(false, false, 1, ShapeDeep);
        else if (experiment == 12) LAUNCH_MM(false, false, 2, ShapeDeep);
        else LAUNCH_MM(false, false, 3, ShapeDeep);
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    MI_REQUIRE(experiment == 0, "rasterize_bwd: unknown variant");
    if (wide) {
        if (backgrounds) { if (absgrad) LAUNCH_MM(true, true, 0, ShapeWide); else LAUNCH_MM(true, false, 0, ShapeWide); }
        else { if (absgrad) LAUNCH_MM(false, true, 0, ShapeWide); else LAUNCH_MM(false, false, 0, ShapeWide); }
    } else {
        if (backgrounds) { if (absgrad) LAUNCH_MM(true, true, 0, ShapeDeep); else LAUNCH_MM(true, false, 0, ShapeDeep); }
        else { if (absgrad) LAUNCH_MM(false, true, 0, ShapeDeep); else LAUNCH_MM(false, false, 0, ShapeDeep); }
    }
#undef LAUNCH_MM
    MI_LAUNCH_CHECK();
    return 0;
}
