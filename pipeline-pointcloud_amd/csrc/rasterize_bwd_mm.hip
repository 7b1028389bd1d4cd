// Backward tile rasteriser with the per-splat pixel sums contracted on the matrix pipe (bf16 MFMA, two-term split).
// gfx950 only.  Replaces gsplat rasterize_to_pixels_bwd, reached by the reference only through
// main.py:1312 / main.py:1343.  Built with -fno-slp-vectorize (see the Makefile).
#include "rasterize_mfma.h"
#include <stdlib.h>
#include <type_traits>

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
//
// Round 4 (same-box A/B of every step with tools/raster_ab.py, profiles/r04_raster_bwd_steps.txt; S2 plain 369 -> 308 us, with
// absgrad 573 -> 471, S1 119 -> 99, wolf 960 x 720 with absgrad 151 -> 122):
//   * the chain carries bd = (colour behind the splat) . v_rgb - tail and forms q = (alpha ra) (T cv - bd);
//   * a splat's colour (and, with ABSGRAD, the five numbers of d sigma / d(mx, my) as forms in (u, v)) is read ONCE per sub-batch,
//     lane l holding splat l's, and handed to the visits by v_readlane: the wave-uniform LDS read per visit queued behind the two
//     stores of the visit before it (LDS serves a wave in order) and every visit's block waited for it;
//   * the rows of one splat are adjacent (row = kinds x splat + kind): a visit's words leave as ds_write2_b32 and a lane of the
//     accumulator holds ONE splat's sums (one cross-lane step instead of four in the ABSGRAD merge, half the stores);
//   * a sub-batch whose 32 splats all have opacity < 0.998, in a wave all of whose pixels composited something, runs an instance
//     without the 0.999 clamp and without the `has` mask (no v_min, no clamp compare, no select on q);
//   * the group flush gives every slot ONE lane (flush_group_lane_per_slot).  Timing probes had priced the old flush's
//     arithmetic at 70 us of S2's 354 -- its atomics at nothing, its two barriers at 7;
//   * ABSGRAD rebuilds the alpha basis in front of every sub-batch instead of holding its ten registers across the visits.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int TR_ROWS = 16;
constexpr int GRP = 64;                   // slots per flush group: the per-wave sums of 64 splats wait in LDS, then the block flushes them
constexpr int AW = 12;                    // floats per (wave, slot) sum row: 48 bytes, read by the flush as three float4
// ---- ONE shape: staged batches of 128 splats, 128 registers, 40 KB of LDS = four blocks per CU, four waves per SIMD; a chunk's
// operands are read and its four MFMAs issued when the chunk ends.  Rounds 2 - 3 also had a DEEP shape (256-splat batches, the
// contraction software-pipelined one MFMA per visit into the NEXT chunk's visits through 20 more live registers, 168 registers,
// three waves per SIMD): 12 % slower on S2, a tie on S1, 2 - 6 % faster only on ~100 k real splats WITHOUT the segment
// workspace -- a case training never runs (profiles/r03_raster_bwd_shape_ab.txt).  Removed in round 4 with its launcher rule.
constexpr int STG = 128;                  // splats per staged batch
constexpr int BWD_WAVES = 4;              // waves per SIMD the register allocation is held to
// columns of a per-wave sum row: the nine of the reduce-scatter kernel (AC_*), then, without ABSGRAD, the lo part of the colour
// sums (9..11; the flush adds them); with ABSGRAD the lo colour sums are added to the hi ones before they are stored (one DPP
// step in mm_finish) and |x|, |y| take columns 9, 10.
constexpr int ABS_COL_X = 9, ABS_COL_Y = 10;

// LDS float atomics are lane-serial on this part (tools/micro/lds_ops.hip: ds_add_f32 takes ~3 LDS cycles per ACTIVE lane,
// 55 for the 18 lanes that would add a chunk's sums, against 3 for a plain ds_write_b32), so every wave keeps its own sums
// (plain stores, each (wave, slot, column) written at most once per group) and the flush adds the four quadrants.
template <bool ABSGRAD>
struct StagedBwdMM {
    // words per transposition row.  (68 puts rows 11 and 12 of a ds_read_b128's 16-lane groups on one bank; 72 is conflict-free for
    // that read and measured the same to 0.1 % on S2, S1 and wolf -- the reads are not what the kernel waits for -- so the smaller one stays.)
    static constexpr int TRS = 68;
    StagedN<STG> f;           // (the Gaussian's index rides in the fourth word of its colour record, f.uni[slot].w)
    float4 geo[STG];          // mx, my (relative to the tile centre), A, B
    // C, 1 / o [, ax = A mx + B my, ay = B mx + C my: the constant terms of d sigma / d(mx, my) as forms in (u, v)]
    typename std::conditional<ABSGRAD, float4, float2>::type geo2[STG];
    alignas(16) float accw[4][GRP][AW];
    alignas(16) unsigned tr[4][TR_ROWS][TRS];      // per wave: rows = (splat of the chunk, value), columns = the wave's 64 pixels
    unsigned hot[STG / SUB];            // per sub-batch: some splat's opacity is >= 0.998 (the alpha clamp at 0.999 can be active)
    unsigned long long gmask[4];        // per wave: slots of the current group whose visit was live (its sums are meaningful)
    int wave_max[4];
};

template <typename LT>
__device__ __forceinline__ int slot_id(const LT& L, int slot) { return reinterpret_cast<const int*>(&L.f.uni[slot])[3]; }

// what a lane does with the accumulator of a chunk: D row 4 (lane >> 4) + r, column lane & 15
struct MMLane {
    u4v bop[4];               // B operand of MFMA m: column (lane & 15) at the pixels 16 m + 4 (lane >> 4) + 0..3, both parts
    int acc_off;              // float offset of this lane's column inside the sum row of its (first) splat of a chunk, or -1: nothing to store
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

// the same for |a|, |b|: the mask that takes the hi part drops the sign with it
__device__ __forceinline__ void split2_abs(float a, float b, unsigned& wa, unsigned& wb) {
    const unsigned ha = __builtin_bit_cast(unsigned, a) & 0x7FFF0000u, hb = __builtin_bit_cast(unsigned, b) & 0x7FFF0000u;
    const f2v r = {__builtin_fabsf(a) - __builtin_bit_cast(float, ha), __builtin_fabsf(b) - __builtin_bit_cast(float, hb)};
    const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf2v));
    wa = ha | (l & 0xFFFFu);
    wb = hb | (l >> 16);
}

// three terms (experiments build, the precision A/B of VERDICT r2 #3 / r3 #4): hi and mid are exact truncations, lo is rounded --
// 24 significant bits.  Word A = hi | mid, word B = lo | 0; both multiply the same basis, so the two words are two more ROWS of
// the same product whose results add up.
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

// the contraction of one chunk: operands, accumulator, first slot of the chunk inside its group
struct MMPend {
    u4v a[4];
    f4v d;
    int slot0;
};

__device__ __forceinline__ void mm_issue(MMPend& P, const MMLane& mm, int m) {
    P.d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, P.a[m]), __builtin_bit_cast(bf8v, mm.bop[m]), P.d, 0, 0, 0);
}

// Row layout of a chunk (T3 = the three-term experiment keeps the round-3 layout, kind-major):
//   plain:    row 2 c + kind,  kind = Q, W             -> D row 4 g + r is splat 2 g + (r >> 1), kind r & 1
//   ABSGRAD:  row 4 c + kind,  kind = Q, W, |x|, |y|   -> D row 4 g + r is splat g, kind r
template <bool ABSGRAD, bool T3>
__device__ __forceinline__ void mm_finish(StagedBwdMM<ABSGRAD>& L, MMPend& P, const MMLane& mm, int wv, int lane) {
    const int j = lane & 15;
    float* a = &L.accw[wv][P.slot0][0];
    if (T3) {
        // rows Q 0..3, W 4..7 (words hi | mid), Q 8..11, W 12..15 (words lo | 0): rows 8..15 (lanes 32..63) onto rows 0..7
#pragma unroll
        for (int r = 0; r < 4; r++) P.d[r] += __shfl_down(P.d[r], 32, 64);
        if (mm.acc_off >= 0) { a += mm.acc_off; a[0] = P.d[0]; a[AW] = P.d[1]; a[2 * AW] = P.d[2]; a[3 * AW] = P.d[3]; }
    } else if (ABSGRAD) {
        // colour sums: column j + 3 (from the lo part of v_rgb) onto column j = 6..8, lane j + 3 of the same 16-lane row: DPP
        // row_shl:3 (a __shfl_down is a ds_bpermute, an LDS round trip per chunk).  (The element is copied out first:
        // __builtin_bit_cast applied directly to `P.d[1]` reads element 0 with this clang.)
        const float d1 = P.d[1];
        const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d1), 0x103, 0xF, 0xF, true));
        const float w1 = d1 + ((j >= 6 && j < 9) ? up : 0.f);
        if (mm.acc_off >= 0) a[mm.acc_off] = j < 6 ? P.d[0] : w1;
        if (j == 5) { a[mm.acc_off - 5 + ABS_COL_X] = P.d[2]; a[mm.acc_off - 5 + ABS_COL_Y] = P.d[3]; }      // column 5 = the plain sums
    } else if (mm.acc_off >= 0) {
        a[mm.acc_off] = j < 6 ? P.d[0] : P.d[1];
        a[mm.acc_off + AW] = j < 6 ? P.d[2] : P.d[3];
    }
}

template <bool ABSGRAD>
__device__ __forceinline__ void mm_read(StagedBwdMM<ABSGRAD>& L, MMPend& P, int wv, int lane, int slot0) {
    asm volatile("" ::: "memory");      // the rows were written by other lanes of this wave: LDS is in order per wave
    const u4v* ap = reinterpret_cast<const u4v*>(&L.tr[wv][lane & 15][4 * (lane >> 4)]);
    P.a[0] = ap[0]; P.a[1] = ap[4]; P.a[2] = ap[8]; P.a[3] = ap[12];
    // (hipcc hoists the MFMAs between these reads to reuse one register quad for three of them: read, read, wait, MFMA, wait, MFMA,
    //  read, wait, ...  Forcing all four reads in front of the first MFMA -- 16 registers, the prologue spills for it -- measured
    //  WORSE: S2 313 -> 322 us, absgrad 484 -> 488, S1 103 -> 109; the first MFMA starts later and nothing else fills the wait.)
    asm volatile("" ::: "memory");      // the next chunk's stores stay behind these reads
    P.d = f4v{0.f, 0.f, 0.f, 0.f};
    P.slot0 = slot0;
}

template <bool ABSGRAD, bool T3>
__device__ __forceinline__ void mm_drain(StagedBwdMM<ABSGRAD>& L, MMPend& P, const MMLane& mm, int wv, int lane) {
#pragma unroll
    for (int m = 0; m < 4; m++) mm_issue(P, mm, m);
    mm_finish<ABSGRAD, T3>(L, P, mm, wv, lane);
}

// what lane l holds of splat (l & 31) of the sub-batch: handed to the visits by v_readlane
struct SubUni { float r, g, b; float A, B, C, ax, ay; };

__device__ __forceinline__ float lane_bcast(float v, int i) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
}

template <bool ABSGRAD>
__device__ __forceinline__ SubUni load_sub_uni(const StagedBwdMM<ABSGRAD>& L, int sb, int lane) {
    SubUni U = {};
    const int k = sb * SUB + (lane & 31);
    const float4 c = L.f.uni[k];
    U.r = c.x; U.g = c.y; U.b = c.z;
    if constexpr (ABSGRAD) {
        const float4 ge = L.geo[k], g2 = L.geo2[k];
        U.A = ge.z; U.B = ge.w; U.C = g2.x; U.ax = g2.z; U.ay = g2.w;
    }
    return U;
}

// One sub-batch of the backward walk, rows i = 0..31 <-> sorted indices be - 32 sb - i (back to front).
// FAST: every pixel of the wave composited something and is already in range (index <= its last contributor), and none of the
// sub-batch's splats can reach the 0.999 clamp (the caller's test): a visit's membership test is the threshold compare alone.
template <bool ABSGRAD, bool FAST, bool T3_>
__device__ __forceinline__ void bwd_sub_batch_mm(StagedBwdMM<ABSGRAD>& L, const float (&s)[SUB], int sb, int be, int lane, int wv,
                                                 int bin_final, unsigned long long has, const PixelBasis& px, const MMLane& mm,
                                                 MMPend& P, unsigned long long& gmask, const float (&vrgb)[3],
                                                 float& T, float& bd, const SubUni& U) {
    constexpr bool T3 = T3_ && !ABSGRAD;         // three-term transport (experiments build)
    constexpr int CH = (ABSGRAD || T3) ? 4 : 8;  // splats per chunk: CH x (2 or 4 values) = 16 rows
    constexpr int KINDS = 16 / CH;
    constexpr int TR_STRIDE = StagedBwdMM<ABSGRAD>::TRS;
    const int gs0 = (sb * SUB) & (GRP - 1);      // slot of row 0 inside its group
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
        // the forward's own membership test (same MFMA result, same compare), for the splats this pixel reached
        unsigned long long valid = mask_ge(s[i], LOG2_ALPHA_THRESHOLD);
        if (!FAST) valid &= has & mask_ge_i(bin_final, be - k);
        if (valid != 0ull) {
            live |= 1u << i;
            const Rgb col = {lane_bcast(U.r, i), lane_bcast(U.g, i), lane_bcast(U.b, i)};
            const float alpha = FAST ? __builtin_amdgcn_exp2f(s[i]) : alpha_of(s[i]);
            // branch-free live part: a lane that does not take part runs it with alpha = 0 (ra = 1, T and bd unchanged bit for
            // bit, both rows 0)
            const float a_eff = lane_of(valid) ? alpha : 0.f;
            const float ra = __builtin_amdgcn_rcpf(1.f - a_eff);
            const float cv = col.x * vrgb[0] + col.y * vrgb[1] + col.z * vrgb[2];            // c . v_rgb
            // v_alpha = T_i cv - ra bd = ra (T cv - bd) with the T BEHIND the splat; q = o vis dL/dalpha = -dL/dsigma = (alpha ra)
            // (T cv - bd), zero where the 0.999 clamp is active (alpha == o vis otherwise)
            float w = __builtin_fmaf(T, cv, -bd);
            // (w before the updates: T and bd are then updated in place; left to itself hipcc forms the new T first and pays a
            //  register copy for each of the two at the end of the visit)
            asm volatile("" : "+v"(w), "+v"(T), "+v"(bd));
            T *= ra;
            const float fac = a_eff * T;
            bd = __builtin_fmaf(cv, fac, bd);
            float q = (a_eff * ra) * w;
            if (!FAST) q = lane_of(valid & mask_le(s[i], LOG2_MAX_ALPHA)) ? q : 0.f;
            unsigned wq, wf;
            if (T3) {
                unsigned wq2, wf2;
                split3w(q, wq, wq2);
                split3w(fac, wf, wf2);
                trw[ci * TR_STRIDE] = wq;
                trw[(CH + ci) * TR_STRIDE] = wf;
                trw[(2 * CH + ci) * TR_STRIDE] = wq2;
                trw[(3 * CH + ci) * TR_STRIDE] = wf2;
            } else {
                split2(q, fac, wq, wf);
                trw[KINDS * ci * TR_STRIDE] = wq;
                trw[(KINDS * ci + 1) * TR_STRIDE] = wf;
            }
            if constexpr (ABSGRAD) {
                // d sigma / d(mx, my) = (A dx + B dy, B dx + C dy) = (ax - A u - B v, ay - B u - C v)
                // (one scalar operand per instruction: a second one costs a v_mov)
                const float sB = lane_bcast(U.B, i);
                const float lx = lane_bcast(U.ax, i) - __builtin_fmaf(lane_bcast(U.A, i), px.u, sB * px.v);
                const float ly = lane_bcast(U.ay, i) - __builtin_fmaf(lane_bcast(U.C, i), px.v, sB * px.u);
                unsigned wx, wy;
                split2_abs(q * lx, q * ly, wx, wy);
                trw[(KINDS * ci + 2) * TR_STRIDE] = wx;
                trw[(KINDS * ci + 3) * TR_STRIDE] = wy;
            }
        }
        if (ci == CH - 1) {
            const unsigned cl = (live >> (i - (CH - 1))) & ((1u << CH) - 1u);
            if (cl) {           // read, four MFMAs, store: nothing stays live across visits
                mm_read<ABSGRAD>(L, P, wv, lane, gs0 + i - (CH - 1));
                mm_drain<ABSGRAD, T3>(L, P, mm, wv, lane);
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

// The flush of one group of 64 slots; wave w takes the slots 16 w .. 16 w + 15.  ONE lane per slot adds the quadrants' sum rows
// (three 128-bit reads each; quadrant 0 first), forms the gradient components with straight-line code and leaves them in the
// wave's own transposition rows (free between two chunks); the float atomics then go out sixteen lanes per slot, one 64-byte
// request per (tile, Gaussian).  The gradients of (x, y, conic A, B, C, opacity) follow from the moments about the tile centre:
//   sum q dx = mx M - Mu, sum q dx^2 = mx^2 M - 2 mx Mu + Muu, ...   (dx = mx - u, dy = my - v)
// The flush this replaces (rounds 2, 3) gave every slot sixteen lanes that each summed one column, fetched M, Mu, Mv and a
// fourth value from their neighbours through ds_bpermute and ran an eleven-way switch on their component: ~120 instructions
// per round of 16 slots.
template <bool ABSGRAD>
__device__ __forceinline__ void flush_group_lane_per_slot(StagedBwdMM<ABSGRAD>& L, int wv, int lane, int g0, int bsz,
                                                          float* __restrict__ v_splats) {
    constexpr int OUT_STRIDE = 20;            // floats per output row: 16-byte aligned, and 20 l mod 64 keeps eight rows' float4 stores on disjoint banks
    constexpr int NCOMP = ABSGRAD ? GR_DEPTH : GR_ABSX;
    const unsigned long long m0 = L.gmask[0], m1 = L.gmask[1], m2 = L.gmask[2], m3 = L.gmask[3];
    const int sh = 16 * wv;
    const unsigned f0 = (unsigned)(m0 >> sh) & 0xFFFFu, f1 = (unsigned)(m1 >> sh) & 0xFFFFu, f2 = (unsigned)(m2 >> sh) & 0xFFFFu,
                   f3 = (unsigned)(m3 >> sh) & 0xFFFFu;
    unsigned any = f0 | f1 | f2 | f3;
    const int left = bsz - (g0 + sh);                      // slots of this wave that exist in the batch
    if (left < 16) any &= left > 0 ? (1u << left) - 1u : 0u;
    if (any == 0u) return;                                 // wave-uniform
    float* const out = reinterpret_cast<float*>(&L.tr[wv][0][0]);
    if (lane < 16 && ((any >> lane) & 1u)) {
        const int sg = sh + lane;
        float acc[12];
#pragma unroll
        for (int c = 0; c < 12; c++) acc[c] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned fq = q == 0 ? f0 : q == 1 ? f1 : q == 2 ? f2 : f3;
            if ((fq >> lane) & 1u) {
                const float4* row = reinterpret_cast<const float4*>(&L.accw[q][sg][0]);
                const float4 a = row[0], b = row[1], c = row[2];
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
                acc[8] += c.x; acc[9] += c.y; acc[10] += c.z; acc[11] += c.w;
            }
        }
        const int slot = g0 + sg;
        const float4 ge = L.geo[slot];
        const auto g2 = L.geo2[slot];
        const float mx = ge.x, my = ge.y;
        const float M = acc[AC_Q], Mu = acc[AC_QU], Mv = acc[AC_QV];
        const float sdx = mx * M - Mu, sdy = my * M - Mv;                 // sum q dx, sum q dy
        float4 o0, o1, o2;
        o0.x = -(ge.z * sdx + ge.w * sdy);                                // GR_X
        o0.y = -(ge.w * sdx + g2.x * sdy);                                // GR_Y
        o0.z = -0.5f * (mx * (mx * M - 2.f * Mu) + acc[AC_QUU]);          // GR_CA
        o0.w = -(mx * (my * M - Mv) - my * Mu + acc[AC_QUV]);             // GR_CB
        o1.x = -0.5f * (my * (my * M - 2.f * Mv) + acc[AC_QVV]);          // GR_CC
        o1.y = M * g2.y;                                                  // GR_OPA
        if (ABSGRAD) {                                                    // (lo colour sums already merged in mm_finish)
            o1.z = acc[AC_R]; o1.w = acc[AC_G]; o2.x = acc[AC_B];
            o2.y = acc[ABS_COL_X]; o2.z = acc[ABS_COL_Y];                 // GR_ABSX, GR_ABSY
        } else {
            o1.z = acc[AC_R] + acc[AC_R + 3]; o1.w = acc[AC_G] + acc[AC_G + 3]; o2.x = acc[AC_B] + acc[AC_B + 3];
            o2.y = 0.f; o2.z = 0.f;
        }
        o2.w = 0.f;
        float4* orow = reinterpret_cast<float4*>(out + lane * OUT_STRIDE);
        orow[0] = o0; orow[1] = o1; orow[2] = o2;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // same-wave LDS hand-off (LDS serves a wave in order)
    __builtin_amdgcn_wave_barrier();
    const int sl = lane >> 4, comp = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (((any >> (4 * r)) & 15u) == 0u) continue;      // wave-uniform
        const int s16 = 4 * r + sl;
        if (((any >> s16) & 1u) && comp < NCOMP)
            atomicAdd(&v_splats[(size_t)slot_id(L, g0 + sh + s16) * GRAD_STRIDE + comp], out[s16 * OUT_STRIDE + comp]);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the rows are rewritten by the next chunk's visits
    __builtin_amdgcn_wave_barrier();
}

// One walk of rasterize_bwd: the entries [lo, ...] of tile t's list, back to front, for the block's 256 pixels.
//   item < 0: the tile's own block -- everything behind the last boundary the forward left (the whole list if it left none);
//   item >= 0: segment `item` of the work list -- the entries in front of a boundary, state from the forward's checkpoint.
template <bool HAS_BG, bool ABSGRAD, bool T3>
__device__ __forceinline__ void bwd_walk(StagedBwdMM<ABSGRAD>& L, int t, int item, const SegWs& seg, const float* __restrict__ render,
    int W, int H, int tw, int th, const float* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
    const float* __restrict__ backgrounds, const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render, const float* __restrict__ v_alphas, float* __restrict__ v_splats) {
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
    float bd = bufdot - tail;                   // what the chain carries
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
            constexpr int TR_STRIDE = StagedBwdMM<ABSGRAD>::TRS;
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
        if (T3 && !ABSGRAD) {
            // three terms: rows Q 0..3, W 4..7 (words hi | mid), Q 8..11, W 12..15 (words lo | 0); groups 2, 3 are added to 0, 1 in mm_finish
            mm.acc_off = g == 0 ? (j < 6 ? j : -1) : g == 1 ? (j >= 6 && j < 12 ? j : -1) : -1;
        } else {
            // rows of one splat adjacent (mm_finish): this lane's column inside the sum row of ITS splat (ABSGRAD: splat g, the lo
            // colour columns merged away; else splats 2 g and 2 g + 1, a row of AW floats apart)
            mm.acc_off = j < (ABSGRAD ? 9 : 12) ? (ABSGRAD ? g : 2 * g) * AW + j : -1;
        }
    }

    const unsigned long long has = wave_ballot(bin_final >= 0);
    MMPend P;
    RB_STAMP(3, wall_clock64()); RB_STAMP(4, 0ull); RB_STAMP(5, 0ull);
    for (int be = bmax; be >= start; be -= STG) {
        __syncthreads();
        if (tid < STG) {
            const int j1 = be - tid;
            const int id_cur = j1 >= start ? flatten_ids[j1] : -1;
            const RecRegs rec = load_rec(splats, id_cur);
            stage_splat(L.f, tid, rec, xc, yc);
            if (id_cur >= 0) {
                const float mx_ = rec.a.x - xc, my_ = rec.a.y - yc, io_ = rec.bb.y > 0.f ? 1.f / rec.bb.y : 0.f;
                L.geo[tid] = make_float4(mx_, my_, rec.a.z, rec.a.w);
                if constexpr (ABSGRAD)
                    L.geo2[tid] = make_float4(rec.bb.x, io_, __builtin_fmaf(rec.a.z, mx_, rec.a.w * my_), __builtin_fmaf(rec.a.w, mx_, rec.bb.x * my_));
                else
                    L.geo2[tid] = make_float2(rec.bb.x, io_);
                reinterpret_cast<int*>(&L.f.uni[tid])[3] = id_cur;       // (behind stage_splat's store of the record, same thread)
            }
            // log2 alpha <= log2 o + 3e-4 (rasterize_mfma.h: alpha_of): below o = 0.998 the clamp at 0.999 can not be reached
            const unsigned long long hb = wave_ballot(id_cur >= 0 && rec.bb.y >= 0.998f);
            if ((lane & 31) == 0) L.hot[tid >> 5] = (unsigned)(hb >> (lane & 32)) != 0u ? 1u : 0u;
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
                const SubUni U = load_sub_uni<ABSGRAD>(L, sb, lane);       // (in front of the MFMAs: their latency covers the reads)
                if (ABSGRAD) {
                    // the alpha basis rebuilt here instead of held across the visits: ten registers that let hipcc keep the ABSGRAD
                    // instance at 128 without spilling (plain: 305 -> 314 us, ABSGRAD 474 -> 471; only the latter takes it)
                    int wv_ = wv, lane_ = lane;
                    asm volatile("" : "+v"(wv_), "+v"(lane_));
                    eval_sub_batch(L.f, sb, lane, make_basis(wv_, lane_), s);
                } else {
                    eval_sub_batch(L.f, sb, lane, basis, s);
                }
                // (the reads have landed behind the MFMAs; used here once, hipcc does not wait for them again in every visit's block)
                asm volatile("" :: "v"(U.r), "v"(U.g), "v"(U.b));
                if (ABSGRAD) asm volatile("" :: "v"(U.A), "v"(U.B), "v"(U.C), "v"(U.ax), "v"(U.ay));
                if (be - sb * SUB <= wmin && has == ~0ull && !L.hot[sb])
                    bwd_sub_batch_mm<ABSGRAD, true, T3>(L, s, sb, be, lane, wv, bin_final, has, px, mm, P, gmask, vrgb, T, bd, U);
                else
                    bwd_sub_batch_mm<ABSGRAD, false, T3>(L, s, sb, be, lane, wv, bin_final, has, px, mm, P, gmask, vrgb, T, bd, U);
            }
            if (lane == 0) L.gmask[wv] = gmask;
            __syncthreads();
            if (be == bmax && g0 == 0) RB_STAMP(5, wall_clock64());
            flush_group_lane_per_slot<ABSGRAD>(L, wv, lane, g0, bsz, v_splats);
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

template <bool HAS_BG, bool ABSGRAD, bool T3>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BWD_WAVES, BWD_WAVES))) void rasterize_bwd_mm_kernel(
    int W, int H, int tw, int th, const float* __restrict__ splats, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ n_isect_ptr, int n_tiles_total,
    const float* __restrict__ backgrounds, const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render, const float* __restrict__ v_alphas, float* __restrict__ v_splats, int n_workers,
    SegWs seg, const float* __restrict__ render) {
    __shared__ StagedBwdMM<ABSGRAD> L;
    __shared__ int s_items;
    int item = (int)blockIdx.x < n_workers ? (int)blockIdx.x : -1;            // < 0: a tile's own block
    int t = -1, n_items = 0;
    if (item < 0) {
        t = (int)blockIdx.x - n_workers;
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
        bwd_walk<HAS_BG, ABSGRAD, T3>(L, t, item, seg, render, W, H, tw, th, splats, tile_offsets, flatten_ids, n_isect_ptr,
                                          n_tiles_total, backgrounds, alphas, last_ids, v_render, v_alphas, v_splats);
        if (item < 0) return;
        item += n_workers;
    }
}

}  // namespace mfma_raster

// experiment: 0 = the product kernel, the only one the product library holds.  Experiments build (libmi3dgs_exp.so):
// 4 = three-term transport of the pixel sums (24 significant bits; the precision A/B of profiles/r03_bwd_terms_ab.txt and
// profiles/r04_precision_ab.txt).
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
    const int grid = n_tiles + n_workers;
#define LAUNCH_MM(BG, AG, T3)                                                                                                     \
    MI_LAUNCH("rasterize_bwd", (rasterize_bwd_mm_kernel<BG, AG, T3>), dim3(grid), dim3(BLOCK), 0, st, width, height, tile_width, \
              tile_height, splats, isect_offsets, flatten_ids, n_isect_dev, n_tiles, backgrounds, alphas, last_ids,        \
              v_render, v_alphas, v_splats, n_workers, seg, render)
#ifdef MI3DGS_EXPERIMENTS
    if (experiment == 4 && !absgrad) {           // three-term transport of the pixel sums (correct results, 24 significant bits)
        if (backgrounds) LAUNCH_MM(true, false, true); else LAUNCH_MM(false, false, true);
        MI_LAUNCH_CHECK();
        return 0;
    }
#endif
    MI_REQUIRE(experiment == 0, "rasterize_bwd: unknown variant");
    (void)n_gauss;
    if (backgrounds) { if (absgrad) LAUNCH_MM(true, true, false); else LAUNCH_MM(true, false, false); }
    else { if (absgrad) LAUNCH_MM(false, true, false); else LAUNCH_MM(false, false, false); }
#undef LAUNCH_MM
    MI_LAUNCH_CHECK();
    return 0;
}
