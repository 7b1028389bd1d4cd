// Tile binning for the 3DGS rasteriser: per-Gaussian tile counts, depth sort, key emission,
// stable per-tile sort, tile offsets.  gfx950 only; hand-written scan + LSD radix sort
// (no rocPRIM/hipCUB).
//
// Replaces gsplat isect_tiles (two passes + cub::DeviceRadixSort on 64-bit keys) and
// isect_offset_encode, reached by the reference only through
// main.py:1312 / main.py:1343.
//
// Design (differs from the upstream on purpose): instead of one 64-bit sort over all I
// intersections (tile bits + 32 depth bits = 6 passes of 8 bits at 1080p), sort the C*N
// Gaussians by depth ONCE (4 passes over C*N 32-bit keys), emit their tile keys in that
// order, then a STABLE sort of the I intersections by tile id only (ceil(log2(C*tiles))
// bits = 2 passes at 1080p).  Stable + same tie-break (Gaussian index) => the per-tile
// lists are identical to the upstream's (tile|depth) ordering.  Algorithmic bytes drop from
// I*12*2*6 to CN*8*2*4 + I*8*2*2.
//
// Every kernel takes a host-known capacity for its grid and reads the live element count
// from device memory, so the stage runs without a host sync when the caller sizes the
// intersection buffers by capacity.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, total in *total
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* total, uint32_t* lds4) {
    int lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan_u32(v);
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t s = lds4[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

constexpr unsigned CHAIN_SPIN_LIMIT = 1u << 24;

// One device word collects the "a bounded spin ran out" bits of every chained kernel (bit 0: onesweep
// radix pass, bit 1: chained scan / fused tile emit).  It stays set until read: mi3dgs_async_errors().
__device__ uint32_t g_async_err = 0;
uint32_t* async_err_ptr() {
    static thread_local uint32_t* p[16] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!p[dev] && hipGetSymbolAddress((void**)&p[dev], HIP_SYMBOL(g_async_err)) != hipSuccess) p[dev] = nullptr;
    return p[dev];
}

// ---- single-value decoupled look-back (chained scan) (used by the device-wide scan and by the fused count+emit kernel).  One
// 64-bit word per block {flag:2 (bits 32..33) | value:32}: flag 1 = the block's own total,
// 2 = inclusive prefix; zero (the memset state) = not there yet.  Block ids are handed out by an
// atomic counter, so every predecessor is already running and each wait ends.  Wave 0 calls this
// with all 64 lanes and inspects 64 predecessors per step.  Agent-scope atomics: the per-XCD L2s
// are not coherent with one another.
__device__ __forceinline__ unsigned long long chain_pack(uint32_t flag, uint32_t v) {
    return ((unsigned long long)flag << 32) | v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t chain_lookback(unsigned long long* status, uint32_t blk, uint32_t total, int lane,
                                                   uint32_t* err) {
    if (lane == 0)
        __hip_atomic_store(status + blk, chain_pack(blk == 0 ? 2u : 1u, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    if (blk > 0) {
        int p = (int)blk - 1;
        while (true) {
            int idx = p - lane;
            unsigned long long w = chain_pack(2u, 0u);              // before block 0: prefix 0
            if (idx >= 0) {
                unsigned spins = 0;
                while (true) {
                    w = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (((w >> 32) & 3ull) != 0ull) break;
                    if (++spins > CHAIN_SPIN_LIMIT) { atomicOr(err, 2u); w = chain_pack(2u, 0u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            uint32_t val = (uint32_t)w;
            unsigned long long pm = wave_ballot(((w >> 32) & 3ull) == 2ull);
            if (pm) {
                int f = __ffsll((long long)pm) - 1;                  // nearest predecessor holding a prefix
                excl += wave_sum_u32(lane <= f ? val : 0u);
                break;
            }
            excl += wave_sum_u32(val);
            p -= 64;
        }
        if (lane == 0)
            __hip_atomic_store(status + blk, chain_pack(2u, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return excl;
}

// The look-back alone: exclusive prefix of the entries before `blk` (entries publish themselves elsewhere).
__device__ __forceinline__ uint32_t chain_prefix_before(const unsigned long long* status, uint32_t blk, int lane, uint32_t* err) {
    uint32_t excl = 0;
    int p = (int)blk - 1;
    while (p >= 0) {
        int idx = p - lane;
        unsigned long long w = chain_pack(2u, 0u);              // before entry 0: prefix 0
        if (idx >= 0) {
            unsigned spins = 0;
            while (true) {
                w = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((w >> 32) & 3ull) != 0ull) break;
                if (++spins > CHAIN_SPIN_LIMIT) { atomicOr(err, 2u); w = chain_pack(2u, 0u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const uint32_t val = (uint32_t)w;
        const unsigned long long pm = wave_ballot(((w >> 32) & 3ull) == 2ull);
        if (pm) {
            const int f = __ffsll((long long)pm) - 1;            // nearest predecessor holding a prefix
            excl += wave_sum_u32(lane <= f ? val : 0u);
            break;
        }
        excl += wave_sum_u32(val);
        p -= 64;
    }
    return excl;
}

// Device-wide exclusive scan in ONE launch: a block scans its 4 096-element tile in registers, chains
// its total through `status` (decoupled look-back) and writes.  Replaces reduce / scan-of-sums /
// final (three launches, ~30 us for the 950 k block-histogram counters of a radix pass).
__global__ __launch_bounds__(SCAN_THREADS) void scan_chained_kernel(const uint32_t* __restrict__ in, uint32_t n,
                                                                    uint32_t* __restrict__ out,
                                                                    unsigned long long* status, uint32_t* counter,
                                                                    uint32_t* err, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t s_blk, s_base;
    if (threadIdx.x == 0) s_blk = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t blk = s_blk;
    uint32_t base = blk * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
    if (base + SCAN_ITEMS <= n && ((uintptr_t)in & 15) == 0) {
        const uint4* i4 = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; i++) {
            uint4 q = i4[i];
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) v[i] = (base + i < n) ? in[base + i] : 0;
    }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    uint32_t tot;
    uint32_t ex = block_excl_scan_u32(s, &tot, lds4);
    if (threadIdx.x < 64) {
        uint32_t excl = chain_lookback(status, blk, tot, (int)threadIdx.x, err);
        if (threadIdx.x == 0) {
            s_base = excl;
            if (blk == gridDim.x - 1 && total_out) *total_out = excl + tot;
        }
    }
    __syncthreads();
    ex += s_base;
    if (base + SCAN_ITEMS <= n && ((uintptr_t)out & 15) == 0) {
        uint4* o4 = reinterpret_cast<uint4*>(out + base);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; i++) {
            uint4 q;
            q.x = ex; ex += v[4 * i];
            q.y = ex; ex += v[4 * i + 1];
            q.z = ex; ex += v[4 * i + 2];
            q.w = ex; ex += v[4 * i + 3];
            o4[i] = q;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (base + i < n) out[base + i] = ex;
            ex += v[i];
        }
    }
}

// u32 words of scratch the scan needs for n elements: status[nb] u64 | counter | err (+ slack)
inline size_t scan_tmp_u32(size_t n) { return 2 * (size_t)mi_div_up((long long)n, SCAN_TILE) + 16; }

// exclusive scan; tmp needs scan_tmp_u32(n) u32, 8-byte aligned.  in may equal out.
int scan_exclusive_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* tmp, uint32_t* total_out,
                       hipStream_t st, const char* tag = "scan", bool tmp_is_zero = false) {
    if (n == 0) {
        if (total_out) MI_HIP(hipMemsetAsync(total_out, 0, 4, st));
        return 0;
    }
    uint32_t nb = mi_div_up(n, SCAN_TILE);
    MI_REQUIRE(async_err_ptr(), "scan: no device error word");
    if (!tmp_is_zero) MI_HIP(hipMemsetAsync(tmp, 0, ((size_t)2 * nb + 2) * sizeof(uint32_t), st));
    MI_LAUNCH(tag, scan_chained_kernel, dim3(nb), dim3(SCAN_THREADS), 0, st, in, n, out,
              reinterpret_cast<unsigned long long*>(tmp), tmp + 2 * (size_t)nb, async_err_ptr(), total_out);
    MI_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------- radix sort
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;   // 4096 keys per block
constexpr int RS_WAVE_TILE = RS_TILE / 4;        // 1024 consecutive keys per wave
// the onesweep passes (small, latency-bound sorts) use LARGER tiles than the classic ones: fewer links in
// the look-back chain.  Measured on the 2 M-key depth sort, us per pass: 8 items/thread 42.7, 16: 32.3,
// 32: 30.0, 48: 45.1 (spills)
// Small sorts (the early, low-resolution phase of real training: 10^4 - 10^5 keys) are a handful of tiles whichever size,
// and a thread then walks its 32 rounds for nothing: 8 items per thread up to 512 K keys (measured: S1, 300 k keys, 23.2 -> 17.3 us per pass; S2, 2 M keys, 29 -> 40 us, so not there; S0 cube, 10 k Gaussians: depth-sort
// pass 18.6 -> 12 us).
constexpr int OS_ITEMS_BIG = 32, OS_ITEMS_SMALL = 8;
inline uint32_t os_small_keys() {
    static const uint32_t v = [] { const char* e = getenv("MI3DGS_OS_SMALL_KEYS"); return e ? (uint32_t)atol(e) : (512u << 10); }();
    return v;
}
// (experiments build only: sort without its prefix -- WRONG results, the bound for the look-back's cost)
inline uint32_t os_nolookback() { const char* e = MI_EXPERIMENT_ENV("MI3DGS_OS_NOLOOKBACK"); return (e && e[0] == '1') ? 0x40000000u : 0u; }
inline int os_big_items() {                        // (experiments build, A/B: MI3DGS_OS_BIG_ITEMS=16 halves the big tile)
    static const int v = [] { const char* e = MI_EXPERIMENT_ENV("MI3DGS_OS_BIG_ITEMS"); return (e && atoi(e) == 16) ? 16 : OS_ITEMS_BIG; }();
    return v;
}
inline int os_items_for(uint32_t cap) { return cap <= os_small_keys() ? OS_ITEMS_SMALL : os_big_items(); }
inline uint32_t os_tiles_for(uint32_t cap) { return (uint32_t)mi_div_up(cap, (long long)RS_THREADS * os_items_for(cap)); }
constexpr int OS_MAX_PASSES = 4;
constexpr int OS_GRP = 16;                            // tiles per look-back group
// control words of a onesweep sort: ghist[4][256] | counters[8] err pad[7] | status[tiles][256] u64 | gagg[4][groups][256] u64
inline size_t os_ctl_u32(uint32_t cap) {
    const size_t B = os_tiles_for(cap), G = (B + OS_GRP - 1) / OS_GRP;
    return (size_t)OS_MAX_PASSES * 256 + 16 + (size_t)512 * B + (size_t)512 * OS_MAX_PASSES * G;
}

__device__ __forceinline__ uint32_t live_count(const uint32_t* n_ptr, uint32_t cap) {
    if (!n_ptr) return cap;
    uint32_t n = *n_ptr;
    return n < cap ? n : cap;
}

// Match-any over the wave on the low BITS bits of `digit`: *npeers = lanes holding the same digit (valid lanes only),
// *rank = those among them below this lane.  Per bit: one sign-extending bit-field extract, one compare (the ballot) and
// one three-input bit operation per 32-lane half (peers &= ~(ballot ^ sext(bit))) -- written out on halves because the
// 64-bit "bit ? m : ~m" form compiled to ten VALU instructions per bit, and these kernels are VALU bound on this loop.
template <int BITS>
__device__ __forceinline__ void match_rank(uint32_t digit, bool valid, uint32_t lt_lo, uint32_t lt_hi, uint32_t* rank,
                                           uint32_t* npeers) {
    const unsigned long long vm = wave_ballot(valid);
    uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const uint32_t sx = (uint32_t)((int32_t)(digit << (31 - b)) >> 31);          // all ones where the bit is set
        const unsigned long long m = wave_ballot(sx != 0u);
        plo &= ~((uint32_t)m ^ sx);
        phi &= ~((uint32_t)(m >> 32) ^ sx);
    }
    *rank = __popc(plo & lt_lo) + __popc(phi & lt_hi);
    *npeers = __popc(plo) + __popc(phi);
}

// KT = key type: uint32_t, or uint16_t for the tile sort when every tile id fits 16 bits (the pairs then move 6 bytes
// per pass instead of 8)
template <typename KT>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const KT* __restrict__ keys,
                                                             const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                             int shift, uint32_t mask, uint32_t* __restrict__ hist,
                                                             uint32_t B) {
    __shared__ uint32_t h[256];
    uint32_t n = live_count(n_ptr, cap);
    h[threadIdx.x] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * RS_TILE;
    if (base < n) {
        // 16 CONSECUTIVE keys per thread, one LDS atomic per run of equal digits: on the pass over
        // the high tile bits a Gaussian's neighbouring tiles share the digit, and 64 lanes adding
        // to one or two counters serialise
        uint32_t first = base + threadIdx.x * RS_ITEMS;
        uint32_t k[RS_ITEMS];
        if (first + RS_ITEMS <= n) {
            const uint4* k4 = reinterpret_cast<const uint4*>(keys + first);
            if (sizeof(KT) == 4) {
#pragma unroll
                for (int i = 0; i < RS_ITEMS / 4; i++) {
                    uint4 v = k4[i];
                    k[4 * i] = v.x; k[4 * i + 1] = v.y; k[4 * i + 2] = v.z; k[4 * i + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < RS_ITEMS / 8; i++) {
                    uint4 v = k4[i];
                    k[8 * i] = v.x & 0xFFFFu; k[8 * i + 1] = v.x >> 16; k[8 * i + 2] = v.y & 0xFFFFu; k[8 * i + 3] = v.y >> 16;
                    k[8 * i + 4] = v.z & 0xFFFFu; k[8 * i + 5] = v.z >> 16; k[8 * i + 6] = v.w & 0xFFFFu; k[8 * i + 7] = v.w >> 16;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < RS_ITEMS; i++) k[i] = first + i < n ? keys[first + i] : 0u;
        }
        uint32_t cnt_valid = first >= n ? 0u : (n - first < RS_ITEMS ? n - first : RS_ITEMS);
        uint32_t run_d = (k[0] >> shift) & mask, run = 0;
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) {
            uint32_t d = (k[i] >> shift) & mask;
            if ((uint32_t)i < cnt_valid) {
                if (d != run_d) {
                    atomicAdd(&h[run_d], run);
                    run_d = d;
                    run = 0;
                }
                run++;
            }
        }
        if (run) atomicAdd(&h[run_d], run);
    }
    __syncthreads();
    if (threadIdx.x <= mask) hist[threadIdx.x * B + blockIdx.x] = h[threadIdx.x];      // (rows of digits this pass does not have are never read)
}

// BITS = width of this pass's digit: the match-any ranking below costs one ballot + a 64-bit per-lane select per
// digit bit and key, and the tile sort's passes are 7 and 6 bits wide, not 8 (the kernel is VALU bound on it)
template <int BITS, typename KT>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(
    const KT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, KT* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr, uint32_t cap, int shift, uint32_t mask,
    const uint32_t* __restrict__ hist_scanned, uint32_t B) {
    __shared__ uint32_t cnt[4][256];
    __shared__ uint32_t delta[256];
    __shared__ KT skey[RS_TILE];
    __shared__ uint32_t sval[RS_TILE];
    __shared__ uint32_t lds4[4];
    uint32_t n = live_count(n_ptr, cap);
    uint32_t block_base = blockIdx.x * RS_TILE;
    if (block_base >= n) return;
    int w = threadIdx.x >> 6, lane = lane_id();
#pragma unroll
    for (int i = 0; i < 4; i++) cnt[i][threadIdx.x] = 0;
    __syncthreads();
    uint32_t wbase = block_base + w * RS_WAVE_TILE;
    uint32_t key[RS_ITEMS], val[RS_ITEMS], loc[RS_ITEMS];
    unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // Every global read of the block is issued here, before any of it is needed: keys, values and
    // the block's row of the scanned histogram used to be three dependent round trips (keys ->
    // rank -> histogram -> barrier -> values -> barrier), each several microseconds under load.
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        uint32_t idx = wbase + r * 64 + lane;
        bool valid = idx < n;
        key[r] = valid ? (uint32_t)keys_in[idx] : 0xFFFFFFFFu;
        val[r] = valid ? (vals_in ? vals_in[idx] : idx) : 0u;
    }
    const uint32_t my_global = hist_scanned[threadIdx.x * B + blockIdx.x];     // digit = threadIdx.x
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        uint32_t idx = wbase + r * 64 + lane;
        bool valid = idx < n;
        uint32_t d = (key[r] >> shift) & mask;
        uint32_t rank, npeers;
        match_rank<BITS>(d, valid, (uint32_t)lt_mask, (uint32_t)(lt_mask >> 32), &rank, &npeers);
        uint32_t pre = valid ? cnt[w][d] : 0;
        if (valid && rank == npeers - 1) cnt[w][d] = pre + npeers;
        loc[r] = pre + rank;
    }
    __syncthreads();
    // Stage the block's pairs in LDS in (digit, input order) order, then stream them out:
    // consecutive threads then store consecutive addresses inside each digit run, instead of
    // one scattered 4-byte store per lane (measured 5.3x HBM write amplification before).
    uint32_t tot;
    {
        uint32_t d = threadIdx.x;
        uint32_t c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d], c3 = cnt[3][d];
        uint32_t lstart = block_excl_scan_u32(c0 + c1 + c2 + c3, &tot, lds4);   // first local slot of digit d
        cnt[0][d] = lstart; cnt[1][d] = lstart + c0; cnt[2][d] = lstart + c0 + c1; cnt[3][d] = lstart + c0 + c1 + c2;
        delta[d] = my_global - lstart;                                           // global = local + delta[digit]
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        uint32_t idx = wbase + r * 64 + lane;
        if (idx < n) {
            uint32_t d = (key[r] >> shift) & mask;
            uint32_t slot = cnt[w][d] + loc[r];
            skey[slot] = (KT)key[r];
            sval[slot] = val[r];
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < tot; j += RS_THREADS) {
        uint32_t k = skey[j];
        uint32_t pos = j + delta[(k >> shift) & mask];
        keys_out[pos] = (KT)k;
        vals_out[pos] = sval[j];
    }
}

// ---- onesweep variant: one upfront kernel builds the digit histograms of ALL passes (the
// digits of a key do not depend on its position), then each pass is ONE kernel: tiles chain
// their per-digit prefixes with decoupled look-back instead of a separate histogram kernel plus
// a three-kernel scan.  The 2 M-key depth sort was launch-latency bound at 5 launches per pass.
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, single-word granule form):
//   status[tile][digit] is ONE 64-bit word {epoch:30 | flag:2 | value:32} written and read
//   with relaxed agent-scope atomics (sc1: bypasses the non-coherent L1, served by L2/memory),
//   so there is no payload/flag ordering to get wrong; flag 1 = this tile's own count,
//   2 = inclusive prefix.  Words whose epoch differs from the pass's epoch are "not yet".
//   Tile ids are handed out by an atomic counter, so every predecessor of a tile belongs to a
//   workgroup that has already started: the chain always resolves, whatever the residency.
//   Every spin is bounded; a timeout raises *err and lets the grid drain.
constexpr unsigned OS_SPIN_LIMIT = CHAIN_SPIN_LIMIT;
constexpr int OS_LB = 8;                              // look-back loads in flight per thread

// DROP: keys equal to 0xFFFFFFFF (the depth sort's "culled" sentinel) are not counted; the number of the others
// goes to *n_live_out, and pass 0 (os_pass_kernel<DROP>) leaves them behind, so the later passes -- and
// everything downstream of the sort -- work on the visible splats only.
template <bool DROP, int OS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void os_hist_kernel(const uint32_t* __restrict__ keys,
                                                             const uint32_t* __restrict__ n_ptr, uint32_t cap, int passes,
                                                             int per, int nbits, uint32_t* __restrict__ ghist /*[passes][256]*/,
                                                             uint32_t* __restrict__ n_live_out) {
    __shared__ uint32_t h[OS_MAX_PASSES][256];
    __shared__ uint32_t s_scan4[4];
    uint32_t n = live_count(n_ptr, cap);
    constexpr int OS_TILE = RS_THREADS * OS_ITEMS;
    uint32_t base = blockIdx.x * OS_TILE;
    if (base >= n) return;
#pragma unroll
    for (int p = 0; p < OS_MAX_PASSES; p++) h[p][threadIdx.x] = 0;
    __syncthreads();
    // (all loads first: one at a time, each of the OS_ITEMS rounds waited out an HBM round trip -- 25 us for 1.33 M keys)
    uint32_t kreg[OS_ITEMS];
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
        uint32_t idx = base + i * RS_THREADS + threadIdx.x;
        kreg[i] = idx < n ? keys[idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < OS_ITEMS; i++) {
        uint32_t idx = base + i * RS_THREADS + threadIdx.x;
        if (idx < n) {
            uint32_t k = kreg[i];
            if (DROP && k == 0xFFFFFFFFu) continue;
            int shift = 0;
            for (int p = 0; p < passes; p++) {
                int bits = (shift + per <= nbits) ? per : (nbits - shift);
                atomicAdd(&h[p][(k >> shift) & ((1u << bits) - 1u)], 1u);
                shift += bits;
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < passes; p++) {
        uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&ghist[p * 256 + threadIdx.x], c);
    }
    if (DROP) {
        // the block's live keys = the sum of any one pass's counters.
        // The scan's four-word scratch is its OWN array.  Until round 3 it was h[1]: lane 63 of wave w stored its partial sum
        // into h[1][w] while threads 1..3 of wave 0 could still be on their way to reading h[1][1..3] in the loop above (no
        // barrier in between), so once in a few thousand blocks the GLOBAL histogram of pass 1 got a wave's key count
        // (~2 000) instead of a digit's (~32) for digit 1, 2 or 3: every later digit's output range of that pass moved up,
        // ~2 000 slots kept stale pairs and as many real ones landed beyond the live count -- a depth-sorted list with a
        // few thousand duplicated / missing splats for ONE call (tests/test_gpu_configs.py [garden-0]: n_isect 16 085 932
        // instead of 15 980 980, once in ~15 suite runs; DESIGN.md "the [garden-0] failure").
        uint32_t tot;
        block_excl_scan_u32(h[0][threadIdx.x], &tot, s_scan4);
        if (threadIdx.x == 0 && tot) atomicAdd(n_live_out, tot);
    }
}

__device__ __forceinline__ unsigned long long os_pack(uint32_t epoch, uint32_t flag, uint32_t value) {
    return ((unsigned long long)epoch << 34) | ((unsigned long long)flag << 32) | (unsigned long long)value;
}

#ifdef MI3DGS_OS_STAMPS
// Probe build only (tools/sort_probe.py): wall-clock stamps (100 MHz) of the phases of every tile of the last pass run.
__device__ unsigned long long g_os_stamps[1024][8];
#define OS_STAMP(i) do { if (threadIdx.x == 0 && tile < 1024) g_os_stamps[tile][i] = wall_clock64(); } while (0)
#else
#define OS_STAMP(i) do { } while (0)
#endif

// exclusive scan over the block's first 256 threads' values (the others pass 0), NW waves in the block
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan_nw(uint32_t v, uint32_t* total, uint32_t* lds_nw) {
    int lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan_u32(v);
    if (lane == 63) lds_nw[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {                     // only the first four waves carry values
        uint32_t s = lds_nw[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// One pass = one kernel.  A block of OS_THREADS (256 or 1024) threads takes a tile of OS_THREADS * OS_ITEMS keys:
// every wave ranks its own contiguous OS_ITEMS * 64 keys (ballot match, one LDS counter row per wave), thread d < 256 owns
// digit d (publish, look back, publish), the pairs are regrouped in LDS and leave in runs.
// Measured with wall-clock stamps (tools/sort_probe.py, 1.33 M keys, 163 tiles of 8192 keys, 256 threads x 32 items):
// ranking 9.4 us of a 22.5 us kernel -- 32 dependent rounds per wave with ONE wave per SIMD to hide nothing behind --,
// look-back 4.8 us median / 6.9 max (an agent-scope load of another XCD's entry is ~0.5 us, and a tile near the end
// needs ~10 dependent windows of 16), regroup + stores 4.4 us.  Hence 1024 threads x 8 items (8 rounds per wave, four
// waves per SIMD) and a look-back window of OS_THREADS / 256 x OS_LB = 64 predecessors per round trip.
template <int OS_THREADS, int OS_ITEMS>
struct OsShared {
    static constexpr int NW = OS_THREADS / 64, NG = OS_THREADS / 256, OS_TILE = OS_THREADS * OS_ITEMS;
    uint32_t cnt[NW][256];
    uint32_t delta[256];
    uint32_t skey[OS_TILE], sval[OS_TILE];
    uint32_t lds_nw[NW];
    uint32_t s_tile;
    uint32_t lb_part[NG][256];             // look-back: the thread groups' partial sums
};

// one tile of one pass (the whole block; returns early, as a block, for a tile past the live count)
// COH: the pairs are read and written with agent-scope accesses (sc1: past the XCD's own L2), for the one-launch sort whose
// passes hand their output to blocks on other XCDs without a kernel boundary in between.
template <bool DROP, int OS_THREADS, int OS_ITEMS, bool COH = false>
__device__ __forceinline__ void os_pass_tile(
    OsShared<OS_THREADS, OS_ITEMS>& S,
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr, uint32_t cap, int shift, uint32_t mask,
    const uint32_t* __restrict__ ghist_pass /*[256]*/, unsigned long long* status /*[tiles][256]*/,
    unsigned long long* gagg /*[groups][256], this pass's*/, uint32_t* tile_counter, uint32_t epoch, uint32_t* err) {
    constexpr int NW = OS_THREADS / 64, NG = OS_THREADS / 256;
    constexpr int OS_TILE = OS_THREADS * OS_ITEMS, OS_WAVE_TILE = OS_TILE / NW;
    auto& cnt = S.cnt;
    auto& delta = S.delta;
    auto& skey = S.skey;
    auto& sval = S.sval;
    auto& lds_nw = S.lds_nw;
    auto& s_tile = S.s_tile;
    auto& lb_part = S.lb_part;
    uint32_t n = COH && n_ptr ? min(__hip_atomic_load(n_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), cap) : live_count(n_ptr, cap);
    if (threadIdx.x == 0) s_tile = atomicAdd(tile_counter, 1u);
#pragma unroll
    for (int i = 0; i < NW * 256 / OS_THREADS; i++) (&cnt[0][0])[i * OS_THREADS + threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    uint32_t block_base = tile * OS_TILE;
    if (block_base >= n) return;                      // later tiles are empty too: nobody waits on this one
    OS_STAMP(0);
    int w = threadIdx.x >> 6, lane = lane_id();
    uint32_t wbase = block_base + w * OS_WAVE_TILE;
    uint32_t key[OS_ITEMS], val[OS_ITEMS], loc[OS_ITEMS];
    unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // all global reads up front: the values travel while the ranking and the look-back run
#pragma unroll
    for (int r = 0; r < OS_ITEMS; r++) {
        uint32_t idx = wbase + r * 64 + lane;
        bool valid = idx < n;
        if (COH) {
            key[r] = valid ? __hip_atomic_load(keys_in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
            val[r] = valid ? (vals_in ? __hip_atomic_load(vals_in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : idx) : 0u;
        } else {
            key[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
            val[r] = valid ? (vals_in ? vals_in[idx] : idx) : 0u;
        }
    }
    const int d = threadIdx.x & 255, g = threadIdx.x >> 8;
    // (COH: the histograms were built by atomics of this same launch)
    const uint32_t my_ghist = g == 0 ? (COH ? __hip_atomic_load(ghist_pass + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ghist_pass[d]) : 0u;
#pragma unroll
    for (int r = 0; r < OS_ITEMS; r++) {
        uint32_t idx = wbase + r * 64 + lane;
        bool valid = idx < n && !(DROP && key[r] == 0xFFFFFFFFu);
        uint32_t k = key[r];
        uint32_t dg = (k >> shift) & mask;
        uint32_t rank, npeers;
        match_rank<8>(dg, valid, (uint32_t)lt_mask, (uint32_t)(lt_mask >> 32), &rank, &npeers);
        uint32_t pre = valid ? cnt[w][dg] : 0;
        if (valid && rank == npeers - 1) cnt[w][dg] = pre + npeers;
        loc[r] = pre + rank;
    }
    OS_STAMP(1);
    __syncthreads();
    OS_STAMP(2);
    uint32_t tot;
    {
        // thread d of the first 256 owns digit d: publish the tile's count, look back, publish the inclusive prefix
        uint32_t mine = 0;
        // Look-back in ONE round trip.  Tiles form groups of OS_GRP; a tile publishes its count in status[tile][digit] and
        // adds {1 << 40 | count} to its group's word gagg[group][digit].  Its exclusive prefix is the sum of the counts of
        // the tiles before it in its own group (<= OS_GRP - 1 entries) and of the words of the groups before (complete
        // once OS_GRP tiles have arrived): at most 15 + tiles/16 loads whose addresses are all known up front, spread
        // over the NG thread groups and issued OS_LB at a time.  (The decoupled look-back this replaces -- aggregate /
        // inclusive flags, walking back until an inclusive entry -- cost 4-7 us per pass with 163 tiles resident at once:
        // an agent-scope load of another XCD's word takes ~0.5 us and a late tile needed several dependent windows.)
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < NW; i++) mine += cnt[i][d];
            __hip_atomic_store(status + (size_t)tile * 256 + d, os_pack(epoch, 1u, mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gagg + (size_t)(tile / OS_GRP) * 256 + d, (1ull << 40) | (unsigned long long)mine,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t excl = 0;
        OS_STAMP(3);
        if (tile > 0 && !(epoch & 0x40000000u)) {          // (bit 30 of the epoch: timing experiment, no look-back)
            const int in_grp = (int)(tile % OS_GRP), n_grp = (int)(tile / OS_GRP), E = in_grp + n_grp;
            const uint32_t ep = epoch & 0x3FFFFFFFu;
            uint32_t part = 0;
            for (int e0 = g; e0 < E; e0 += NG * OS_LB) {
                unsigned long long wv[OS_LB];
#pragma unroll
                for (int i = 0; i < OS_LB; i++) {
                    const int e = e0 + i * NG;
                    const unsigned long long* a = e < in_grp ? status + (size_t)(tile - 1 - e) * 256 + d
                                                             : gagg + (size_t)(e - in_grp) * 256 + d;
                    wv[i] = e < E ? __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
#pragma unroll
                for (int i = 0; i < OS_LB; i++) {
                    const int e = e0 + i * NG;
                    if (e >= E) continue;
                    const bool is_tile = e < in_grp;
                    const unsigned long long* a = is_tile ? status + (size_t)(tile - 1 - e) * 256 + d
                                                          : gagg + (size_t)(e - in_grp) * 256 + d;
                    unsigned long long v = wv[i];
                    unsigned spins = 0;
                    while (is_tile ? !((uint32_t)(v >> 34) == ep && ((v >> 32) & 3ull) != 0ull) : (v >> 40) != (unsigned long long)OS_GRP) {
                        if (++spins > OS_SPIN_LIMIT) { atomicOr(err, 1u); v = 0ull; break; }
                        __builtin_amdgcn_s_sleep(1);
                        v = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    part += (uint32_t)v;
                }
            }
            if (NG == 1) {
                excl = part;
            } else {
                lb_part[g][d] = part;
                __syncthreads();
                if (g == 0) {
#pragma unroll
                    for (int gg = 0; gg < NG; gg++) excl += lb_part[gg][d];
                }
            }
        }
        OS_STAMP(4);
        // digit base = exclusive scan of the global histogram; local start = exclusive scan of this tile's counts
        uint32_t t2;
        uint32_t dbase = block_excl_scan_nw<NW>(my_ghist, &t2, lds_nw);
        uint32_t lstart = block_excl_scan_nw<NW>(mine, &tot, lds_nw);
        if (g == 0) {
            uint32_t run = lstart;
#pragma unroll
            for (int i = 0; i < NW; i++) { const uint32_t ci = cnt[i][d]; cnt[i][d] = run; run += ci; }
            delta[d] = dbase + excl - lstart;                             // global = local slot + delta[digit]
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < OS_ITEMS; r++) {
        uint32_t idx = wbase + r * 64 + lane;
        if (idx < n && !(DROP && key[r] == 0xFFFFFFFFu)) {
            uint32_t dg = (key[r] >> shift) & mask;
            uint32_t slot = cnt[w][dg] + loc[r];
            skey[slot] = key[r];
            sval[slot] = val[r];
        }
    }
    OS_STAMP(5);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < tot; j += OS_THREADS) {
        uint32_t k = skey[j];
        uint32_t pos = j + delta[(k >> shift) & mask];
        if (COH) {
            __hip_atomic_store(keys_out + pos, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(vals_out + pos, sval[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            keys_out[pos] = k;
            vals_out[pos] = sval[j];
        }
    }
    OS_STAMP(6);
#ifdef MI3DGS_OS_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    OS_STAMP(7);
#endif
}


template <bool DROP, int OS_THREADS, int OS_ITEMS>
__global__ __launch_bounds__(OS_THREADS) void os_pass_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr, uint32_t cap, int shift, uint32_t mask,
    const uint32_t* __restrict__ ghist_pass, unsigned long long* status, unsigned long long* gagg, uint32_t* tile_counter,
    uint32_t epoch, uint32_t* err) {
    __shared__ OsShared<OS_THREADS, OS_ITEMS> S;
    os_pass_tile<DROP, OS_THREADS, OS_ITEMS>(S, keys_in, vals_in, keys_out, vals_out, n_ptr, cap, shift, mask, ghist_pass, status, gagg,
                                             tile_counter, epoch, err);
}

// ---- all passes of a SMALL sort in one launch.  A 40 k-key depth sort (the reference's own trained scene) is five launches of
// 8 - 10 us each in which 20 blocks do 2 us of work: launch ramp and tail, five times.  Here the blocks of one launch run the
// passes back to back with a device-wide barrier in between.  The barrier needs every block RESIDENT at once, so the launcher only takes this path for grids that fill at most a
// quarter of the device (two processes sharing the card can then not starve each other's blocks), and the wait is bounded like
// every other chain's (bit 1 of the error word).
struct OsFusedArgs {
    uint32_t *key[2], *val[2];             // ping-pong buffers: pass p reads [p & 1], writes [(p + 1) & 1]
    const uint32_t* n_first;               // element count of pass 0 (null: cap)
    const uint32_t* n_later;               // element count of the later passes (the live count when sentinels are dropped)
    uint32_t cap;
    int passes, shift[OS_MAX_PASSES];
    uint32_t mask[OS_MAX_PASSES];
    const uint32_t* ghist;                 // [passes][256]
    unsigned long long *status, *gagg;
    size_t gagg_pass;
    uint32_t* counters;                    // [0..3] tile counters, [4..7] barrier counters
    uint32_t epoch_flags;
    uint32_t* err;
    int identity_vals;
    uint32_t* ghist_w;                     // the same histograms, for the in-kernel histogram phase
    uint32_t* n_live_out;                  // DROP: receives the number of keys that are not sentinels
};

__device__ __forceinline__ void os_grid_barrier(uint32_t* bar, uint32_t nblocks, uint32_t* err) {
    // The pass's pairs were stored with agent-scope (write-through) accesses: once a wave's stores have completed they are
    // where every XCD reads them.  So the barrier is: wait for this wave's stores, block barrier, count the block in, wait for
    // the others.  (First version: release / acquire fences at agent scope, i.e. a write-back and an invalidate of the whole
    // L2 per block and pass -- 60 us for a 40 k-key sort that takes 41 us as five launches.)
    // The wait is spelled out: a workgroup-scope release fence compiles to `s_waitcnt lgkmcnt(0)` only on gfx950 (the CU's
    // waves share an L1, so the compiler owes them nothing more) and a block would count itself in with its sc1 stores and its
    // histogram atomics still in flight.  `make check-isa` greps the ISA for this wait in front of the barrier.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores and returning atomics are complete
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblocks) {
            if (++spins > OS_SPIN_LIMIT) { atomicOr(err, 1u); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// the digit histograms of every pass over this block's tile (what os_hist_kernel does as a launch of its own), and the live count
template <bool DROP, int OS_THREADS, int OS_ITEMS>
__device__ __forceinline__ void os_hist_phase(OsShared<OS_THREADS, OS_ITEMS>& S, const OsFusedArgs& A) {
    constexpr int OS_TILE = OS_THREADS * OS_ITEMS;
    static_assert(OS_THREADS / 64 >= OS_MAX_PASSES, "the per-wave counter rows double as the histogram rows");
    const uint32_t n = live_count(A.n_first, A.cap);
    const uint32_t base = blockIdx.x * OS_TILE;
    for (int i = threadIdx.x; i < OS_MAX_PASSES * 256; i += OS_THREADS) (&S.cnt[0][0])[i] = 0;
    __syncthreads();
    if (base < n) {
        const uint32_t* keys = A.key[0];
        uint32_t kreg[OS_ITEMS];
#pragma unroll
        for (int i = 0; i < OS_ITEMS; i++) {
            const uint32_t idx = base + i * OS_THREADS + threadIdx.x;
            kreg[i] = idx < n ? keys[idx] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int i = 0; i < OS_ITEMS; i++) {
            const uint32_t idx = base + i * OS_THREADS + threadIdx.x;
            if (idx >= n || (DROP && kreg[i] == 0xFFFFFFFFu)) continue;
            for (int p = 0; p < A.passes; p++) atomicAdd(&S.cnt[p][(kreg[i] >> A.shift[p]) & A.mask[p]], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        for (int p = 0; p < A.passes; p++) {
            const uint32_t c = S.cnt[p][threadIdx.x];
            if (c) __hip_atomic_fetch_add(A.ghist_w + p * 256 + threadIdx.x, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (DROP) {                                       // the block's live keys = the sum of any one pass's counters
            const uint32_t tot = wave_sum_all_u32(S.cnt[0][threadIdx.x]);
            if (lane_id() == 0 && tot) __hip_atomic_fetch_add(A.n_live_out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // (the first pass's prologue clears the counter rows again behind its own barrier)
}

template <bool DROP, int OS_THREADS, int OS_ITEMS>
__global__ __launch_bounds__(OS_THREADS) void os_sort_fused_kernel(OsFusedArgs A) {
    __shared__ OsShared<OS_THREADS, OS_ITEMS> S;
    os_hist_phase<DROP, OS_THREADS, OS_ITEMS>(S, A);
    os_grid_barrier(A.counters + 4 + (OS_MAX_PASSES - 1), gridDim.x, A.err);
    for (int p = 0; p < A.passes; p++) {
        const uint32_t* ki = A.key[p & 1];
        const uint32_t* vi = (p == 0 && A.identity_vals) ? nullptr : A.val[p & 1];
        uint32_t* ko = A.key[(p + 1) & 1];
        uint32_t* vo = A.val[(p + 1) & 1];
        const uint32_t ep = (uint32_t)(p + 1) | A.epoch_flags;
        if (p == 0)
            os_pass_tile<DROP, OS_THREADS, OS_ITEMS, true>(S, ki, vi, ko, vo, A.n_first, A.cap, A.shift[0], A.mask[0], A.ghist, A.status, A.gagg,
                                                           A.counters, ep, A.err);
        else
            os_pass_tile<false, OS_THREADS, OS_ITEMS, true>(S, ki, vi, ko, vo, A.n_later, A.cap, A.shift[p], A.mask[p], A.ghist + p * 256, A.status,
                                                      A.gagg + (size_t)p * A.gagg_pass, A.counters + p, ep, A.err);
        if (p + 1 < A.passes) os_grid_barrier(A.counters + 4 + p, gridDim.x, A.err);
    }
}

#ifdef MI3DGS_OS_STAMPS
extern "C" int mi3dgs_debug_read_os_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_os_stamps), bytes < sizeof(g_os_stamps) ? bytes : sizeof(g_os_stamps));
}
#endif

__global__ void set_u32_kernel(uint32_t* p, uint32_t v) { *p = v; }

// 0 = classic (histogram + 3-kernel scan + scatter per pass), 1 = onesweep, 2 = by size (default), 3 = onesweep with one launch
// per pass even where 1 and 2 would run all passes in one launch (small sorts).
// Measured on MI355X: onesweep wins while the sort is launch-latency bound (2 M keys, 4 passes:
// 216 -> 178 us) and loses on large sorts (15 M keys, 2 passes: 350 -> 429 us; the look-back
// chains are dependent sc1 loads served by L2, several hundred cycles each).
int g_sort_mode = 2;
inline uint32_t os_max_keys() {                       // (MI3DGS_OS_MAX_KEYS: A/B of the switch-over to the classic passes)
    static const uint32_t v = [] { const char* e = getenv("MI3DGS_OS_MAX_KEYS"); return e ? (uint32_t)atol(e) : (4u << 20); }();
    return v;
}

inline size_t align_u32(size_t n) { return (n + 63) & ~(size_t)63; }

// words of the pre-cleared control block radix_sort_pairs(..., zeroed) uses for a sort of `cap` keys on `nbits` bits
size_t rs_zero_u32(uint32_t cap, int nbits);

// u32 words of scratch: classic needs the digit-major histogram + its scan scratch, onesweep the
// global histograms, counters, error word and the 64-bit status table
size_t rs_tmp_u32(uint32_t cap) {
    uint32_t B = mi_div_up(cap, RS_TILE);
    size_t hist = (size_t)256 * B;
    size_t classic = hist + scan_tmp_u32(hist);
    size_t onesweep = os_ctl_u32(cap) + 16;
    return classic > onesweep ? classic : onesweep;
}

size_t rs_zero_u32(uint32_t cap, int nbits) {
    const int passes = (nbits + 7) / 8;
    size_t onesweep = os_ctl_u32(cap);
    size_t classic = (size_t)passes * align_u32(scan_tmp_u32((size_t)256 * mi_div_up(cap, RS_TILE)));
    return align_u32(onesweep > classic ? onesweep : classic);
}

// The classic passes (histogram, scan, scatter per digit) on keys of type KT.
template <typename KT>
int rs_classic(KT* keys_a, uint32_t* vals_a, KT* keys_b, uint32_t* vals_b, const uint32_t* n_ptr, uint32_t cap, int nbits,
               uint32_t* tmp, int* result_in_b, hipStream_t st, const char* htag, const char* ctag, const char* s0,
               bool identity_vals, uint32_t* zeroed) {
    *result_in_b = 0;
    if (cap == 0 || nbits <= 0) return 0;
    const uint32_t B = mi_div_up(cap, RS_TILE);
    const int passes = (nbits + 7) / 8;
    const int per = (nbits + passes - 1) / passes;       // the bits spread evenly over the passes (13 bits -> 7 + 6)
    int shift = 0;
    KT *ki = keys_a, *ko = keys_b;
    uint32_t *vi = vals_a, *vo = vals_b;
    uint32_t* hist = tmp;
    uint32_t* scan_tmp = tmp + (size_t)256 * B;
    for (int p = 0; p < passes; p++) {
        int bits = (shift + per <= nbits) ? per : (nbits - shift);
        uint32_t mask = (1u << bits) - 1u;
        MI_LAUNCH(htag, rs_hist_kernel<KT>, dim3(B), dim3(RS_THREADS), 0, st, ki, n_ptr, cap, shift, mask, hist, B);
        // (pre-cleared scan scratch: one region per pass)
        // (digit-major histogram: the digits this pass does not have are zeros at its tail and need no scanning)
        const uint32_t nscan = (mask + 1u) * B;
        int rc = zeroed ? scan_exclusive_u32(hist, hist, nscan, zeroed + (size_t)p * align_u32(scan_tmp_u32((size_t)256 * B)), nullptr, st, s0, true)
                        : scan_exclusive_u32(hist, hist, nscan, scan_tmp, nullptr, st, s0);
        if (rc) return rc;
#define RS_SCATTER(NB) MI_LAUNCH(ctag, (rs_scatter_kernel<NB, KT>), dim3(B), dim3(RS_THREADS), 0, st, ki, (p == 0 && identity_vals) ? nullptr : vi, ko, vo, \
                                 n_ptr, cap, shift, mask, hist, B)
        switch (bits) {
            case 1: case 2: case 3: case 4: RS_SCATTER(4); break;
            case 5: RS_SCATTER(5); break;
            case 6: RS_SCATTER(6); break;
            case 7: RS_SCATTER(7); break;
            default: RS_SCATTER(8); break;
        }
#undef RS_SCATTER
        MI_LAUNCH_CHECK();
        KT* t = ki; ki = ko; ko = t;
        uint32_t* tv = vi; vi = vo; vo = tv;
        shift += bits;
    }
    *result_in_b = passes & 1;
    return 0;
}

// LSD radix sort of (key,val) u32 pairs on bits [0, nbits).  Result ends up in (keys_a, vals_a)
// if the number of passes is even, else in (keys_b, vals_b); returns via *result_in_b.
int radix_sort_pairs(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, const uint32_t* n_ptr,
                     uint32_t cap, int nbits, uint32_t* tmp, int* result_in_b, hipStream_t st, const char* what = "sort",
                     bool identity_vals = false, uint32_t* n_live_out = nullptr, uint32_t* zeroed = nullptr) {
    // zeroed (optional): rs_zero_u32(cap, nbits) words the caller has ALREADY cleared in this stream, together with
    // *n_live_out: the control state (histograms, counters, status tables, scan scratch) then lives there and the sort
    // issues no clear of its own.  A tiny hipMemsetAsync costs ~4.5 us of stream time; the training step had eight of them.
    // n_live_out (device word; n_ptr must be null): keys equal to 0xFFFFFFFF are sentinels.  Onesweep path: they are
    // dropped in pass 0, *n_live_out receives the number of real keys and only those come out (sorted) at the head
    // of the result.  Classic path (large sorts): nothing is dropped, *n_live_out = cap (the sentinels sort last).
    // profiler tags carry the caller's name so the 2M-key depth sort and the I-key tile sort stay apart
    static thread_local char htag_buf[48], ctag_buf[48], s0[48];
    snprintf(htag_buf, sizeof(htag_buf), "rs_hist/%s", what);
    snprintf(ctag_buf, sizeof(ctag_buf), "rs_scatter/%s", what);
    snprintf(s0, sizeof(s0), "scan/%s", what);
    const char* htag = htag_buf;
    const char* ctag = ctag_buf;
    *result_in_b = 0;
    if (cap == 0 || nbits <= 0) return 0;
    int passes = (nbits + 7) / 8;
    // spread the bits evenly over the passes (13 bits -> 7 + 6)
    int per = (nbits + passes - 1) / passes;
    int shift = 0;
    uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    if (g_sort_mode == 1 || g_sort_mode == 3 || (g_sort_mode == 2 && cap <= os_max_keys())) {
        const uint32_t B = os_tiles_for(cap);          // (shadows the classic tile count)
        const bool small = os_items_for(cap) == OS_ITEMS_SMALL;
        // layout: ghist[4][256] | counters[8] err[1] pad[7] | status[B][256] u64
        uint32_t* ctl = zeroed ? zeroed : tmp;
        uint32_t* ghist = ctl;
        uint32_t* counters = ctl + OS_MAX_PASSES * 256;
        uint32_t* err = async_err_ptr();
        MI_REQUIRE(err, "sort: no device error word");
        unsigned long long* status = reinterpret_cast<unsigned long long*>(ctl + OS_MAX_PASSES * 256 + 16);
        // one clear per sort call: histograms, counters, and the status table (epochs 1..passes)
        if (!zeroed) MI_HIP(hipMemsetAsync(tmp, 0, os_ctl_u32(cap) * sizeof(uint32_t), st));
        unsigned long long* gagg = status + (size_t)256 * B;
        const size_t gagg_pass = (size_t)256 * ((B + OS_GRP - 1) / OS_GRP);
        static const bool wide = [] { const char* e = MI_EXPERIMENT_ENV("MI3DGS_OS_THREADS"); return !(e && atoi(e) == 256); }();
        // small sorts: the histograms and every pass in ONE launch (os_sort_fused_kernel).  Its device-wide barriers need all B blocks
        // resident: one 1024-thread block per CU (87 KB of LDS, or 76 VGPRs x 16 waves), 256 slots, and a sort takes at most 3/8 of
        // them, so that two processes sharing the card can not hold each other's blocks out.
        const bool half_tile = !small && os_big_items() == 16;
        const bool fused = g_sort_mode != 3 && wide && !half_tile && passes >= 2 && B <= 96u;
        if (n_live_out) {
            MI_REQUIRE(!n_ptr, "sort: sentinel dropping needs a host-known input size");
            if (!zeroed) MI_HIP(hipMemsetAsync(n_live_out, 0, sizeof(uint32_t), st));
        }
        if (fused) {
        } else if (n_live_out) {
            if (small) MI_LAUNCH(htag, (os_hist_kernel<true, OS_ITEMS_SMALL>), dim3(B), dim3(RS_THREADS), 0, st, ki, n_ptr, cap, passes, per, nbits, ghist, n_live_out);
            else if (os_big_items() == 16) MI_LAUNCH(htag, (os_hist_kernel<true, 16>), dim3(B), dim3(RS_THREADS), 0, st, ki, n_ptr, cap, passes, per, nbits, ghist, n_live_out);
            else MI_LAUNCH(htag, (os_hist_kernel<true, OS_ITEMS_BIG>), dim3(B), dim3(RS_THREADS), 0, st, ki, n_ptr, cap, passes, per, nbits, ghist, n_live_out);
        } else {
            if (small) MI_LAUNCH(htag, (os_hist_kernel<false, OS_ITEMS_SMALL>), dim3(B), dim3(RS_THREADS), 0, st, ki, n_ptr, cap, passes, per, nbits, ghist, n_live_out);
            else if (os_big_items() == 16) MI_LAUNCH(htag, (os_hist_kernel<false, 16>), dim3(B), dim3(RS_THREADS), 0, st, ki, n_ptr, cap, passes, per, nbits, ghist, n_live_out);
            else MI_LAUNCH(htag, (os_hist_kernel<false, OS_ITEMS_BIG>), dim3(B), dim3(RS_THREADS), 0, st, ki, n_ptr, cap, passes, per, nbits, ghist, n_live_out);
        }
        if (fused) {
            OsFusedArgs A;
            A.key[0] = ki; A.key[1] = ko; A.val[0] = vi; A.val[1] = vo;
            A.n_first = n_ptr;                                   // (null with sentinel dropping: the input size is cap)
            A.n_later = n_live_out ? n_live_out : n_ptr;
            A.cap = cap;
            A.passes = passes;
            int sh = 0;
            for (int p = 0; p < passes; p++) {
                const int bits = (sh + per <= nbits) ? per : (nbits - sh);
                A.shift[p] = sh; A.mask[p] = (1u << bits) - 1u;
                sh += bits;
            }
            A.ghist = ghist; A.status = status; A.gagg = gagg; A.gagg_pass = gagg_pass; A.counters = counters;
            A.epoch_flags = os_nolookback();
            A.err = err;
            A.identity_vals = identity_vals ? 1 : 0;
            A.ghist_w = ghist; A.n_live_out = n_live_out;
            const bool drop = n_live_out != nullptr;
            if (drop) { if (small) MI_LAUNCH(ctag, (os_sort_fused_kernel<true, 1024, OS_ITEMS_SMALL / 4>), dim3(B), dim3(1024), 0, st, A);
                        else MI_LAUNCH(ctag, (os_sort_fused_kernel<true, 1024, OS_ITEMS_BIG / 4>), dim3(B), dim3(1024), 0, st, A); }
            else { if (small) MI_LAUNCH(ctag, (os_sort_fused_kernel<false, 1024, OS_ITEMS_SMALL / 4>), dim3(B), dim3(1024), 0, st, A);
                   else MI_LAUNCH(ctag, (os_sort_fused_kernel<false, 1024, OS_ITEMS_BIG / 4>), dim3(B), dim3(1024), 0, st, A); }
            MI_LAUNCH_CHECK();
            *result_in_b = passes & 1;
            return 0;
        }
        for (int p = 0; p < passes; p++) {
            int bits = (shift + per <= nbits) ? per : (nbits - shift);
            uint32_t mask = (1u << bits) - 1u;
            const bool drop = n_live_out && p == 0;
            const uint32_t* vin = (p == 0 && identity_vals) ? nullptr : vi;
            const uint32_t* np = drop ? n_ptr : (n_live_out ? n_live_out : n_ptr);
            const uint32_t ep = (uint32_t)(p + 1) | os_nolookback();
#define OS_PASS(DROP_, T_, I_) MI_LAUNCH(ctag, (os_pass_kernel<DROP_, T_, I_>), dim3(B), dim3(T_), 0, st, ki, vin, ko, vo, np, cap, shift, mask, \
                                         ghist + p * 256, status, gagg + p * gagg_pass, counters + p, ep, err)
            // the same tile (2048 / 8192 keys) either way: 1024 threads x 2 / 8 items, or (A/B) 256 threads x 8 / 32
            if (wide) {
                const bool half = !small && os_big_items() == 16;
                if (drop) { if (small) OS_PASS(true, 1024, OS_ITEMS_SMALL / 4); else if (half) OS_PASS(true, 1024, 4); else OS_PASS(true, 1024, OS_ITEMS_BIG / 4); }
                else { if (small) OS_PASS(false, 1024, OS_ITEMS_SMALL / 4); else if (half) OS_PASS(false, 1024, 4); else OS_PASS(false, 1024, OS_ITEMS_BIG / 4); }
            } else {
                MI_REQUIRE(os_big_items() == OS_ITEMS_BIG, "sort: MI3DGS_OS_THREADS=256 goes with the default tile");
                if (drop) { if (small) OS_PASS(true, 256, OS_ITEMS_SMALL); else OS_PASS(true, 256, OS_ITEMS_BIG); }
                else { if (small) OS_PASS(false, 256, OS_ITEMS_SMALL); else OS_PASS(false, 256, OS_ITEMS_BIG); }
            }
#undef OS_PASS
            uint32_t* t;
            t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
            shift += bits;
        }
        MI_LAUNCH_CHECK();
        *result_in_b = passes & 1;
        return 0;
    }
    if (n_live_out) MI_LAUNCH("set_u32", set_u32_kernel, dim3(1), dim3(1), 0, st, n_live_out, cap);
    return rs_classic<uint32_t>(keys_a, vals_a, keys_b, vals_b, n_ptr, cap, nbits, tmp, result_in_b, st, htag, ctag, s0,
                                identity_vals, zeroed);
}

// The same for 16-bit keys (the tile sort when no tile id needs more): classic passes only.
int radix_sort_pairs_k16(uint16_t* keys_a, uint32_t* vals_a, uint16_t* keys_b, uint32_t* vals_b, const uint32_t* n_ptr,
                         uint32_t cap, int nbits, uint32_t* tmp, int* result_in_b, hipStream_t st, const char* what,
                         uint32_t* zeroed) {
    static thread_local char htag_buf[48], ctag_buf[48], s0[48];
    snprintf(htag_buf, sizeof(htag_buf), "rs_hist/%s", what);
    snprintf(ctag_buf, sizeof(ctag_buf), "rs_scatter/%s", what);
    snprintf(s0, sizeof(s0), "scan/%s", what);
    return rs_classic<uint16_t>(keys_a, vals_a, keys_b, vals_b, n_ptr, cap, nbits, tmp, result_in_b, st, htag_buf, ctag_buf, s0,
                                false, zeroed);
}

// ------------------------------------------------------------------------ tile binning
__device__ __forceinline__ void tile_bbox(float mx, float my, int rx, int ry, int tile_size, int tw, int th, int& x0,
                                          int& y0, int& x1, int& y1) {
    float ts = (float)tile_size;
    x0 = min(max(0, (int)floorf((mx - (float)rx) / ts)), tw);
    y0 = min(max(0, (int)floorf((my - (float)ry) / ts)), th);
    x1 = min(max(0, (int)ceilf((mx + (float)rx) / ts)), tw);
    y1 = min(max(0, (int)ceilf((my + (float)ry) / ts)), th);
}

// ---- exact ("tight") tile culling.  A splat contributes to a pixel only if
// alpha = o exp(-sigma) >= 1/255, i.e. sigma <= tau = ln(255 o): the contributing region is the
// ellipse  A dx^2 + 2 B dx dy + C dy^2 <= 2 tau.  The upstream bins by the ellipse's bounding
// box; here each tile ROW of the box is reduced to the span of tiles whose pixel-centre
// rectangle the ellipse actually reaches (the ellipse is convex, so the span is contiguous).
// Culled (tile, splat) pairs are exactly pairs every pixel of which would have been skipped by
// the rasteriser's alpha test, so renders and gradients are bit-identical; every downstream
// kernel just sees fewer intersections.  Conservative: tau + 1e-3, 0.1 % + 0.02 px on every
// extent, and a fall-back to the plain box row when the stored conic is near-degenerate.
struct SpanGeom {
    float mx, my, A, B, C, tau2;    // tau2 = 2 (tau + margin)
    int x0, y0, x1, y1;             // bounding box in tiles, [x0,x1) x [y0,y1)
    bool exact;
};

__device__ __forceinline__ SpanGeom span_geom(const float* __restrict__ s, int2 r, int tile_size, int tw, int th) {
    SpanGeom g;
    g.mx = s[SP_X]; g.my = s[SP_Y]; g.A = s[SP_CA]; g.B = s[SP_CB]; g.C = s[SP_CC];
    tile_bbox(g.mx, g.my, r.x, r.y, tile_size, tw, th, g.x0, g.y0, g.x1, g.y1);
    float o = s[SP_OPA];
    g.tau2 = 2.f * (__logf(fmaxf(o, 1e-12f) * 255.f) + 1e-3f);
    float det = g.A * g.C - g.B * g.B;
    g.exact = g.A > 0.f && g.C > 0.f && det > 1e-3f * g.A * g.C && g.tau2 > 0.f;
    return g;
}

// tiles [tx0, tx0 + len) of tile row ty that the splat can touch (within its box).  NOT inlined:
// tile_count and tile_emit must execute the very same instructions (fp contraction differs
// between inlining contexts) or their per-row lengths could disagree.  Everything travels in registers (scalars in,
// tx0 | len << 16 out): a struct or an output passed by reference lives on the stack of a non-inlined call, and scratch loads in
// the row loop of the wave-granular emit cost more than the arithmetic.
__device__ __attribute__((noinline)) uint32_t row_span_packed(float mx, float my, float A, float B, float C, float tau2,
                                                              int x0, int x1, int exact, int ty, int tile_size, int H) {
    if (!exact) return (uint32_t)x0 | ((uint32_t)(x1 - x0) << 16);
    const float ts = (float)tile_size;
    float ylo = (float)ty * ts + 0.5f, yhi = fminf((float)ty * ts + ts - 0.5f, (float)H - 0.5f);
    float det = A * C - B * B;
    float Ymax = sqrtf(tau2 * A / det) * 1.001f + 0.02f;
    float lo = fmaxf(my - yhi, -Ymax), hi = fminf(my - ylo, Ymax);       // dy = my - y
    if (!(lo <= hi)) return (uint32_t)x0;
    float dstar = B * sqrtf(tau2 / (det * C));                          // dy of the rightmost point
    float dyR = fminf(fmaxf(dstar, lo), hi), dyL = fminf(fmaxf(-dstar, lo), hi);
    float rA = 1.f / A;
    float eR = (B * dyR + sqrtf(fmaxf(tau2 * A - det * dyR * dyR, 0.f))) * rA;   // px - mx, right end
    float eL = (B * dyL - sqrtf(fmaxf(tau2 * A - det * dyL * dyL, 0.f))) * rA;   // left end
    float xR = mx + eR + fabsf(eR) * 1e-3f + 0.02f;
    float xL = mx + eL - fabsf(eL) * 1e-3f - 0.02f;
    int a = max((int)ceilf((xL - (ts - 0.5f)) / ts), x0);       // pixel centres of tile t: 16t+0.5 .. 16t+15.5
    int b = min((int)floorf((xR - 0.5f) / ts), x1 - 1);
    return (uint32_t)a | ((uint32_t)max(0, b - a + 1) << 16);
}

__device__ __forceinline__ void row_span(const SpanGeom& g, int ty, int tile_size, int H, int& tx0, int& len) {
    const uint32_t r = row_span_packed(g.mx, g.my, g.A, g.B, g.C, g.tau2, g.x0, g.x1, g.exact ? 1 : 0, ty, tile_size, H);
    tx0 = (int)(r & 0xFFFFu);
    len = (int)(r >> 16);
}

// ---- block-level row table for the tight path.  A block owns 256 splats; all their tile rows
// ("items") are spread over the 256 threads, each item's span is computed ONCE, and an
// exclusive prefix over the span lengths lets any thread find "the k-th tile of splat g" with a
// binary search over g's rows.  (A first version walked the rows from the top for every emitted
// key: 656 us instead of 87 for tile_emit.)  Splats whose rows do not fit the pool (more than
// ROW_POOL rows in one block: rare, huge splats) are flagged `slow` and use the row walk.
constexpr int ROW_POOL = 2048;
constexpr int SLOW_BLOCKS = 2048;   // x 4 waves striding over the list of big splats
constexpr int SLOW_WORDS = 16;      // one 64-byte list entry per big splat: SpanGeom (11 words), id, first output position, key base

struct RowTable {
    uint32_t row_base[257];        // exclusive scan of rows per splat, [256] = total
    uint32_t cum[ROW_POOL + 1];    // exclusive prefix of span lengths over the items
    uint16_t tx0[ROW_POOL];        // first tile column of each item's span
    uint32_t scan4[4];
};

// g's own rows = y1 - y0 (0 when culled).  Returns g's tile count.  All 256 threads must call.
__device__ __forceinline__ uint32_t build_row_table(RowTable& T, const SpanGeom* geo, bool visible, int tile_size, int H,
                                                    bool& slow) {
    const int tid = threadIdx.x;
    uint32_t rows = visible ? (uint32_t)(geo[tid].y1 - geo[tid].y0) : 0u;
    uint32_t total;
    uint32_t base = block_excl_scan_u32(rows, &total, T.scan4);
    T.row_base[tid] = base;
    if (tid == 255) T.row_base[256] = total;
    __syncthreads();
    uint32_t R = min(total, (uint32_t)ROW_POOL);
    slow = visible && (base + rows > (uint32_t)ROW_POOL);
    for (uint32_t item = tid; item < R; item += 256) {
        uint32_t lo = 0, hi = 255;                 // largest g with row_base[g] <= item (skips 0-row splats)
#pragma unroll
        for (int it = 0; it < 8; it++) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (T.row_base[mid] <= item) lo = mid; else hi = mid - 1;
        }
        int tx0, len;
        row_span(geo[lo], geo[lo].y0 + (int)(item - T.row_base[lo]), tile_size, H, tx0, len);
        T.tx0[item] = (uint16_t)tx0;
        T.cum[item] = (uint32_t)len;
    }
    __syncthreads();
    // exclusive scan of cum[0..R): 16 consecutive items per thread
    {
        uint32_t v[ROW_POOL / 256], sum = 0;
#pragma unroll
        for (int i = 0; i < ROW_POOL / 256; i++) {
            uint32_t item = tid * (ROW_POOL / 256) + i;
            v[i] = item < R ? T.cum[item] : 0u;
            sum += v[i];
        }
        uint32_t tot;
        uint32_t ex = block_excl_scan_u32(sum, &tot, T.scan4);
#pragma unroll
        for (int i = 0; i < ROW_POOL / 256; i++) {
            uint32_t item = tid * (ROW_POOL / 256) + i;
            if (item < R) T.cum[item] = ex;
            ex += v[i];
        }
        if (tid == 255) T.cum[R] = tot;
    }
    __syncthreads();
    uint32_t n = 0;
    if (visible) {
        if (!slow) n = T.cum[base + rows] - T.cum[base];
        else
            for (int ty = geo[tid].y0; ty < geo[tid].y1; ty++) {
                int tx0, len;
                row_span(geo[tid], ty, tile_size, H, tx0, len);
                n += (uint32_t)len;
            }
    }
    return n;
}

// per (c,n): tiles touched + depth key for the depth sort.  Culled -> 0 tiles, key 0xFFFFFFFF.
template <bool TIGHT>
__global__ __launch_bounds__(256) void tile_count_kernel(uint32_t CN, const int32_t* __restrict__ radii,
                                                         const float* __restrict__ splats, int tile_size, int tw,
                                                         int th, int H, uint32_t* __restrict__ tiles_per_gauss,
                                                         uint32_t* __restrict__ depth_keys,
                                                         uint32_t* __restrict__ ids) {
    __shared__ SpanGeom s_geo[TIGHT ? 256 : 1];
    __shared__ RowTable T[TIGHT ? 1 : 0 + 1];
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < CN;
    int2 r = make_int2(0, 0);
    if (live) r = *reinterpret_cast<const int2*>(radii + 2 * (size_t)idx);
    const bool vis = live && r.x > 0 && r.y > 0;
    uint32_t tiles = 0, key = 0xFFFFFFFFu;
    const float* s = splats + (size_t)(live ? idx : 0) * SPLAT_STRIDE;
    if (TIGHT) {
        if (vis) s_geo[threadIdx.x] = span_geom(s, r, tile_size, tw, th);
        __syncthreads();
        bool slow;
        tiles = build_row_table(T[0], s_geo, vis, tile_size, H, slow);
    } else if (vis) {
        int x0, y0, x1, y1;
        tile_bbox(s[SP_X], s[SP_Y], r.x, r.y, tile_size, tw, th, x0, y0, x1, y1);
        tiles = (uint32_t)((x1 - x0) * (y1 - y0));
    }
    if (!live) return;
    if (vis) key = __float_as_uint(s[SP_DEPTH]);   // depth > 0 => bit pattern is order preserving
    tiles_per_gauss[idx] = tiles;
    depth_keys[idx] = key;
    ids[idx] = idx;
}

__global__ __launch_bounds__(256) void gather_u32_kernel(uint32_t n, const uint32_t* __restrict__ src,
                                                         const uint32_t* __restrict__ index,
                                                         uint32_t* __restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

// One block per 256 consecutive depth-sorted Gaussians.  Their output range is contiguous
// ([cum[first], cum[last] + tiles[last])), so the block walks it with consecutive threads on
// consecutive slots (coalesced 1-KiB stores) and finds each slot's Gaussian by binary search
// over the 256 exclusive offsets held in LDS.  (Thread-per-Gaussian emission measured 3.5x
// HBM write amplification.)
// sort input of the fused path (mi3dgs_bin_tiles): depth key + identity, culled splats last
__global__ __launch_bounds__(256) void depth_keys_kernel(uint32_t CN, const int32_t* __restrict__ radii,
                                                         const float* __restrict__ splats,
                                                         uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ ids) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= CN) return;
    int2 r = *reinterpret_cast<const int2*>(radii + 2 * (size_t)idx);
    uint32_t key = 0xFFFFFFFFu;
    if (r.x > 0 && r.y > 0) key = __float_as_uint(splats[(size_t)idx * SPLAT_STRIDE + SP_DEPTH]);
    depth_keys[idx] = key;
    ids[idx] = idx;
}

// CHAINED = false: offsets come from `cum` (mi3dgs_bin_count ran tile_count + gather + scan before).
// CHAINED = true : count and emit in ONE pass over the depth-sorted splats: the block builds its row
// table once, chains its total through `status`, and emits; tile_count, the gather and the
// three-kernel scan (and the second row-table build) disappear.  The last block leaves the
// total in *n_isect_out.
template <bool TIGHT, bool CHAINED>
__global__ __launch_bounds__(256) void tile_emit_kernel(uint32_t CN_cap, const uint32_t* __restrict__ n_sorted_ptr, bool radii_in_records,
                                                        uint32_t N, const uint32_t* __restrict__ sorted_ids,
                                                        const uint32_t* __restrict__ cum,
                                                        const int32_t* __restrict__ radii,
                                                        const float* __restrict__ splats, int tile_size, int tw, int th,
                                                        int H, uint32_t cap, uint32_t* __restrict__ tile_keys,
                                                        uint32_t* __restrict__ flat_ids,
                                                        unsigned long long* status, uint32_t* chain_counter,
                                                        uint32_t* chain_err, uint32_t* __restrict__ n_isect_out,
                                                        uint32_t* __restrict__ tiles_out, uint32_t* __restrict__ slow_list,
                                                        uint32_t* slow_count) {
    __shared__ uint32_t s_cum[257];
    __shared__ uint32_t s_id[256], s_key0[256];
    __shared__ int s_w[256];
    __shared__ uint8_t s_slow[256];
    __shared__ SpanGeom s_geo[TIGHT ? 256 : 1];
    __shared__ RowTable T[1];
    __shared__ uint32_t s_blk, s_base, s_scan4[4];
    uint32_t blk = blockIdx.x;
    if (CHAINED) {
        if (threadIdx.x == 0) s_blk = atomicAdd(chain_counter, 1u);
        __syncthreads();
        blk = s_blk;
    }
    // the sorted list may be shorter than its capacity (the depth sort dropped the culled splats); blocks past
    // its end have nothing to do, and nobody waits on them: block ids are handed out in order
    const uint32_t CN = live_count(n_sorted_ptr, CN_cap);
    if (blk * blockDim.x >= CN) return;
    uint32_t i = blk * blockDim.x + threadIdx.x;
    uint32_t my_cum = 0, my_n = 0, idx = 0, key0 = 0;
    int w = 1;
    bool vis = false;
    if (i < CN) {
        idx = sorted_ids[i];
        if (!CHAINED) my_cum = cum[i];
        const float* sp = splats + (size_t)idx * SPLAT_STRIDE;
        // ONE gather per splat when the radii travel inside the record (project_fwd writes them there); the
        // separate radii array costs a second random 64-byte sector per splat for 8 bytes
        int2 r = radii_in_records ? make_int2(__float_as_int(sp[SP_RX]), __float_as_int(sp[SP_RY]))
                                  : *reinterpret_cast<const int2*>(radii + 2 * (size_t)idx);
        if (r.x > 0 && r.y > 0) {
            vis = true;
            if (TIGHT) {
                s_geo[threadIdx.x] = span_geom(sp, r, tile_size, tw, th);
                key0 = (idx / N) * (uint32_t)(tw * th);
            } else {
                int x0, y0, x1, y1;
                tile_bbox(sp[SP_X], sp[SP_Y], r.x, r.y, tile_size, tw, th, x0, y0, x1, y1);
                w = max(x1 - x0, 1);
                my_n = (uint32_t)((x1 - x0) * (y1 - y0));
                key0 = (idx / N) * (uint32_t)(tw * th) + (uint32_t)(y0 * tw + x0);
            }
        }
    }
    if (TIGHT) {
        __syncthreads();
        bool slow;
        my_n = build_row_table(T[0], s_geo, vis, tile_size, H, slow);   // same arithmetic as tile_count_kernel<true>
        s_slow[threadIdx.x] = slow ? 1 : 0;
    }
    if (CHAINED) {
        uint32_t total;
        uint32_t local = block_excl_scan_u32(my_n, &total, s_scan4);
        if (threadIdx.x < 64) {
            uint32_t excl = chain_lookback(status, blk, total, (int)threadIdx.x, chain_err);
            if (threadIdx.x == 0) {
                s_base = excl;
                if (blk == (CN - 1) / blockDim.x) {
                    // the count every later kernel (sort, offsets, rasterisers) reads is clamped to the
                    // capacity of the buffers; an overflow is reported through the sticky error word (bit 2)
                    uint32_t tot = excl + total;
                    if (tot > cap) { atomicOr(chain_err, 4u); tot = cap; }
                    *n_isect_out = tot;
                }
            }
        }
        __syncthreads();
        my_cum = s_base + local;
        if (tiles_out && i < CN) tiles_out[idx] = my_n;
    }
    if (TIGHT) {
        // Splats with more tile rows than the block's row table holds (big splats: few Gaussians at a
        // high resolution, early training) go on a list and are emitted by tile_emit_slow_kernel, one
        // wave per splat across the whole GPU.  The entry carries the span geometry itself, so that the
        // rows come out exactly as they were counted here.
        bool slow_me = s_slow[threadIdx.x] != 0;
        unsigned long long sm = wave_ballot(slow_me);
        if (sm) {
            uint32_t basepos = 0;
            int leader = (int)__builtin_ctzll(sm);
            if ((int)lane_id() == leader) basepos = atomicAdd(slow_count, (uint32_t)__popcll(sm));
            basepos = __shfl(basepos, leader, 64);
            if (slow_me) {
                uint32_t* e = slow_list + (size_t)(basepos + (uint32_t)__popcll(sm & ((1ull << lane_id()) - 1ull))) * SLOW_WORDS;
                const SpanGeom gg = s_geo[threadIdx.x];
                e[0] = __float_as_uint(gg.mx); e[1] = __float_as_uint(gg.my); e[2] = __float_as_uint(gg.A);
                e[3] = __float_as_uint(gg.B); e[4] = __float_as_uint(gg.C); e[5] = __float_as_uint(gg.tau2);
                e[6] = (uint32_t)gg.x0; e[7] = (uint32_t)gg.y0; e[8] = (uint32_t)gg.x1; e[9] = (uint32_t)gg.y1;
                e[10] = gg.exact ? 1u : 0u; e[11] = idx; e[12] = my_cum; e[13] = key0;
            }
        }
    }
    s_cum[threadIdx.x] = my_cum;
    s_id[threadIdx.x] = idx;
    s_key0[threadIdx.x] = key0;
    s_w[threadIdx.x] = w;
    // the last live thread of the block publishes the end of the block's range
    uint32_t last = min(CN - blk * blockDim.x, blockDim.x) - 1;
    if (threadIdx.x == last) s_cum[256] = my_cum + my_n;
    __syncthreads();
    uint32_t begin = s_cum[0], end = s_cum[256];
    if (threadIdx.x > last) s_cum[threadIdx.x] = end;     // padding so the search never lands on a dead slot
    __syncthreads();
    for (uint32_t p = begin + threadIdx.x; p < end; p += 256) {
        // largest g with s_cum[g] <= p
        uint32_t lo = 0, hi = 255;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            uint32_t mid = (lo + hi + 1) >> 1;
            if (s_cum[mid] <= p) lo = mid; else hi = mid - 1;
        }
        uint32_t local = p - s_cum[lo];
        uint32_t key;
        if (TIGHT) {
            const int y0 = s_geo[lo].y0;
            if (!s_slow[lo]) {
                // largest row item of splat `lo` whose exclusive prefix <= local
                uint32_t b0 = T[0].row_base[lo], b1 = T[0].row_base[lo + 1] - 1;
                uint32_t c0 = T[0].cum[b0];
                uint32_t a = b0, b = b1;
                while (a < b) {
                    uint32_t mid = (a + b + 1) >> 1;
                    if (T[0].cum[mid] - c0 <= local) a = mid; else b = mid - 1;
                }
                key = s_key0[lo] + (uint32_t)((y0 + (int)(a - b0)) * tw) + (uint32_t)T[0].tx0[a] + (local - (T[0].cum[a] - c0));
            } else {
                continue;           // splats whose rows did not fit the table: emitted wave-per-splat below
            }
        } else {
            int ww = s_w[lo];
            uint32_t row = local / (uint32_t)ww;
            uint32_t col = local - row * (uint32_t)ww;
            key = s_key0[lo] + row * (uint32_t)tw + col;
        }
        if (p < cap) {
            tile_keys[p] = key;
            flat_ids[p] = s_id[lo];
        }
    }
}

// ---- wave-granular chained count + emit for the exact ("tight") path: what the training step and the renderer run.
// The block-cooperative kernel above spends ~70 % of its wave cycles waiting (SQ_WAIT_ANY): a dozen barriers per block, two
// dependent binary searches in LDS per emitted key (8 + ~3 steps), 29 KB of LDS = 5 blocks per CU.  Here a WAVE owns 64
// consecutive depth-sorted splats; nothing is shared between waves but the look-back chain (one status word per wave) and the
// block's ticket.  The unit of work is the tile ROW of a splat ("item"; items in splat-major order ARE the emission order):
//   count  lane = item, 64 at a time: owner splat by a 6-step search over the wave's row prefixes, span by row_span_packed
//          (the same non-inlined arithmetic as every other path); the results of the first WE_CACHE chunks stay in registers;
//   chain  the wave's total goes through the decoupled look-back;
//   emit   per 64 items: a wave scan gives every item its first output slot; items with keys are compacted into a 64-entry
//          table and set ONE bit per item in a bitmap of the chunk's output slots (ds_or_b64); every 64 outputs are then
//          resolved with one broadcast read of the bitmap word + a popcount (rank = number of item starts at or before the
//          slot), no search, and stored coalesced.
// There is no row pool and therefore no slow list: a 68-row splat is just 68 items.  No barrier after the ticket, 4.6 KB of LDS
// per wave.  A first wave-granular version with lane = splat (8 rows in registers, taller splats on the slow list) needed
// 9.7e7 VALU instructions (the block kernel: 5.3e7) and 316 + 24 us; kept out of the tree.
// Same output, bit for bit, as the two-phase path: tests/test_gpu_parity.py::test_fused_binning_equals_the_two_phase_path,
// tests/test_gpu_configs.py (15 - 40 M keys).
// Waves per block = 64-splat chunks per ticket.  The kernel needs 106 scalar registers, i.e. at most SEVEN waves per SIMD
// (800 SGPRs per SIMD in blocks of 16): a 16-wave block takes four per SIMD, so only ONE such block fitted a CU although two
// fit its LDS and vector registers -- the phase stamps (tools/emit_probe.py) showed generations of ~300 blocks, 28 us each,
// five of them.  8-wave blocks: three per CU (24 waves; 12-wave blocks the same), 152 -> 127 us; 6- and 4-wave blocks lose
// again to the ticket word (145, 151 us).  Capping the scalar registers for eight waves per SIMD (four blocks per CU, 39
// SGPR spills) did not help: 132 us.
constexpr int WE_WAVES = 8;
constexpr int WE_CACHE = 6;               // item chunks whose spans stay in registers between count and emit (384 rows)
constexpr int WE_GROUPS = 128;            // bitmap words per item chunk: 64 items x at most 128 tile columns / 64 (images up to 2 048 px wide; wider ones take the block kernel)
struct WaveEmitLds {
    float geo[64][6];                     // mx my A B C tau2
    uint32_t box[64];                     // x0 | x1 << 16 (tile columns of the bounding box)
    uint32_t flags[64];                   // y0 | exact << 31
    uint32_t row_base[65];                // exclusive prefix of the splats' row counts; [64] = the wave's items
    uint32_t key0[64];                    // camera * tiles + y0 * tw
    uint32_t id[64];
    uint32_t cnt[64];                     // keys per splat (tiles_per_gauss, on request)
    uint32_t it_start[64], it_key[64], it_id[64];       // compacted items of the current chunk
    unsigned long long bitmap[WE_GROUPS];
};

// SPW = splats per wave: 64, or 16 for small scenes (a wave's items are walked 64 at a time, serially: at 20 k big Gaussians
// and 1080p a 64-splat wave has thousands of tile rows and the whole launch is 313 waves).  The switch-over is 256 K Gaussians:
// 16 per wave measured 48 -> 32 us at 175 k real splats (wolf, 960 x 720, round 4) and 44 -> 79 us at 300 k small ones (S1, round 2).
inline uint32_t we_small_splats() {
    static const uint32_t v = [] { const char* e = getenv("MI3DGS_EMIT_SMALL_SPLATS"); return e ? (uint32_t)atol(e) : (256u << 10); }();
    return v;
}
inline int we_spw_for(uint32_t CN) {
    static const int big = [] { const char* e = MI_EXPERIMENT_ENV("MI3DGS_EMIT_SPW"); int v = e ? atoi(e) : 64; return (v == 16 || v == 32) ? v : 64; }();
    return CN <= we_small_splats() ? 16 : big;
}
inline size_t we_chain_entries(uint32_t CN) { return (size_t)WE_WAVES * (size_t)mi_div_up(CN, (long long)we_spw_for(CN) * WE_WAVES); }

#ifdef MI3DGS_OS_STAMPS
// Probe build only (tools/emit_probe.py): wall-clock stamps (100 MHz) of the phases of waves 0 and 15 of every block.
__device__ unsigned long long g_we_stamps[4096][2][8];
#define WE_STAMP(i) do { if (lane == 0 && (wv == 0 || wv == WE_WAVES - 1) && s_blk < 4096) g_we_stamps[s_blk][wv ? 1 : 0][i] = wall_clock64(); } while (0)
extern "C" int mi3dgs_debug_read_we_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_we_stamps), bytes < sizeof(g_we_stamps) ? bytes : sizeof(g_we_stamps));
}
#else
#define WE_STAMP(i) do { } while (0)
#endif

template <int SPW, typename KT>
__global__ __launch_bounds__(64 * WE_WAVES) void tile_emit_wave_kernel(
    uint32_t CN_cap, const uint32_t* __restrict__ n_sorted_ptr, bool radii_in_records, uint32_t N,
    const uint32_t* __restrict__ sorted_ids, const int32_t* __restrict__ radii, const float* __restrict__ splats, int tile_size,
    int tw, int th, int H, uint32_t cap, KT* __restrict__ tile_keys, uint32_t* __restrict__ flat_ids,
    unsigned long long* status, uint32_t* chain_counter, uint32_t* chain_err, uint32_t* __restrict__ n_isect_out,
    uint32_t* __restrict__ tiles_out) {
    __shared__ WaveEmitLds Lw[WE_WAVES];
    __shared__ uint32_t s_blk;
    WaveEmitLds& L = Lw[threadIdx.x >> 6];
    const int lane = lane_id();
    // ONE ticket per block of 16 waves = 1 024 splats: a returning atomic on one word saturates near 88 per microsecond
    // (MI355X_MICROARCH, "dequeue"); a ticket per wave (31 k of them) cost 770 us.  Wave w of ticket b is chain entry
    // 16 b + w; all 16 run concurrently, so every predecessor of a chain entry belongs to a block that has started and
    // publishes its aggregate without waiting for anybody.  (Several chunks per wave do NOT work: a wave could publish the
    // aggregate of its second chunk only after its first chunk's look-back, which serialises the whole chain: 115 ms.)
    // The chain is TWO-LEVEL.  All 8 192 waves the GPU holds start together, so with one chain entry per wave a wave's look-back
    // never met an inclusive prefix (those appear behind its walking front): wave w walked all w / 64 windows.  The chain
    // entry is the BLOCK (16 waves) instead (S1 45 -> 40 us, 20 k big Gaussians at 1080p 55 -> 51 us, S2 unchanged at 157 us:
    // running S2 without any chain takes 93 us, but that also turns its 139 MB of output into overwrites of one small region,
    // and walking the list from its far end, so that nobody waits for the big near splats, did not help either: 172 us):
    // the waves of a block exchange their totals through LDS (they run concurrently), the wave that arrives last publishes the
    // block's total, wave 0 looks back over the blocks (64 of them = 1 024 waves per step), hands the block's exclusive prefix to
    // its fifteen sisters through LDS and publishes the block's inclusive prefix.
    __shared__ uint32_t s_tot[WE_WAVES];
    __shared__ uint32_t s_mask, s_excl, s_excl_ready;
    if (threadIdx.x == 0) { s_blk = atomicAdd(chain_counter, 1u); s_mask = 0u; s_excl = 0u; s_excl_ready = 0u; }
    __syncthreads();
    const uint32_t CN = live_count(n_sorted_ptr, CN_cap);
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t wid = s_blk * (uint32_t)WE_WAVES + wv;
    // a wave arrives with its total; the one that completes the block publishes the block's total for the blocks after it
    // (block 0 has no predecessors: that IS its inclusive prefix)
    auto arrive = [&](uint32_t total_) {
        constexpr uint32_t FULL_ = (1u << WE_WAVES) - 1u;
        uint32_t old_mask = 0u;
        if (lane == 0) {
            s_tot[wv] = total_;
            old_mask = __hip_atomic_fetch_or(&s_mask, 1u << wv, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        old_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)old_mask);
        if (status && (old_mask | (1u << wv)) == FULL_) {
            const uint32_t bt = wave_sum_u32(lane < WE_WAVES ? s_tot[lane] : 0u);
            // atomic MAX, not a store: this word has a second writer (wave 0's inclusive prefix, flag 2, below), a different
            // wave whose store is not ordered against this one.  With the flag in the top bits max() makes the word
            // monotonic -- 0 -> total -> inclusive, whichever write lands first -- so a reader can never find an entry
            // going back from "inclusive" to "total" (either value is correct on its own; the max removes the case
            // distinction from the audit of VERDICT r2 #1)
            if (lane == 0)
                __hip_atomic_fetch_max(status + s_blk, chain_pack(s_blk == 0 ? 2u : 1u, bt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (wid * (uint32_t)SPW >= CN) {          // past the end of the sorted list: an empty member of its block
        if (CN == 0u && wid == 0u && lane == 0) *n_isect_out = 0u;     // nothing visible: the count is written here (no clear in front of the kernel)
        arrive(0u);
        return;
    }
    WE_STAMP(0);
    const uint32_t i = wid * (uint32_t)SPW + (uint32_t)lane;
    const bool owns = lane < SPW && i < CN;   // this lane brings in a splat
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t idx = 0, rows = 0;
    {
        SpanGeom geo;
        geo.mx = geo.my = geo.A = geo.B = geo.C = geo.tau2 = 0.f;
        geo.x0 = geo.y0 = geo.x1 = geo.y1 = 0;
        geo.exact = false;
        uint32_t kc = 0;
        if (owns) {
            idx = sorted_ids[i];
            const float* sp = splats + (size_t)idx * SPLAT_STRIDE;
            // the geometry is requested together with the radii, not behind the test of them: the record is one 64-byte line,
            // and as "radii -> wait -> branch -> geometry -> wait" the gather was three dependent round trips instead of two
            // (tools: the ISA scan of round 3, serial load -> s_waitcnt vmcnt(0) chains)
            const float4 g0 = *reinterpret_cast<const float4*>(sp);              // x y A B
            const float2 g1 = *reinterpret_cast<const float2*>(sp + SP_CC);      // C o
            int2 r = radii_in_records ? make_int2(__float_as_int(sp[SP_RX]), __float_as_int(sp[SP_RY]))
                                      : *reinterpret_cast<const int2*>(radii + 2 * (size_t)idx);
            asm volatile("" :: "v"(g0.x), "v"(g1.x), "v"(r.x));          // (keeps the three loads in front of the branch: hipcc sinks them into it)
            if (r.x > 0 && r.y > 0) {
                const float rec6[6] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y};      // SP_X .. SP_OPA
                geo = span_geom(rec6, r, tile_size, tw, th);
                rows = (uint32_t)(geo.y1 - geo.y0);
                kc = (idx / N) * (uint32_t)(tw * th);
            }
        }
        L.geo[lane][0] = geo.mx; L.geo[lane][1] = geo.my; L.geo[lane][2] = geo.A;
        L.geo[lane][3] = geo.B; L.geo[lane][4] = geo.C; L.geo[lane][5] = geo.tau2;
        L.box[lane] = (uint32_t)geo.x0 | ((uint32_t)geo.x1 << 16);
        L.flags[lane] = (uint32_t)geo.y0 | (geo.exact ? 0x80000000u : 0u);
        L.key0[lane] = kc + (uint32_t)(geo.y0 * tw);
        L.id[lane] = idx;
        L.cnt[lane] = 0u;
    }
    WE_STAMP(1);                              // records gathered, geometry in LDS
    const uint32_t rincl = wave_incl_scan_u32(rows);
    L.row_base[lane] = rincl - rows;
    const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)rincl, 63);
    if (lane == 63) L.row_base[SPW] = R;      // (lanes >= SPW carry no rows: their prefix is R as well)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // same-wave LDS hand-off
    __builtin_amdgcn_wave_barrier();

    // item j -> (owner splat g, tx0 | len << 16)
    auto eval_item = [&](uint32_t j, uint32_t& g, uint32_t& span) {
        uint32_t lo = 0, hi = SPW - 1;             // largest g with row_base[g] <= j (splats without rows share a prefix with their successor)
#pragma unroll
        for (int it = 0; it < (SPW == 64 ? 6 : SPW == 32 ? 5 : 4); it++) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (L.row_base[mid] <= j) lo = mid; else hi = mid - 1;
        }
        g = lo;
        const uint32_t fl = L.flags[lo], bx = L.box[lo];
        const int ty = (int)(fl & 0x7FFFFFFFu) + (int)(j - L.row_base[lo]);
        span = row_span_packed(L.geo[lo][0], L.geo[lo][1], L.geo[lo][2], L.geo[lo][3], L.geo[lo][4], L.geo[lo][5],
                               (int)(bx & 0xFFFFu), (int)(bx >> 16), (int)(fl >> 31), ty, tile_size, H);
    };

    // ---- count
    uint32_t c_g[WE_CACHE], c_span[WE_CACHE];
    uint32_t mine = 0;
#pragma unroll
    for (int c = 0; c < WE_CACHE; c++) {
        c_g[c] = 0; c_span[c] = 0;
        const uint32_t j = (uint32_t)c * 64u + (uint32_t)lane;
        if ((uint32_t)c * 64u < R) {               // wave-uniform
            if (j < R) eval_item(j, c_g[c], c_span[c]);
            mine += c_span[c] >> 16;
            if (tiles_out && (c_span[c] >> 16)) atomicAdd(&L.cnt[c_g[c]], c_span[c] >> 16);
        }
    }
    for (uint32_t j0 = (uint32_t)WE_CACHE * 64u; j0 < R; j0 += 64u) {
        uint32_t g = 0, span = 0;
        if (j0 + lane < R) eval_item(j0 + (uint32_t)lane, g, span);
        mine += span >> 16;
        if (tiles_out && (span >> 16)) atomicAdd(&L.cnt[g], span >> 16);
    }
    const uint32_t total = wave_sum_u32(mine);
    WE_STAMP(2);                              // counted
    // ---- chain
    uint32_t base = 0u;
    if (status) {
        constexpr uint32_t FULL = (1u << WE_WAVES) - 1u;
        const uint32_t b = s_blk;
        arrive(total);
        unsigned spins = 0;
        if (wv == 0) {
            const uint32_t excl_b = b > 0 ? chain_prefix_before(status, b, lane, chain_err) : 0u;
            if (lane == 0) {
                s_excl = excl_b;
                __hip_atomic_store(&s_excl_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (b > 0) {
                while (__hip_atomic_load(&s_mask, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != FULL) {
                    if (++spins > CHAIN_SPIN_LIMIT) { if (lane == 0) atomicOr(chain_err, 2u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const uint32_t bt = wave_sum_u32(lane < WE_WAVES ? s_tot[lane] : 0u);
                if (lane == 0)
                    __hip_atomic_fetch_max(status + b, chain_pack(2u, excl_b + bt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            base = excl_b;
        } else {
            // the totals of the waves before this one in the block, then the block's own prefix from wave 0
            const uint32_t need = (1u << wv) - 1u;
            while ((__hip_atomic_load(&s_mask, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & need) != need) {
                if (++spins > CHAIN_SPIN_LIMIT) { if (lane == 0) atomicOr(chain_err, 2u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            const uint32_t part = wave_sum_u32((uint32_t)lane < wv ? s_tot[lane] : 0u);
            while (__hip_atomic_load(&s_excl_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                if (++spins > CHAIN_SPIN_LIMIT) { if (lane == 0) atomicOr(chain_err, 2u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            base = s_excl + part;
        }
    }
    WE_STAMP(3);                              // base known
    if (wid == (CN - 1u) / (uint32_t)SPW && lane == 0) {
        uint32_t tot = base + total;
        if (tot > cap) { atomicOr(chain_err, 4u); tot = cap; }
        *n_isect_out = tot;
    }
    if (tiles_out) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (owns) tiles_out[idx] = L.cnt[lane];
    }
    // ---- emit
    uint32_t running = 0;                            // outputs of the chunks before this one (wave-local)
    auto emit_chunk = [&](uint32_t j, uint32_t g, uint32_t span) {
        const uint32_t len = span >> 16;
        const uint32_t incl = wave_incl_scan_u32(len);
        const uint32_t Tc = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (Tc == 0u) return;
        const uint32_t start = incl - len;           // chunk-local first slot of this item
        const uint32_t ng = (Tc + 63u) >> 6;
        for (uint32_t w = (uint32_t)lane; w < ng; w += 64u) L.bitmap[w] = 0ull;
        const bool has = len != 0u;
        const unsigned long long hb = wave_ballot(has);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (has) {
            const uint32_t ci = (uint32_t)__popcll(hb & lt_mask);
            const uint32_t row = j - L.row_base[g];
            L.it_start[ci] = start;
            L.it_key[ci] = L.key0[g] + row * (uint32_t)tw + (span & 0xFFFFu);
            L.it_id[ci] = L.id[g];
            atomicOr(&L.bitmap[start >> 6], 1ull << (start & 63u));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t before = 0;                         // item starts in the bitmap words already walked (wave-uniform)
        const unsigned long long le_mask = lt_mask | (1ull << lane);
        for (uint32_t w = 0; w < ng; w++) {
            const unsigned long long m = L.bitmap[w];
            const uint32_t q = (w << 6) + (uint32_t)lane;
            if (q < Tc) {
                const uint32_t rank = before + (uint32_t)__popcll(m & le_mask) - 1u;     // slot 0 of the chunk starts an item: rank >= 0
                const uint32_t p = base + running + q;
                if (p < cap) {
                    tile_keys[p] = (KT)(L.it_key[rank] + (q - L.it_start[rank]));
                    flat_ids[p] = L.it_id[rank];
                }
            }
            before += (uint32_t)__popcll(m);
        }
        running += Tc;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // the tables are rewritten by the next chunk
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll
    for (int c = 0; c < WE_CACHE; c++)
        if ((uint32_t)c * 64u < R) emit_chunk((uint32_t)c * 64u + (uint32_t)lane, c_g[c], c_span[c]);
    for (uint32_t j0 = (uint32_t)WE_CACHE * 64u; j0 < R; j0 += 64u) {
        uint32_t g = 0, span = 0;
        if (j0 + lane < R) eval_item(j0 + (uint32_t)lane, g, span);
        emit_chunk(j0 + (uint32_t)lane, g, span);
    }
    WE_STAMP(4);                              // stores issued
#ifdef MI3DGS_OS_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    WE_STAMP(5);
    if (lane == 0 && (wv == 0 || wv == WE_WAVES - 1) && s_blk < 4096) { g_we_stamps[s_blk][wv ? 1 : 0][6] = R; g_we_stamps[s_blk][wv ? 1 : 0][7] = total; }
#endif
}

// Emission of the big splats listed by tile_emit_kernel: one wave per splat, waves stride over the list.
// Lane = tile row: the spans of 64 rows are computed at once and scanned, then the lanes write the keys
// of those rows side by side.  (Walking the rows from the top for every key, as the first version did
// inside tile_emit, cost 1.38 ms of a 3.0 ms step at 260 k Gaussians and 1080p.)
__global__ __launch_bounds__(256) void tile_emit_slow_kernel(const uint32_t* __restrict__ slow_list,
                                                             const uint32_t* __restrict__ slow_count, int tile_size,
                                                             int tw, int H, uint32_t cap,
                                                             uint32_t* __restrict__ tile_keys,
                                                             uint32_t* __restrict__ flat_ids) {
    __shared__ uint32_t s_wcum[4][65];
    __shared__ uint16_t s_wtx0[4][64];
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const uint32_t n_slow = *slow_count;
    for (uint32_t ei = blockIdx.x * 4 + wv; ei < n_slow; ei += gridDim.x * 4) {
        const uint32_t* e = slow_list + (size_t)ei * SLOW_WORDS;
        SpanGeom geo;
        geo.mx = __uint_as_float(e[0]); geo.my = __uint_as_float(e[1]); geo.A = __uint_as_float(e[2]);
        geo.B = __uint_as_float(e[3]); geo.C = __uint_as_float(e[4]); geo.tau2 = __uint_as_float(e[5]);
        geo.x0 = (int)e[6]; geo.y0 = (int)e[7]; geo.x1 = (int)e[8]; geo.y1 = (int)e[9];
        geo.exact = e[10] != 0u;
        const uint32_t id = e[11], key0 = e[13];
        uint32_t outbase = e[12];
        const int rows = geo.y1 - geo.y0;
        for (int r0 = 0; r0 < rows; r0 += 64) {
            int tx0 = 0, len = 0;
            if (r0 + lane < rows) row_span(geo, geo.y0 + r0 + lane, tile_size, H, tx0, len);
            uint32_t inc = wave_incl_scan_u32((uint32_t)len);
            uint32_t total = __shfl(inc, 63, 64);
            s_wcum[wv][lane] = inc - (uint32_t)len;
            s_wtx0[wv][lane] = (uint16_t)tx0;
            if (lane == 63) s_wcum[wv][64] = total;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // same-wave LDS hand-off
            __builtin_amdgcn_wave_barrier();
            for (uint32_t q = lane; q < total; q += 64) {
                int lo = 0, hi = 63;                                    // last row whose first key is <= q
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    int mid = (lo + hi + 1) >> 1;
                    if (s_wcum[wv][mid] <= q) lo = mid; else hi = mid - 1;
                }
                uint32_t key = key0 + (uint32_t)((geo.y0 + r0 + lo) * tw) + (uint32_t)s_wtx0[wv][lo] + (q - s_wcum[wv][lo]);
                uint32_t p = outbase + q;
                if (p < cap) {
                    tile_keys[p] = key;
                    flat_ids[p] = id;
                }
            }
            __builtin_amdgcn_wave_barrier();
            outbase += total;
        }
    }
}

// offsets[t] = first sorted position whose key >= t   (t in [0, n_tiles_total)).  Four keys per thread
// (one 16-byte load plus the key before them); a boundary between two different keys writes the
// offsets of every tile id in between.
template <typename KT>
__global__ __launch_bounds__(256) void tile_offsets_kernel(const KT* __restrict__ keys,
                                                           const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                           uint32_t n_tiles_total, int32_t* __restrict__ offsets) {
    constexpr int KPT = 16 / (int)sizeof(KT);          // keys per thread: one 16-byte load (4 x u32 or 8 x u16)
    uint32_t n = live_count(n_ptr, cap);
    uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * KPT;
    if (n == 0) {
        for (uint32_t t = i0; t < i0 + KPT && t < n_tiles_total; t++) offsets[t] = 0;
        return;
    }
    if (i0 >= n) return;
    uint32_t k[KPT];
    if (i0 + KPT <= n) {
        const uint4 q = *reinterpret_cast<const uint4*>(keys + i0);
        if (sizeof(KT) == 4) {
            k[0] = q.x; k[1] = q.y; k[2] = q.z; k[3] = q.w;
        } else {
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++) { k[(2 * j) % KPT] = w[j] & 0xFFFFu; k[(2 * j + 1) % KPT] = w[j] >> 16; }
        }
    } else {
#pragma unroll
        for (int j = 0; j < KPT; j++) k[j] = i0 + j < n ? (uint32_t)keys[i0 + j] : 0u;
    }
    uint32_t prev = i0 == 0 ? 0u : (uint32_t)keys[i0 - 1];
    if (i0 == 0)
        for (uint32_t t = 0; t <= k[0]; t++) offsets[t] = 0;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        uint32_t i = i0 + j;
        if (i >= n) break;
        if (i > 0)
            for (uint32_t t = prev + 1; t <= k[j]; t++) offsets[t] = (int32_t)i;
        prev = k[j];
        if (i == n - 1)
            for (uint32_t t = k[j] + 1; t < n_tiles_total; t++) offsets[t] = (int32_t)n;
    }
}

__global__ __launch_bounds__(256) void isect_ids_kernel(const uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ flat_ids,
                                                        const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                        const float* __restrict__ splats, int64_t* __restrict__ out) {
    uint32_t n = live_count(n_ptr, cap);
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d = __float_as_uint(splats[(size_t)flat_ids[i] * SPLAT_STRIDE + SP_DEPTH]);
    out[i] = ((int64_t)keys[i] << 32) | (int64_t)d;
}

// two-phase path: mi3dgs_bin_count left the unclamped total; clamp it to the buffers' capacity
__global__ void clamp_count_kernel(uint32_t* n_isect, uint32_t cap, uint32_t* err) {
    if (*n_isect > cap) { *n_isect = cap; atomicOr(err, 4u); }
}

int g_emit_mode = 1;

struct BinWs {
    uint32_t *tiles, *dkeys_a, *dkeys_b, *ids_a, *ids_b, *cum, *n_isect, *slow, *tmp;
    uint32_t *tk_b, *fi_b;
    uint32_t* zero;          // fused path: every control word of the call, cleared by ONE memset (see bin_zero_layout)
    size_t zero_words;
};

// the fused path's control block: [n_live 16] [chain state 2 nchain + 16] [depth sort] [tile sort]
struct BinZero { uint32_t *n_live, *chain, *depth, *isect; };
size_t bin_zero_layout(uint32_t CN, uint32_t cap, uint32_t* base, BinZero* z);

size_t bin_zero_layout(uint32_t CN, uint32_t cap, uint32_t* base, BinZero* z) {
    size_t o = 0;
    auto take = [&](size_t n) { uint32_t* p = base ? base + o : nullptr; o += align_u32(n); return p; };
    const size_t nchain = we_chain_entries(CN);
    uint32_t* n_live = take(16);
    uint32_t* chain = take(2 * nchain + 16);
    uint32_t* depth = take(rs_zero_u32(CN, 32));
    uint32_t* isect = take(rs_zero_u32(cap, 32));
    if (z) *z = BinZero{n_live, chain, depth, isect};
    return o;
}

size_t bin_ws_layout(uint32_t CN, uint32_t cap, uint32_t* base, BinWs* ws) {
    size_t o = 0;
    auto take = [&](size_t n) { uint32_t* p = base ? base + o : nullptr; o += align_u32(n); return p; };
    uint32_t* tiles = take(CN);
    uint32_t* dka = take(CN);
    uint32_t* dkb = take(CN);
    uint32_t* ia = take(CN);
    uint32_t* ib = take(CN);
    uint32_t* cum = take(CN);
    uint32_t* ni = take(16);
    uint32_t* slow = take((size_t)CN * SLOW_WORDS);
    size_t t1 = rs_tmp_u32(CN), t2 = rs_tmp_u32(cap), t3 = scan_tmp_u32(CN);
    size_t tm = t1 > t2 ? t1 : t2;
    if (t3 > tm) tm = t3;
    uint32_t* tmp = take(tm);
    uint32_t* tkb = take(cap);
    uint32_t* fib = take(cap);
    const size_t zw = bin_zero_layout(CN, cap, nullptr, nullptr);
    uint32_t* zero = take(zw);
    if (ws) { *ws = BinWs{tiles, dka, dkb, ia, ib, cum, ni, slow, tmp, tkb, fib, zero, zw}; }
    return o * sizeof(uint32_t);
}

}  // namespace

extern "C" size_t mi3dgs_bin_workspace_bytes(int C, int N, long long max_isect) {
    return bin_ws_layout((uint32_t)((long long)C * N), (uint32_t)max_isect, nullptr, nullptr);
}

// Phase 1: tiles per Gaussian, depth sort, exclusive scan in depth order.
// Writes tiles_per_gauss[C*N] (caller's) and the total intersection count to n_isect_dev[0].
extern "C" int mi3dgs_bin_count(int C, int N, const int32_t* radii, const float* splats, int tile_size,
                                int tile_width, int tile_height, int height, int tight, int32_t* tiles_per_gauss,
                                int32_t* n_isect_dev, void* workspace, size_t workspace_bytes, long long max_isect,
                                void* stream) {
    long long CNl = (long long)C * N;
    MI_REQUIRE(CNl < (1ll << 31), "bin_count: C*N must be < 2^31");
    MI_REQUIRE((long long)C * tile_width * tile_height < (1ll << 31), "bin_count: too many tiles");
    uint32_t CN = (uint32_t)CNl;
    hipStream_t st = (hipStream_t)stream;
    if (CN == 0) { MI_HIP(hipMemsetAsync(n_isect_dev, 0, 4, st)); return 0; }
    BinWs ws;
    size_t need = bin_ws_layout(CN, (uint32_t)max_isect, (uint32_t*)workspace, &ws);
    MI_REQUIRE(workspace && workspace_bytes >= need, "bin_count: workspace too small");
    tight &= MI_BIN_TIGHT;
    if (tight)
        MI_LAUNCH("tile_count", tile_count_kernel<true>, dim3(mi_div_up(CN, 256)), dim3(256), 0, st, CN, radii, splats,
                  tile_size, tile_width, tile_height, height, ws.tiles, ws.dkeys_a, ws.ids_a);
    else
        MI_LAUNCH("tile_count", tile_count_kernel<false>, dim3(mi_div_up(CN, 256)), dim3(256), 0, st, CN, radii, splats,
                  tile_size, tile_width, tile_height, height, ws.tiles, ws.dkeys_a, ws.ids_a);
    int in_b = 0;
    int rc = radix_sort_pairs(ws.dkeys_a, ws.ids_a, ws.dkeys_b, ws.ids_b, nullptr, CN, 32, ws.tmp, &in_b, st, "depth");
    if (rc) return rc;
    // 4 passes -> result back in a
    uint32_t* sorted_ids = in_b ? ws.ids_b : ws.ids_a;
    uint32_t* gathered = in_b ? ws.dkeys_a : ws.dkeys_b;   // free key buffer as scratch
    MI_LAUNCH("gather_tiles", gather_u32_kernel, dim3(mi_div_up(CN, 256)), dim3(256), 0, st, CN, ws.tiles, sorted_ids, gathered);
    rc = scan_exclusive_u32(gathered, ws.cum, CN, ws.tmp, (uint32_t*)n_isect_dev, st);
    if (rc) return rc;
    if (tiles_per_gauss)
        MI_HIP(hipMemcpyAsync(tiles_per_gauss, ws.tiles, (size_t)CN * 4, hipMemcpyDeviceToDevice, st));
    if (in_b) MI_HIP(hipMemcpyAsync(ws.ids_a, ws.ids_b, (size_t)CN * 4, hipMemcpyDeviceToDevice, st));
    MI_LAUNCH_CHECK();
    return 0;
}

// Phase 2: emit (tile key, flat id) in depth order, stable sort by tile key, tile offsets.
// flatten_ids / tile_keys have max_isect entries; the live count is n_isect_dev[0] (device).
// stable sort of the emitted (tile key, splat) pairs on the tile bits, then the per-tile offsets
static int bin_sort_and_offsets(const BinWs& ws, uint32_t* tk, uint32_t* fi, const int32_t* n_isect_dev, uint32_t cap,
                                uint32_t n_tiles_total, const float* splats, int32_t* isect_offsets,
                                int64_t* isect_ids_opt, hipStream_t st, uint32_t* zeroed = nullptr) {
    int nbits = 1;
    while ((1u << nbits) < n_tiles_total) nbits++;
    int in_b = 0;
    int rc = radix_sort_pairs(tk, fi, ws.tk_b, ws.fi_b, (const uint32_t*)n_isect_dev, cap, nbits, ws.tmp, &in_b, st, "isect",
                              false, nullptr, zeroed);
    if (rc) return rc;
    if (in_b) {
        MI_HIP(hipMemcpyAsync(tk, ws.tk_b, (size_t)cap * 4, hipMemcpyDeviceToDevice, st));
        MI_HIP(hipMemcpyAsync(fi, ws.fi_b, (size_t)cap * 4, hipMemcpyDeviceToDevice, st));
    }
    uint32_t g = cap > n_tiles_total ? cap : n_tiles_total;
    MI_LAUNCH("tile_offsets", tile_offsets_kernel<uint32_t>, dim3(mi_div_up(mi_div_up(g, 4), 256)), dim3(256), 0, st, tk, (const uint32_t*)n_isect_dev,
                       cap, n_tiles_total, isect_offsets);
    if (isect_ids_opt)
        MI_LAUNCH("isect_ids", isect_ids_kernel, dim3(mi_div_up(cap, 256)), dim3(256), 0, st, tk, fi,
                           (const uint32_t*)n_isect_dev, cap, splats, isect_ids_opt);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_bin_emit(int C, int N, const int32_t* radii, const float* splats, int tile_size, int tile_width,
                               int tile_height, int height, int tight, int32_t* n_isect_dev, long long max_isect,
                               int32_t* flatten_ids,
                               int32_t* tile_keys, int32_t* isect_offsets, int64_t* isect_ids_opt, void* workspace,
                               size_t workspace_bytes, void* stream) {
    uint32_t CN = (uint32_t)((long long)C * N);
    uint32_t cap = (uint32_t)max_isect;
    uint32_t n_tiles_total = (uint32_t)(C * tile_width * tile_height);
    hipStream_t st = (hipStream_t)stream;
    if (CN == 0 || cap == 0) {
        MI_HIP(hipMemsetAsync(isect_offsets, 0, (size_t)n_tiles_total * 4, st));
        return 0;
    }
    BinWs ws;
    size_t need = bin_ws_layout(CN, cap, (uint32_t*)workspace, &ws);
    MI_REQUIRE(workspace && workspace_bytes >= need, "bin_emit: workspace too small");
    uint32_t* err = async_err_ptr();
    MI_REQUIRE(err, "bin_emit: no device error word");
    MI_LAUNCH("clamp_count", clamp_count_kernel, dim3(1), dim3(1), 0, st, (uint32_t*)n_isect_dev, cap, err);
    uint32_t* tk = (uint32_t*)tile_keys;
    uint32_t* fi = (uint32_t*)flatten_ids;
    uint32_t* slow_count = ws.n_isect + 4;
    const bool rir = (tight & MI_BIN_RADII_IN_RECORDS) != 0;
    tight &= MI_BIN_TIGHT;
    if (tight) {
        MI_HIP(hipMemsetAsync(slow_count, 0, 4, st));
        MI_LAUNCH("tile_emit", (tile_emit_kernel<true, false>), dim3(mi_div_up(CN, 256)), dim3(256), 0, st, CN, nullptr, rir, (uint32_t)N, ws.ids_a,
                  ws.cum, radii, splats, tile_size, tile_width, tile_height, height, cap, tk, fi, nullptr, nullptr,
                  nullptr, nullptr, nullptr, ws.slow, slow_count);
        MI_LAUNCH("tile_emit_slow", tile_emit_slow_kernel, dim3(SLOW_BLOCKS), dim3(256), 0, st, ws.slow, slow_count, tile_size,
                  tile_width, height, cap, tk, fi);
    } else {
        MI_LAUNCH("tile_emit", (tile_emit_kernel<false, false>), dim3(mi_div_up(CN, 256)), dim3(256), 0, st, CN, nullptr, rir, (uint32_t)N, ws.ids_a,
                  ws.cum, radii, splats, tile_size, tile_width, tile_height, height, cap, tk, fi, nullptr, nullptr,
                  nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    MI_LAUNCH_CHECK();
    return bin_sort_and_offsets(ws, tk, fi, n_isect_dev, cap, n_tiles_total, splats, isect_offsets, isect_ids_opt, st);
}

// One-call binning for callers that bring a capacity (the training step): depth sort, then count
// and emit fused in one chained pass.  Same outputs as mi3dgs_bin_count + mi3dgs_bin_emit.
extern "C" int mi3dgs_bin_tiles(int C, int N, const int32_t* radii, const float* splats, int tile_size, int tile_width,
                                int tile_height, int height, int tight, int32_t* n_isect_dev, long long max_isect,
                                int32_t* flatten_ids, int32_t* tile_keys, int32_t* isect_offsets,
                                int64_t* isect_ids_opt, int32_t* tiles_per_gauss_opt, uint32_t* depth_keys_opt,
                                void* workspace, size_t workspace_bytes, void* stream) {
    long long CNl = (long long)C * N;
    MI_REQUIRE(CNl < (1ll << 31), "bin_tiles: C*N must be < 2^31");
    MI_REQUIRE((long long)C * tile_width * tile_height < (1ll << 31), "bin_tiles: too many tiles");
    MI_REQUIRE(max_isect >= 0 && max_isect < (1ll << 31), "bin_tiles: bad max_isect");
    uint32_t CN = (uint32_t)CNl;
    uint32_t cap = (uint32_t)max_isect;
    uint32_t n_tiles_total = (uint32_t)(C * tile_width * tile_height);
    hipStream_t st = (hipStream_t)stream;
    MI_REQUIRE(n_isect_dev && isect_offsets, "bin_tiles: null output");
    if (CN == 0 || cap == 0) {
        MI_HIP(hipMemsetAsync(n_isect_dev, 0, 4, st));
        MI_HIP(hipMemsetAsync(isect_offsets, 0, (size_t)n_tiles_total * 4, st));
        if (tiles_per_gauss_opt && CN) MI_HIP(hipMemsetAsync(tiles_per_gauss_opt, 0, (size_t)CN * 4, st));
        return 0;
    }
    BinWs ws;
    size_t need = bin_ws_layout(CN, cap, (uint32_t*)workspace, &ws);
    MI_REQUIRE(workspace && workspace_bytes >= need, "bin_tiles: workspace too small");
    MI_REQUIRE(flatten_ids && tile_keys, "bin_tiles: null output");
    // depth keys: handed in by mi3dgs_project_fwd (consumed: the sort ping-pongs through them), or
    // gathered from the splat records here; the payload of the first pass is the index itself
    uint32_t* dkeys = depth_keys_opt ? depth_keys_opt : ws.dkeys_a;
    if (!depth_keys_opt)
        MI_LAUNCH("depth_keys", depth_keys_kernel, dim3(mi_div_up(CN, 256)), dim3(256), 0, st, CN, radii, splats, dkeys,
                  ws.ids_a);
    // the depth sort leaves the culled splats (key 0xFFFFFFFF) behind: everything after it walks the visible ones only
    const bool rir = (tight & MI_BIN_RADII_IN_RECORDS) != 0;
    const bool keys_scratch = (tight & MI_BIN_KEYS_SCRATCH) != 0;
    tight &= MI_BIN_TIGHT;
    // every control word of this call (live count, chain state, both sorts' histograms / status tables / scan scratch) sits in
    // one block cleared by one memset: the call used to issue seven small clears at ~4.5 us of stream time each
    BinZero z;
    bin_zero_layout(CN, cap, ws.zero, &z);
    MI_HIP(hipMemsetAsync(ws.zero, 0, ws.zero_words * sizeof(uint32_t), st));
    uint32_t* n_live = z.n_live;
    const bool wave_emit_path = (tight & MI_BIN_TIGHT) && g_emit_mode == 1 && tile_width <= WE_GROUPS;
    if (!wave_emit_path) MI_HIP(hipMemsetAsync(n_isect_dev, 0, 4, st));       // (the wave-granular emit always writes the count itself)
    if (tiles_per_gauss_opt) MI_HIP(hipMemsetAsync(tiles_per_gauss_opt, 0, (size_t)CN * 4, st));
    int in_b = 0;
    int rc = radix_sort_pairs(dkeys, ws.ids_a, ws.dkeys_b, ws.ids_b, nullptr, CN, 32, ws.tmp, &in_b, st, "depth",
                              /*identity_vals=*/depth_keys_opt != nullptr, n_live, z.depth);
    if (rc) return rc;
    const uint32_t* sorted_ids = in_b ? ws.ids_b : ws.ids_a;
    // chain state: status[nchain] u64 | counter | slow count, in the pre-cleared control block
    uint32_t nblocks = (uint32_t)mi_div_up(CN, 256);
    const uint32_t nchain = (uint32_t)we_chain_entries(CN);   // one status word per wave of the wave-granular emit (>= nblocks)
    unsigned long long* status = reinterpret_cast<unsigned long long*>(z.chain);
    uint32_t* counter = z.chain + 2 * (size_t)nchain;
    uint32_t* err = async_err_ptr();
    MI_REQUIRE(err, "bin_tiles: no device error word");
    uint32_t* slow_count = counter + 2;                    // cleared with the chain state
    uint32_t* tk = (uint32_t*)tile_keys;
    uint32_t* fi = (uint32_t*)flatten_ids;
    const bool wave_emit = tight && g_emit_mode == 1 && tile_width <= WE_GROUPS;
    // 16-bit tile keys: the caller does not want the sorted keys back (its tile_keys buffer is scratch), every tile id fits, and the sort is a classic one (the 16-bit kernels exist for that path only)
    static const bool k16_on = [] { const char* e = getenv("MI3DGS_KEYS16"); return !(e && e[0] == '0'); }();
    const bool k16 = k16_on && keys_scratch && wave_emit && !isect_ids_opt && n_tiles_total <= 65536u && cap >= 64u &&
                     !(g_sort_mode == 1 || g_sort_mode == 3 || (g_sort_mode == 2 && cap <= os_max_keys()));
    uint16_t* tk16 = (uint16_t*)tile_keys;
#define WE_LAUNCH(SPW_, KT_, TK_) MI_LAUNCH("tile_emit", (tile_emit_wave_kernel<SPW_, KT_>), dim3(mi_div_up(CN, SPW_ * WE_WAVES)), dim3(64 * WE_WAVES), 0, st, \
        CN, n_live, rir, (uint32_t)N, sorted_ids, radii, splats, tile_size, tile_width, tile_height, height, cap, TK_, fi, status, counter, err, \
        (uint32_t*)n_isect_dev, (uint32_t*)tiles_per_gauss_opt)
    if (wave_emit && we_spw_for(CN) == 16) { if (k16) WE_LAUNCH(16, uint16_t, tk16); else WE_LAUNCH(16, uint32_t, tk); }
    else if (wave_emit && we_spw_for(CN) == 32) { if (k16) WE_LAUNCH(32, uint16_t, tk16); else WE_LAUNCH(32, uint32_t, tk); }
    else if (wave_emit) { if (k16) WE_LAUNCH(64, uint16_t, tk16); else WE_LAUNCH(64, uint32_t, tk); }
#undef WE_LAUNCH
    else if (tight)
        MI_LAUNCH("tile_emit", (tile_emit_kernel<true, true>), dim3(nblocks), dim3(256), 0, st, CN, n_live, rir, (uint32_t)N, sorted_ids,
                  nullptr, radii, splats, tile_size, tile_width, tile_height, height, cap, tk, fi, status, counter, err,
                  (uint32_t*)n_isect_dev, (uint32_t*)tiles_per_gauss_opt, ws.slow, slow_count);
    else
        MI_LAUNCH("tile_emit", (tile_emit_kernel<false, true>), dim3(nblocks), dim3(256), 0, st, CN, n_live, rir, (uint32_t)N, sorted_ids,
                  nullptr, radii, splats, tile_size, tile_width, tile_height, height, cap, tk, fi, status, counter, err,
                  (uint32_t*)n_isect_dev, (uint32_t*)tiles_per_gauss_opt, nullptr, nullptr);
    if (tight && !wave_emit)
        MI_LAUNCH("tile_emit_slow", tile_emit_slow_kernel, dim3(SLOW_BLOCKS), dim3(256), 0, st, ws.slow, slow_count, tile_size,
                  tile_width, height, cap, tk, fi);
    MI_LAUNCH_CHECK();
    if (k16) {
        int nbits = 1;
        while ((1u << nbits) < n_tiles_total) nbits++;
        uint16_t* tk16_b = (uint16_t*)ws.tk_b;                      // the workspace's second key buffer (cap 32-bit words, half used)
        int in_b = 0;
        rc = radix_sort_pairs_k16(tk16, fi, tk16_b, ws.fi_b, (const uint32_t*)n_isect_dev, cap, nbits, ws.tmp, &in_b, st, "isect", z.isect);
        if (rc) return rc;
        const uint16_t* sorted_keys = in_b ? tk16_b : tk16;
        if (in_b) MI_HIP(hipMemcpyAsync(fi, ws.fi_b, (size_t)cap * 4, hipMemcpyDeviceToDevice, st));
        uint32_t g = cap > n_tiles_total ? cap : n_tiles_total;
        MI_LAUNCH("tile_offsets", tile_offsets_kernel<uint16_t>, dim3(mi_div_up(mi_div_up(g, 8), 256)), dim3(256), 0, st, sorted_keys,
                  (const uint32_t*)n_isect_dev, cap, n_tiles_total, isect_offsets);
        MI_LAUNCH_CHECK();
        return 0;
    }
    return bin_sort_and_offsets(ws, tk, fi, n_isect_dev, cap, n_tiles_total, splats, isect_offsets, isect_ids_opt, st, z.isect);
}


// Standalone entry points (exported for tests and for reuse by densify compaction).
extern "C" size_t mi3dgs_sort_workspace_bytes(long long n) {
    return (rs_tmp_u32((uint32_t)n) + 2 * align_u32((size_t)n)) * sizeof(uint32_t);
}

extern "C" int mi3dgs_sort_pairs_u32(uint32_t* keys, uint32_t* vals, long long n, int nbits, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    MI_REQUIRE(n >= 0 && n < (1ll << 31), "sort_pairs: bad n");
    MI_REQUIRE(nbits >= 1 && nbits <= 32, "sort_pairs: nbits must be in [1,32]");
    if (n == 0) return 0;
    MI_REQUIRE(workspace && workspace_bytes >= mi3dgs_sort_workspace_bytes(n), "sort_pairs: workspace too small");
    uint32_t* base = (uint32_t*)workspace;
    uint32_t* kb = base;
    uint32_t* vb = kb + align_u32((size_t)n);
    uint32_t* tmp = vb + align_u32((size_t)n);
    int in_b = 0;
    hipStream_t st = (hipStream_t)stream;
    int rc = radix_sort_pairs(keys, vals, kb, vb, nullptr, (uint32_t)n, nbits, tmp, &in_b, st);
    if (rc) return rc;
    if (in_b) {
        MI_HIP(hipMemcpyAsync(keys, kb, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        MI_HIP(hipMemcpyAsync(vals, vb, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

extern "C" size_t mi3dgs_scan_workspace_bytes(long long n) { return scan_tmp_u32((size_t)(n > 0 ? n : 0)) * 4; }

extern "C" int mi3dgs_scan_exclusive_u32(const uint32_t* in, uint32_t* out, long long n, uint32_t* total_dev,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(n >= 0 && n < (1ll << 31), "scan: bad n");
    MI_REQUIRE(n == 0 || (workspace && workspace_bytes >= mi3dgs_scan_workspace_bytes(n)), "scan: workspace too small");
    return scan_exclusive_u32(in, out, (uint32_t)n, (uint32_t*)workspace, total_dev, (hipStream_t)stream);
}

// A/B switch for benchmarking and tests: 0 = classic multi-kernel passes, 1 = onesweep (default).
// Bits set by chained kernels whose bounded waits ran out (0 = all chains resolved).  Synchronises
// with the device; `reset` clears the word.
extern "C" int mi3dgs_async_errors(uint32_t* out, int reset) {
    uint32_t v = 0;
    MI_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_async_err), sizeof(v)));
    if (out) *out = v;
    if (reset && v) {
        uint32_t z = 0;
        MI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_async_err), &z, sizeof(z)));
    }
    return 0;
}

// A/B switch: 1 (default) = wave-granular chained emit (tile_emit_wave_kernel), 0 = the block-cooperative one of round 1.
extern "C" int mi3dgs_debug_set_emit_mode(int mode) {
    g_emit_mode = mode ? 1 : 0;
    return 0;
}

extern "C" int mi3dgs_debug_set_sort_mode(int mode) {
    g_sort_mode = (mode < 0 || mode > 3) ? 2 : mode;
    return 0;
}
