// Shared helpers for the mi3dgs HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI_WAVE 64

// ---- splat record: one 64-byte line per (camera, Gaussian), written by project_fwd and
// gathered (one HBM/L2 request each) by the rasteriser.  See DESIGN.md "Data layout".
#define SPLAT_STRIDE 16
#define SP_X 0
#define SP_Y 1
#define SP_CA 2
#define SP_CB 3
#define SP_CC 4
#define SP_OPA 5
#define SP_R 6
#define SP_G 7
#define SP_B 8
#define SP_DEPTH 9
#define SP_COMP 10
#define SP_RX 11       // the integer radii (bit patterns), so that the binning needs ONE gather per splat
#define SP_RY 12

// ---- gradient record: one 64-byte line per (camera, Gaussian), accumulated by
// rasterize_bwd with float atomics (one 64-B atomic request per (tile, Gaussian)).
#define GRAD_STRIDE 16
#define GR_X 0
#define GR_Y 1
#define GR_CA 2
#define GR_CB 3
#define GR_CC 4
#define GR_OPA 5
#define GR_R 6
#define GR_G 7
#define GR_B 8
#define GR_ABSX 9
#define GR_ABSY 10
#define GR_DEPTH 11

#define MI_FLAG_LOG_SCALES 1      // scales are log-space parameters (exp applied in-kernel)
#define MI_FLAG_LOGIT_OPAC 2      // opacities are logits (sigmoid applied in-kernel)
#define MI_FLAG_ANTIALIASED 4     // rasterize_mode == "antialiased": opacity *= compensation
#define MI_FLAG_CLEAR_VSPLATS 8   // project_bwd / project_bwd_adam (one camera, SH): a visible Gaussian's v_splats row is zeroed once read
#define MI_FLAG_ONLY_CULLED_WAVES 16   // project_bwd_adam: only the 64-Gaussian groups none of whose members is visible (a pure Adam stream)
#define MI_FLAG_ONLY_VISIBLE_WAVES 32  // project_bwd_adam: only the groups with a visible member
#define MI_BIN_TIGHT 1             // mi3dgs_bin_*: `tight` argument, bit 0 = exact ellipse culling
#define MI_BIN_RADII_IN_RECORDS 2  //   bit 1 = take the radii from record slots SP_RX / SP_RY (written by project_fwd)
#define MI_BIN_KEYS_SCRATCH 4      //   bit 2 (mi3dgs_bin_tiles) = the caller does not read tile_keys back: the buffer is scratch (16-bit keys)

// ---- experiments.  Everything that can return WRONG results (timing experiments) and every variant that was measured and
// rejected lives behind -DMI3DGS_EXPERIMENTS (make exp -> libmi3dgs_exp.so, used by A/B tools and by the tests that
// compare the product path against a rejected-but-correct variant).  The product library accepts none of those switches:
// MI_EXPERIMENT_ENV() is a null pointer there, so the variable's NAME is not even in the binary
// (tests/test_cabi_cpu.py: `strings libmi3dgs.so | grep MI3DGS_` = the documented tuning knobs of include/mi3dgs.h).
#ifdef MI3DGS_EXPERIMENTS
#include <stdlib.h>
#define MI_EXPERIMENT_ENV(name) getenv(name)
#else
#define MI_EXPERIMENT_ENV(name) ((const char*)nullptr)
#endif

#define ALPHA_THRESHOLD (1.0f / 255.0f)
#define MAX_ALPHA 0.999f
#define T_STOP 1e-4f

int mi_set_error(const char* what, hipError_t e, const char* file, int line);
int mi_set_error_msg(const char* msg);

#define MI_LAUNCH_CHECK()                                                            \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) return mi_set_error("launch", e__, __FILE__, __LINE__); \
    } while (0)

#define MI_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) return mi_set_error(#call, e__, __FILE__, __LINE__); \
    } while (0)

#define MI_REQUIRE(cond, msg)                        \
    do {                                             \
        if (!(cond)) return mi_set_error_msg(msg);   \
    } while (0)

// Optional per-kernel profiler (api.cpp): when enabled every MI_LAUNCH is bracketed by HIP
// events recorded on the launch stream; bench.py reads the table back.
void mi_prof_begin(const char* tag, hipStream_t st);
void mi_prof_end(hipStream_t st);

#define MI_LAUNCH(tag, kernel, grid, block, shmem, st, ...)               \
    do {                                                                  \
        mi_prof_begin(tag, st);                                           \
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);  \
        mi_prof_end(st);                                                  \
    } while (0)

static inline int mi_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- wave64 reductions via DPP (no LDS traffic).  Result valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    // row_bcast:15 into rows 1 and 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));
    // row_bcast:31 into rows 2 and 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, true));
    return v;
}

// Nine wave64 sums at once, results in lane 63.  One fused v_add_f32_dpp per value and step,
// the nine independent chains interleaved: a DPP read needs 2 wait states after the VALU write
// of the same register, which the 8 other adds in between provide (the leading s_nop covers
// the compiler's last write).  hipcc turns the builtin form into mov_dpp + zero-init + add.
#define MI_DPP9(ctrl)                                                           \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n v_add_f32_dpp %1, %1, %1 " ctrl "\n"  \
    "v_add_f32_dpp %2, %2, %2 " ctrl "\n v_add_f32_dpp %3, %3, %3 " ctrl "\n"  \
    "v_add_f32_dpp %4, %4, %4 " ctrl "\n v_add_f32_dpp %5, %5, %5 " ctrl "\n"  \
    "v_add_f32_dpp %6, %6, %6 " ctrl "\n v_add_f32_dpp %7, %7, %7 " ctrl "\n"  \
    "v_add_f32_dpp %8, %8, %8 " ctrl "\n"
__device__ __forceinline__ void wave_sum9_to_lane63(float& a, float& b, float& c, float& d, float& e, float& f,
                                                    float& g, float& h, float& i) {
    asm volatile("s_nop 1\n"
                 MI_DPP9("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 MI_DPP9("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 MI_DPP9("row_half_mirror row_mask:0xf bank_mask:0xf")
                 MI_DPP9("row_mirror row_mask:0xf bank_mask:0xf")
                 MI_DPP9("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 MI_DPP9("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i));
}

// Eight wave64 sums as a reduce-SCATTER plus a ninth as a plain reduction.  On return every
// lane l holds in `a` the total of input value number (l >> 3) (a..h = values 0..7), and lane
// 63 holds the total of `i`.  24 VALU instead of 54: levels 32 and 16 exchange half of the
// registers with gfx950's v_permlane32_swap / v_permlane16_swap (2 instructions per pair of
// values), level 8 is a select + row_ror:8, and only ONE register goes through the last
// three in-row steps.  Checked on hardware by tools/micro/reduce_scatter_test.hip.
__device__ __forceinline__ void wave_reduce_scatter8_plus1(float& a, float& b, float& c, float& d, float& e, float& f,
                                                           float& g, float& h, float& i) {
    const unsigned long long m8 = 0xFF00FF00FF00FF00ull;   // lanes with bit 3 set
    float t;
    asm volatile(
        "s_nop 1\n"
        "v_permlane32_swap_b32 %0, %4\n v_permlane32_swap_b32 %1, %5\n"
        "v_permlane32_swap_b32 %2, %6\n v_permlane32_swap_b32 %3, %7\n"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7\n"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n"
        "s_nop 1\n"
        "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
        "v_add_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_cndmask_b32 %9, %1, %0, %10\n"
        "v_cndmask_b32 %0, %0, %1, %10\n"
        "v_add_f32_dpp %8, %8, %8 row_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_add_f32_dpp %0, %9, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 0\n"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %8, %8, %8 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 0\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i), "=&v"(t)
        : "s"(m8));
}

__device__ __forceinline__ float wave_sum_all(float v) {
    v = wave_sum_to_lane63(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ unsigned wave_sum_all_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One Adam update (torch.optim.Adam semantics): shared by optim.hip and the fused backward.
__device__ __forceinline__ void mi_adam1(float& p, float g, float& m, float& v, float step_size, float b1, float b2,
                                         float inv_bc2_sqrt, float eps) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p -= step_size * m / denom;
}

// 64-bit lane mask of a predicate.  HIP's __ballot(int) compares an INTEGER against zero, so a bool goes
// bool -> v_cndmask(0,1) -> v_cmp_ne_u32: two VALU instructions to rebuild a mask that already sits
// in an SGPR pair.  The builtin takes the i1 directly (an s_and with exec, or nothing at all).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
