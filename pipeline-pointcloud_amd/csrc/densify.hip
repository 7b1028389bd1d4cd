// Adaptive density control (densify / prune / opacity reset) with optimiser-state surgery,
// on device.  gfx950 only.
//
// Replaces gsplat DefaultStrategy._grow_gs / _prune_gs / reset_opa and the torch
// cat/index/zeros optimiser surgery of gsplat.strategy.ops (duplicate, split, remove,
// reset_opa), reached by the reference through main.py:1312 / main.py:1343
// (SURVEY.md 8a row a13).  Decision rule, per Gaussian with running stats (grad2d, count):
//   g = grad2d / max(count, 1);  high = g > grow_grad2d
//   small = max(exp(scale)) <= grow_scale3d * scene_scale
//   duplicate = high & small        -> original keeps its Adam state, copy gets zeros
//   split     = high & !small       -> original replaced by 2 samples, scale / 1.6, zero state
//   prune (evaluated on the grown set, as upstream does): sigmoid(opacity) < prune_opa,
//          or (step > reset_every and max(exp(scale)) > prune_scale3d * scene_scale)
// Screen-size rules (gsplat grow_scale2d / prune_scale2d / refine_scale2d_stop_iter = nerfstudio splatfacto's
// split_screen_size 0.05 / cull_screen_size 0.15 / stop_screen_size_at 4000 -- the DEFAULT job of the reference,
// main.py:1270-1306), on the running maximum r of radius / max(W, H) (stat_radii), while step < the stop iteration:
//   split |= r > grow_scale2d            (a Gaussian that is ALSO a duplicate gives three outputs: upstream duplicates
//                                         first and then splits the original: copy + two samples, all with zero state)
//   prune |= step > reset_every and r > prune_scale2d      (children inherit the parent's statistic upstream)
// The pass is: decide -> exclusive scan of output counts -> scatter into fresh buffers.
// Output order: survivors keep their relative order, children sit next to their parent
// (upstream appends them at the end; the order only breaks exact-depth ties).
//
// Bound: HBM streaming, (59 params + 2*59 Adam moments) * 4 B read and written per survivor;
// runs every 100 steps.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
    uint32_t state = x * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

// two independent N(0,1) samples from a counter
__device__ __forceinline__ void randn2(uint32_t seed, uint32_t ctr, float& a, float& b) {
    uint32_t h1 = pcg_hash(seed ^ pcg_hash(ctr * 2u + 0u));
    uint32_t h2 = pcg_hash(seed ^ pcg_hash(ctr * 2u + 1u));
    float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
    float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
    float r = sqrtf(-2.f * __logf(u1));
    float s, c;
    __sincosf(6.283185307179586f * u2, &s, &c);
    a = r * c; b = r * s;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// flags: bit0 duplicate, bit1 split, bit2 prune
__global__ __launch_bounds__(256) void densify_decide_kernel(
    int N, const float* __restrict__ scales_log, const float* __restrict__ opac_logit,
    const float* __restrict__ stat_grad2d, const float* __restrict__ stat_count, const float* __restrict__ stat_radii,
    float grow_grad2d, float grow_scale3d_abs, float grow_scale2d, float prune_opa, float prune_scale3d_abs,
    float prune_scale2d, int do_grow, int check_too_big, uint8_t* __restrict__ flags, uint32_t* __restrict__ out_count) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float smax = __expf(fmaxf(scales_log[3 * n], fmaxf(scales_log[3 * n + 1], scales_log[3 * n + 2])));
    bool dup = false, split = false;
    if (do_grow) {
        float g = stat_grad2d[n] / fmaxf(stat_count[n], 1.f);
        bool high = g > grow_grad2d;
        bool small = smax <= grow_scale3d_abs;
        dup = high && small;
        split = high && !small;
    }
    const float r2d = stat_radii ? stat_radii[n] : 0.f;            // (null = past the stop iteration, or a preset without the rule)
    if (do_grow && r2d > grow_scale2d) split = true;
    // (split children are smaller; a duplicate's copy keeps smax, but then smax <= grow_scale3d < prune_scale3d and nothing changes)
    float s_eff = (split && !dup) ? smax / 1.6f : smax;
    bool prune = sigmoidf_(opac_logit[n]) < prune_opa;
    if (check_too_big) prune = prune || (s_eff > prune_scale3d_abs) || (r2d > prune_scale2d);
    flags[n] = (uint8_t)((dup ? 1 : 0) | (split ? 2 : 0) | (prune ? 4 : 0));
    out_count[n] = prune ? 0u : 1u + (dup ? 1u : 0u) + (split ? 1u : 0u);
}

// ---- scatter = map + row gathers.  A first version copied rows one thread per Gaussian (45 floats at a
// 180-byte stride per lane): 3.4 ms for 2 M Gaussians, 1.1 TB/s.  Now a map pass writes, for every OUTPUT row,
// where it comes from; then each of the 18 arrays is rebuilt by a flat gather whose consecutive threads
// store consecutive 16-byte pieces of the output and read (almost always) consecutive pieces of the input,
// because survivors keep their order.
constexpr uint32_t MAP_ZERO_STATE = 1u << 31;     // the copy starts with zero Adam moments
constexpr uint32_t MAP_SPLIT = 1u << 30;          // position re-sampled, scale / 1.6
constexpr uint32_t MAP_SECOND = 1u << 29;         // second child (its own random sample)
constexpr uint32_t MAP_INDEX = (1u << 29) - 1u;

__global__ __launch_bounds__(256) void densify_map_kernel(int N, const uint8_t* __restrict__ flags,
                                                          const uint32_t* __restrict__ offsets, uint32_t cap,
                                                          uint32_t* __restrict__ src_of) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    uint8_t f = flags[n];
    if (f & 4) return;
    uint32_t o = offsets[n];
    bool dup = f & 1, split = f & 2;
    int copies = 1 + (dup ? 1 : 0) + (split ? 1 : 0);
    if (o + copies > cap) return;   // capacity guard; the host checks the total first
    if (dup && split) {             // duplicated, then the original split: the untouched copy first, then the two samples
        src_of[o] = (uint32_t)n | MAP_ZERO_STATE;
        src_of[o + 1] = (uint32_t)n | MAP_ZERO_STATE | MAP_SPLIT;
        src_of[o + 2] = (uint32_t)n | MAP_ZERO_STATE | MAP_SPLIT | MAP_SECOND;
        return;
    }
    for (int c = 0; c < copies; c++)
        src_of[o + c] = (uint32_t)n | ((split || c > 0) ? MAP_ZERO_STATE : 0u) | (split ? MAP_SPLIT : 0u) | (c ? MAP_SECOND : 0u);
}

// out[o][0..W) = in[src_of[o]][0..W), or zeros for a moment array of a fresh copy; 4 floats per thread
template <int W, bool MOMENT>
__global__ __launch_bounds__(256) void gather_rows_kernel(long long n_floats, const float* __restrict__ in,
                                                          float* __restrict__ out, const uint32_t* __restrict__ src_of) {
    long long e0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e0 >= n_floats) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        long long e = e0 + j;
        v[j] = 0.f;
        if (e < n_floats) {
            uint32_t o = (uint32_t)(e / W);
            uint32_t c = (uint32_t)(e - (long long)o * W);
            uint32_t m = src_of[o];
            if (!(MOMENT && (m & MAP_ZERO_STATE))) v[j] = in[(size_t)(m & MAP_INDEX) * W + c];
        }
    }
    if (e0 + 4 <= n_floats) *reinterpret_cast<float4*>(out + e0) = make_float4(v[0], v[1], v[2], v[3]);
    else
        for (int j = 0; j < 4 && e0 + j < n_floats; j++) out[e0 + j] = v[j];
}

// the two children of a split Gaussian: mean += R(q) (exp(s) .* randn(3)), s = log(exp(s) / 1.6)
__global__ __launch_bounds__(256) void densify_split_kernel(uint32_t n_out, const uint32_t* __restrict__ src_of,
                                                            const float* __restrict__ quats_in,
                                                            const float* __restrict__ scales_in,
                                                            float* __restrict__ means_out, float* __restrict__ scales_out,
                                                            uint32_t seed) {
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    uint32_t m = src_of[o];
    if (!(m & MAP_SPLIT)) return;
    uint32_t n = m & MAP_INDEX, c = (m & MAP_SECOND) ? 1u : 0u;
    const float* q = quats_in + (size_t)n * 4;
    float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    float inv = rsqrtf(fmaxf(n2, 1e-24f));
    float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
    float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
                  2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
                  2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)};
    const float* sl = scales_in + (size_t)n * 3;
    float r0, r1, r2, r3;
    randn2(seed, n * 4u + c * 2u, r0, r1);
    randn2(seed, n * 4u + c * 2u + 1u, r2, r3);
    float e[3] = {__expf(sl[0]) * r0, __expf(sl[1]) * r1, __expf(sl[2]) * r2};
    float* mo = means_out + (size_t)o * 3;
    float* so = scales_out + (size_t)o * 3;
    for (int i = 0; i < 3; i++) {
        mo[i] += R[3 * i] * e[0] + R[3 * i + 1] * e[1] + R[3 * i + 2] * e[2];
        so[i] = sl[i] - 0.47000362924573563f;   // log(1.6)
    }
}

template <int W>
int gather_group(long long n_out, const float* p_in, const float* m_in, const float* v_in, float* p_out, float* m_out,
                 float* v_out, const uint32_t* src_of, hipStream_t st) {
    long long nf = n_out * W;
    dim3 grid((unsigned)mi_div_up(mi_div_up(nf, 4), 256));
    MI_LAUNCH("densify_gather", (gather_rows_kernel<W, false>), grid, dim3(256), 0, st, nf, p_in, p_out, src_of);
    MI_LAUNCH("densify_gather", (gather_rows_kernel<W, true>), grid, dim3(256), 0, st, nf, m_in, m_out, src_of);
    MI_LAUNCH("densify_gather", (gather_rows_kernel<W, true>), grid, dim3(256), 0, st, nf, v_in, v_out, src_of);
    MI_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void reset_opacity_kernel(int N, float* __restrict__ opac_logit, float max_logit,
                                                            float* __restrict__ m, float* __restrict__ v) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    opac_logit[n] = fminf(opac_logit[n], max_logit);
    m[n] = 0.f;
    v[n] = 0.f;
}

}  // namespace

// Step 1: decision flags + per-Gaussian output counts.
extern "C" int mi3dgs_densify_decide(int N, const float* scales_log, const float* opacities_logit,
                                     const float* stat_grad2d, const float* stat_count, const float* stat_radii,
                                     float grow_grad2d, float grow_scale3d_abs, float grow_scale2d, float prune_opa,
                                     float prune_scale3d_abs, float prune_scale2d, int do_grow, int check_too_big,
                                     uint8_t* flags, uint32_t* out_count, void* stream) {
    if (N <= 0) return 0;
    MI_REQUIRE(!do_grow || (stat_grad2d && stat_count), "densify_decide: growing needs the running statistics");
    MI_REQUIRE(!stat_radii || (grow_scale2d > 0.f && prune_scale2d > 0.f), "densify_decide: screen-size thresholds must be positive");
    MI_LAUNCH("densify_decide", densify_decide_kernel, dim3(mi_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N, scales_log,
                       opacities_logit, stat_grad2d, stat_count, stat_radii, grow_grad2d, grow_scale3d_abs, grow_scale2d,
                       prune_opa, prune_scale3d_abs, prune_scale2d, do_grow, check_too_big, flags, out_count);
    MI_LAUNCH_CHECK();
    return 0;
}

// Step 2 (after mi3dgs_scan_exclusive_u32 over out_count, whose total `n_out` the host has read back):
// rebuild parameters and Adam moments of the 6 groups {means[3], quats[4], scales[3], opacities[1], sh0[3],
// shN[45]} in fresh buffers.  map_workspace: `capacity` u32.
extern "C" int mi3dgs_densify_scatter(int N, long long n_out, const float* const* params_in, const float* const* exp_avg_in,
                                      const float* const* exp_avg_sq_in, float* const* params_out,
                                      float* const* exp_avg_out, float* const* exp_avg_sq_out, const uint8_t* flags,
                                      const uint32_t* offsets, long long capacity, uint32_t seed, uint32_t* map_workspace,
                                      void* stream) {
    if (N <= 0 || n_out <= 0) return 0;
    MI_REQUIRE(n_out <= capacity && capacity <= (long long)MAP_INDEX, "densify_scatter: n_out exceeds the capacity");
    MI_REQUIRE(map_workspace, "densify_scatter: null map workspace");
    for (int g = 0; g < 6; g++)
        MI_REQUIRE(params_in[g] && exp_avg_in[g] && exp_avg_sq_in[g] && params_out[g] && exp_avg_out[g] && exp_avg_sq_out[g],
                   "densify_scatter: null buffer");
    hipStream_t st = (hipStream_t)stream;
    MI_LAUNCH("densify_map", densify_map_kernel, dim3(mi_div_up(N, 256)), dim3(256), 0, st, N, flags, offsets,
              (uint32_t)capacity, map_workspace);
    int rc = 0;
#define GG(g, W) rc = rc ? rc : gather_group<W>(n_out, params_in[g], exp_avg_in[g], exp_avg_sq_in[g], params_out[g], exp_avg_out[g], exp_avg_sq_out[g], map_workspace, st)
    GG(0, 3); GG(1, 4); GG(2, 3); GG(3, 1); GG(4, 3); GG(5, 45);
#undef GG
    if (rc) return rc;
    MI_LAUNCH("densify_split", densify_split_kernel, dim3(mi_div_up(n_out, 256)), dim3(256), 0, st, (uint32_t)n_out, map_workspace,
              params_in[1], params_in[2], params_out[0], params_out[2], seed);
    MI_LAUNCH_CHECK();
    return 0;
}

// opacities <- min(opacities, max_logit); Adam moments of the opacity group zeroed.
extern "C" int mi3dgs_reset_opacity(int N, float* opacities_logit, float max_logit, float* exp_avg, float* exp_avg_sq,
                                    void* stream) {
    if (N <= 0) return 0;
    MI_LAUNCH("reset_opacity", reset_opacity_kernel, dim3(mi_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N,
                       opacities_logit, max_logit, exp_avg, exp_avg_sq);
    MI_LAUNCH_CHECK();
    return 0;
}

// ======================================================================================
// MCMC strategy ("3D Gaussian Splatting as Markov Chain Monte Carlo", Kheradmand et al. 2024;
// gsplat MCMCStrategy + relocation.cu), selected by the reference with MODEL=splatfacto-mcmc
// (main.py:1285-1291) or `simple_trainer.py mcmc` (main.py:1324-1327).  Two kernels; the
// multinomial sampling and the row copies around them are host-side tensor plumbing.
// ======================================================================================
namespace {

constexpr int RELOC_N_MAX = 51;

// A Gaussian that is about to be split into `ratio` co-located copies keeps the rendered
// result (to first order) if every copy gets
//   o' = 1 - (1 - o)^(1/ratio),   s' = s * o / sum_{i=1..ratio} sum_{k=0..i-1} C(i-1,k) (-1)^k o'^(k+1) / sqrt(k+1)
__global__ __launch_bounds__(256) void mcmc_relocation_kernel(int n, const float* __restrict__ opac_in,
                                                              const float* __restrict__ scale_in,
                                                              const int32_t* __restrict__ ratios,
                                                              const float* __restrict__ binoms,
                                                              float* __restrict__ opac_out, float* __restrict__ scale_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int N = min(max(ratios[i], 1), RELOC_N_MAX);
    float o = opac_in[i];
    float no = 1.f - powf(1.f - o, 1.f / (float)N);
    float denom = 0.f;
    for (int a = 1; a <= N; a++) {
        float pw = no;                 // no^(k+1)
        for (int k = 0; k < a; k++) {
            float term = binoms[(a - 1) * RELOC_N_MAX + k] * ((k & 1) ? -1.f : 1.f) * pw * rsqrtf((float)(k + 1));
            denom += term;
            pw *= no;
        }
    }
    float coeff = o / denom;
    opac_out[i] = no;
    scale_out[3 * i] = coeff * scale_in[3 * i];
    scale_out[3 * i + 1] = coeff * scale_in[3 * i + 1];
    scale_out[3 * i + 2] = coeff * scale_in[3 * i + 2];
}

// means += Sigma * (randn(3) * gate(opacity) * scaler),  gate(o) = sigmoid(-k (o - (1 - x0)))
// with k = 100, x0 = 0.995: only nearly transparent Gaussians are perturbed.
__global__ __launch_bounds__(256) void mcmc_noise_kernel(int N, float* __restrict__ means, const float* __restrict__ quats,
                                                         const float* __restrict__ scales_log,
                                                         const float* __restrict__ opac_logit, float scaler,
                                                         uint32_t seed) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float o = sigmoidf_(opac_logit[n]);
    float gate = 1.f / (1.f + __expf(-100.f * ((1.f - o) - 0.995f)));
    float amp = gate * scaler;
    if (amp == 0.f) return;
    const float* q = quats + 4 * (size_t)n;
    float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    float inv = rsqrtf(fmaxf(n2, 1e-24f));
    float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
    float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
                  2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
                  2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)};
    float s2[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { float e = __expf(scales_log[3 * n + i]); s2[i] = e * e; }
    float r0, r1, r2, r3;
    randn2(seed, (uint32_t)n * 2u, r0, r1);
    randn2(seed, (uint32_t)n * 2u + 1u, r2, r3);
    float e[3] = {r0 * amp, r1 * amp, r2 * amp};
    // Sigma e = R S^2 R^T e
    float t[3];
#pragma unroll
    for (int j = 0; j < 3; j++) t[j] = (R[j] * e[0] + R[3 + j] * e[1] + R[6 + j] * e[2]) * s2[j];
#pragma unroll
    for (int i = 0; i < 3; i++) means[3 * n + i] += R[3 * i] * t[0] + R[3 * i + 1] * t[1] + R[3 * i + 2] * t[2];
}

// gradients of opacity_reg * mean(sigmoid(o)) + scale_reg * mean(exp(s)), accumulated
__global__ __launch_bounds__(256) void mcmc_reg_kernel(int N, const float* __restrict__ opac_logit,
                                                       const float* __restrict__ scales_log, float opacity_reg,
                                                       float scale_reg, float* __restrict__ v_opac,
                                                       float* __restrict__ v_scales) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float o = sigmoidf_(opac_logit[n]);
    v_opac[n] += opacity_reg * o * (1.f - o) / (float)N;
#pragma unroll
    for (int i = 0; i < 3; i++) v_scales[3 * n + i] += scale_reg * __expf(scales_log[3 * n + i]) / (3.f * (float)N);
}

}  // namespace

// opacities / scales are ACTIVATED values (sigmoid / exp); ratios[i] = number of copies the
// Gaussian is split into; binoms = [51][51] table of binomial coefficients C(n, k).
extern "C" int mi3dgs_mcmc_relocation(int n, const float* opacities, const float* scales, const int32_t* ratios,
                                      const float* binoms, float* new_opacities, float* new_scales, void* stream) {
    if (n <= 0) return 0;
    MI_LAUNCH("mcmc_relocation", mcmc_relocation_kernel, dim3(mi_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
              opacities, scales, ratios, binoms, new_opacities, new_scales);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_mcmc_inject_noise(int N, float* means, const float* quats, const float* scales_log,
                                        const float* opacities_logit, float scaler, uint32_t seed, void* stream) {
    if (N <= 0) return 0;
    MI_LAUNCH("mcmc_noise", mcmc_noise_kernel, dim3(mi_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N, means, quats,
              scales_log, opacities_logit, scaler, seed);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi3dgs_mcmc_regularise(int N, const float* opacities_logit, const float* scales_log, float opacity_reg,
                                      float scale_reg, float* v_opacities, float* v_scales, void* stream) {
    if (N <= 0) return 0;
    MI_LAUNCH("mcmc_reg", mcmc_reg_kernel, dim3(mi_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N, opacities_logit,
              scales_log, opacity_reg, scale_reg, v_opacities, v_scales);
    MI_LAUNCH_CHECK();
    return 0;
}
